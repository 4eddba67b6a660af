import ctypes, sys, os
for name in sys.argv[1:]:
    L = ctypes.CDLL(name)
    L.xfg_grid_blocks.restype = ctypes.c_uint32
    print(os.path.basename(name), L.xfg_grid_blocks(ctypes.c_uint32(1 << 20), 256))
