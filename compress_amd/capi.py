"""ctypes binding of include/xflate_amd.h.  Fails loudly if libxflate_amd.so is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# XFG_LIB selects another build of the same library (e.g. the diagnostic "make prof" build).
LIB_PATH = os.environ.get("XFG_LIB") or os.path.join(_HERE, "lib", "libxflate_amd.so")

# transport codes
XFG_OK, XFG_E_ARG, XFG_E_HIP, XFG_E_NODEVICE, XFG_E_NOMEM, XFG_E_LIMIT = range(6)
MAX_CHUNK_COMP, MAX_CHUNK_RAW = (1 << 28) - 64, (1 << 31) - 64      # XFG_MAX_CHUNK_COMP / _RAW
# per-chunk status
ST_OK, ST_CORRUPTED, ST_UNEXPECTED_EOF, ST_SIZE_MISMATCH, ST_NO_SYNC, ST_INTERNAL, ST_OUT_FULL = range(7)
ST_NAMES = ["ok", "corrupted", "unexpected_eof", "size_mismatch", "no_sync", "internal", "out_full"]
# chunk flags
CHUNK_END_BLOCK, CHUNK_SYNC, CHUNK_VERIFY, CHUNK_FOOTER, CHUNK_RFC_REP16, CHUNK_HOST_META = 1, 2, 4, 8, 16, 32
DEFLATE_DYNAMIC = 0x100
CHUNK_XFLATE_DEFLATE = CHUNK_END_BLOCK | CHUNK_SYNC | CHUNK_VERIFY | CHUNK_RFC_REP16
CHUNK_XFLATE_INDEX = CHUNK_END_BLOCK | CHUNK_VERIFY | CHUNK_RFC_REP16
CHUNK_XFLATE_FOOTER = CHUNK_END_BLOCK | CHUNK_VERIFY | CHUNK_FOOTER | CHUNK_RFC_REP16
# error classes
ERR_NONE, ERR_INTERNAL, ERR_INVALID, ERR_DEPRECATED, ERR_CORRUPTED, ERR_CLOSED = range(6)
ERR_EOF, ERR_UNEXPECTED_EOF, ERR_IO, ERR_DEVICE, ERR_LIMIT = 16, 17, 18, 19, 20
REC_UNKNOWN, REC_DEFLATE, REC_INDEX, REC_FOOTER = range(4)


class Chunk(ctypes.Structure):
    _fields_ = [("comp_off", ctypes.c_uint64), ("comp_size", ctypes.c_uint64),
                ("raw_off", ctypes.c_uint64), ("raw_size", ctypes.c_uint64),
                ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class Result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("reserved", ctypes.c_uint32),
                ("in_consumed", ctypes.c_int64), ("out_produced", ctypes.c_int64)]


class Record(ctypes.Structure):
    _fields_ = [("comp_offset", ctypes.c_int64), ("raw_offset", ctypes.c_int64),
                ("type", ctypes.c_int32), ("reserved", ctypes.c_int32)]


READ_FN = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64)
SEEK_FN = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int)

# every symbol include/xflate_amd.h declares
SYMBOLS = [
    "xfg_device_count", "xfg_create", "xfg_destroy", "xfg_strerror", "xfg_last_error",
    "xfg_decode", "xfg_decode_device", "xfg_sync",
    "xfg_reader_new", "xfg_reader_new_mem", "xfg_reader_reset_mem", "xfg_reader_reset", "xfg_reader_read", "xfg_reader_seek", "xfg_reader_close",
    "xfg_reader_free", "xfg_reader_num_records", "xfg_reader_get_records", "xfg_reader_bytes_read",
    "xfg_reader_set_window", "xfg_index_parse", "xfg_shard_plan", "xfg_flate_decode",
    "xfg_crc32_device", "xfg_crc32_combine", "xfg_decode_device_noalloc",
    "xfg_deflate_bound", "xfg_deflate", "xfg_deflate_device",
    "xfg_set_decode_waves", "xfg_decode_waves_for",
]

_lib = None


def lib():
    """Load libxflate_amd.so (built by __graft_entry__.build() / make -C compress_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libxflate_amd.so not built: run `make -C compress_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64
    L.xfg_device_count.restype = ctypes.c_int
    L.xfg_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.xfg_destroy.argtypes = [vp]
    L.xfg_destroy.restype = None
    L.xfg_strerror.argtypes = [ctypes.c_int]
    L.xfg_strerror.restype = ctypes.c_char_p
    L.xfg_last_error.restype = ctypes.c_char_p
    L.xfg_decode.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    L.xfg_decode_device.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    L.xfg_decode_device_noalloc.argtypes = [ctypes.c_int, vp, sz, vp, sz, vp, sz, vp, vp]
    L.xfg_sync.argtypes = [vp]
    L.xfg_reader_new.argtypes = [vp, READ_FN, SEEK_FN, vp, ctypes.POINTER(vp)]
    L.xfg_reader_new_mem.argtypes = [vp, vp, sz, ctypes.POINTER(vp)]
    L.xfg_reader_reset_mem.argtypes = [vp, vp, sz]
    L.xfg_reader_reset.argtypes = [vp, READ_FN, SEEK_FN, vp]
    L.xfg_reader_read.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_int)]
    L.xfg_reader_read.restype = i64
    L.xfg_reader_seek.argtypes = [vp, i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.xfg_reader_seek.restype = i64
    L.xfg_reader_close.argtypes = [vp]
    L.xfg_reader_free.argtypes = [vp]
    L.xfg_reader_free.restype = None
    L.xfg_reader_num_records.argtypes = [vp]
    L.xfg_reader_num_records.restype = i64
    L.xfg_reader_get_records.argtypes = [vp, vp, i64]
    L.xfg_reader_bytes_read.argtypes = [vp]
    L.xfg_reader_bytes_read.restype = i64
    L.xfg_reader_set_window.argtypes = [vp, i64]
    L.xfg_index_parse.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz)]
    L.xfg_shard_plan.argtypes = [vp, sz, ctypes.c_int, ctypes.POINTER(sz)]
    L.xfg_flate_decode.argtypes = [vp, vp, sz, vp, sz, ctypes.POINTER(i64), ctypes.POINTER(i64),
                                   ctypes.POINTER(ctypes.c_int)]
    L.xfg_deflate_bound.argtypes = [sz]
    L.xfg_deflate_bound.restype = sz
    L.xfg_deflate.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    L.xfg_deflate_device.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, vp]
    L.xfg_set_decode_waves.argtypes = [ctypes.c_int]
    L.xfg_decode_waves_for.argtypes = [ctypes.c_uint32, ctypes.c_int]
    _lib = L
    return L


_hooks = None


def testhooks_lib():
    """libxflate_amd_testhooks.so: host_reader.cpp built once more with the test hooks of include/xflate_amd_testhooks.h
    (tests only; the shipped library does not export them)."""
    global _hooks
    if _hooks is None:
        lib()                                           # the hooks library is linked against the shipped one
        path = os.path.join(os.path.dirname(LIB_PATH), "libxflate_amd_testhooks.so")
        if not os.path.exists(path):
            raise ImportError("libxflate_amd_testhooks.so not built: run `make -C compress_amd/csrc`")
        _hooks = ctypes.CDLL(path)
    return _hooks


class TransportError(RuntimeError):
    pass


def set_decode_waves(waves):
    """xfg_set_decode_waves: 0 = by chunk count, 1 = one wavefront per chunk, 2..16 = a workgroup of that many."""
    check(lib().xfg_set_decode_waves(waves), "xfg_set_decode_waves(%d)" % waves)


def check(rc, what="xfg call"):
    if rc != XFG_OK:
        L = lib()
        raise TransportError("%s failed: %s (%s)" % (what, L.xfg_strerror(rc).decode(), L.xfg_last_error().decode()))


class Context:
    """One device context (xfg_create/xfg_destroy)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        check(lib().xfg_create(device, ctypes.byref(self._h)), "xfg_create(%d)" % device)
        self.device = device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().xfg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib().xfg_sync(self._h), "xfg_sync")

    def decode(self, comp, chunks, raw_len):
        """Host-pointer batch decode.  chunks: list of (comp_off, comp_size, raw_off, raw_size, flags).
        -> (raw bytearray, [Result...])"""
        n = len(chunks)
        arr = (Chunk * max(n, 1))()
        for i, c in enumerate(chunks):
            arr[i] = Chunk(c[0], c[1], c[2], c[3], c[4], 0)
        res = (Result * max(n, 1))()
        comp = bytes(comp)
        raw = ctypes.create_string_buffer(max(raw_len, 1))
        check(lib().xfg_decode(self._h, comp, len(comp), arr, n, raw, raw_len, res), "xfg_decode")
        return raw, [res[i] for i in range(n)]

    def decode_device(self, d_comp, comp_len, d_chunks, n, d_raw, raw_len, d_res, stream=None):
        """Device-pointer batch decode (async).  Pointers are ints (e.g. torch tensor .data_ptr())."""
        check(lib().xfg_decode_device(self._h, d_comp, comp_len, d_chunks, n, d_raw, raw_len, d_res,
                                      stream or None), "xfg_decode_device")

    def deflate(self, raw, chunk_size, dynamic=False):
        """xfg_deflate: compress `raw` in chunks of chunk_size bytes (the last one shorter).
        -> list of compressed chunks, each a DEFLATE stream without a final block ending in 00 00 ff ff.
        dynamic: XFG_DEFLATE_DYNAMIC (two passes, Huffman codes built for the chunk)."""
        raw = bytes(raw)
        sizes = [min(chunk_size, len(raw) - o) for o in range(0, len(raw), chunk_size)] or [0]
        n = len(sizes)
        arr = (Chunk * n)()
        co = ro = 0
        for i, rs in enumerate(sizes):
            b = lib().xfg_deflate_bound(rs)
            arr[i] = Chunk(co, b, ro, rs, DEFLATE_DYNAMIC if dynamic else 0, 0)
            co += b
            ro += rs
        res = (Result * n)()
        comp = ctypes.create_string_buffer(max(co, 1))
        check(lib().xfg_deflate(self._h, raw, len(raw), arr, n, comp, co, res), "xfg_deflate")
        out = []
        for i in range(n):
            if res[i].status != ST_OK:
                raise TransportError("xfg_deflate: chunk %d status %s" % (i, ST_NAMES[res[i].status]))
            out.append(comp.raw[arr[i].comp_off:arr[i].comp_off + res[i].out_produced])
        return out

    def deflate_chunks(self, chunks, dynamic=False):
        """xfg_deflate over chunks of any sizes (a list of bytes-like) in one launch -> list of compressed chunks."""
        n = len(chunks)
        if n == 0:
            return []
        raw = b"".join(bytes(c) for c in chunks)
        arr = (Chunk * n)()
        co = ro = 0
        for i, c in enumerate(chunks):
            b = lib().xfg_deflate_bound(len(c))
            arr[i] = Chunk(co, b, ro, len(c), DEFLATE_DYNAMIC if dynamic else 0, 0)
            co += b
            ro += len(c)
        res = (Result * n)()
        comp = ctypes.create_string_buffer(max(co, 1))
        check(lib().xfg_deflate(self._h, raw, len(raw), arr, n, comp, co, res), "xfg_deflate")
        mv = memoryview(comp).cast("B")
        out = []
        for i in range(n):
            if res[i].status != ST_OK:
                raise TransportError("xfg_deflate: chunk %d status %s" % (i, ST_NAMES[res[i].status]))
            out.append(bytes(mv[arr[i].comp_off:arr[i].comp_off + res[i].out_produced]))
        return out

    def deflate_device(self, d_raw, raw_len, d_chunks, n, d_comp, comp_len, d_res, stream=None):
        """xfg_deflate_device (async); pointers are ints."""
        check(lib().xfg_deflate_device(self._h, d_raw, raw_len, d_chunks, n, d_comp, comp_len, d_res,
                                       stream or None), "xfg_deflate_device")

    def crc32_device(self, d_raw, raw_len, d_chunks, n, d_res, d_crc, stream=None):
        """xfg_crc32_device (async): CRC-32 of every chunk's decoded bytes into uint32 d_crc[n]; d_res may be 0."""
        f = lib().xfg_crc32_device
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_void_p]
        check(f(self._h, d_raw, raw_len, d_chunks, n, d_res or None, d_crc, stream or None), "xfg_crc32_device")


def crc32_combine(crc_a, crc_b, len_b):
    """xfg_crc32_combine: crc32(A || B) from crc32(A), crc32(B), len(B)."""
    f = lib().xfg_crc32_combine
    f.restype = ctypes.c_uint32
    f.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
    return f(crc_a, crc_b, len_b)


def index_parse(stream):
    """xfg_index_parse: -> (err class, [Chunk...]) for every record of an in-memory XFLATE stream."""
    stream = bytes(stream)
    n = ctypes.c_size_t()
    err = lib().xfg_index_parse(stream, len(stream), None, 0, ctypes.byref(n))
    if err != ERR_NONE:
        return err, []
    arr = (Chunk * max(n.value, 1))()
    err = lib().xfg_index_parse(stream, len(stream), arr, n.value, ctypes.byref(n))
    return err, [arr[i] for i in range(n.value)]


def shard_plan(chunks, n_shards):
    """xfg_shard_plan: -> [first_0, ..., first_n_shards]; shard k owns records [first[k], first[k+1])."""
    n = len(chunks)
    arr = (Chunk * max(n, 1))(*chunks)
    first = (ctypes.c_size_t * (n_shards + 1))()
    check(lib().xfg_shard_plan(arr, n, n_shards, first), "xfg_shard_plan")
    return list(first)
