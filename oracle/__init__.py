"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (compress_amd/) never does.

  inflate_oracle.c  C restatement of the reference's flate.Reader (see its header)
  xflate_oracle.py  Python restatement of meta.Reader / index / xflate.Reader
  bitgen.py         decoder for the reference's BitGen vector DSL
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

OK, CORRUPTED, UNEXPECTED_EOF, OUT_FULL = 0, 1, 2, 3
STATUS_NAMES = {0: "ok", 1: "corrupted", 2: "unexpected_eof", 3: "out_full"}


def build(force=False):
    """Compile liboracle.so with gcc (no-op if already built)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "inflate_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


class OrcChunk(ctypes.Structure):
    _fields_ = [("comp_off", ctypes.c_uint64), ("comp_size", ctypes.c_uint64),
                ("raw_off", ctypes.c_uint64), ("raw_size", ctypes.c_uint64),
                ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class OrcResult(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("reserved", ctypes.c_uint32),
                ("in_consumed", ctypes.c_int64), ("out_produced", ctypes.c_int64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_inflate.restype = ctypes.c_int
        _lib.oracle_inflate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _lib.oracle_inflate_ex.restype = ctypes.c_int
        _lib.oracle_inflate_ex.argtypes = _lib.oracle_inflate.argtypes + [ctypes.c_int]
        _lib.oracle_inflate_chunk.restype = ctypes.c_int
        _lib.oracle_inflate_chunk.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _lib.oracle_inflate_batch.restype = ctypes.c_int
        _lib.oracle_inflate_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _lib


def inflate(data, out_cap=None, rfc_rep16=False):
    """Inflate one raw DEFLATE stream. -> (status, output bytes, in_off, out_off).

    rfc_rep16=False: the in-repo flate.Reader rule for code-length symbol 16;
    True: the RFC 1951 / Go stdlib rule that xflate.Reader's inflater uses
    (see read_prefix_codes in inflate_oracle.c)."""
    data = bytes(data)
    if out_cap is None:
        out_cap = max(1 << 16, len(data) * 1100)       # DEFLATE expands at most ~1032x
    out = ctypes.create_string_buffer(max(out_cap, 1))
    i, o = ctypes.c_int64(), ctypes.c_int64()
    st = lib().oracle_inflate_ex(data, len(data), out, out_cap, ctypes.byref(i), ctypes.byref(o),
                                 1 if rfc_rep16 else 0)
    return st, out.raw[:o.value], i.value, o.value


def inflate_chunk(comp, out_cap):
    """xflate.Reader's view of one chunk: inflate comp || 01 00 00 ff ff."""
    comp = bytes(comp)
    tmp = ctypes.create_string_buffer(len(comp) + 5)
    out = ctypes.create_string_buffer(max(out_cap, 1))
    i, o = ctypes.c_int64(), ctypes.c_int64()
    st = lib().oracle_inflate_chunk(comp, len(comp), tmp, out, out_cap, ctypes.byref(i), ctypes.byref(o))
    return st, out.raw[:o.value], i.value, o.value
