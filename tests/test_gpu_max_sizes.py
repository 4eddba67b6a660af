"""The documented per-chunk limits of the device path (include/xflate_amd.h: XFG_MAX_CHUNK_RAW = 2^31 - 64 decompressed,
XFG_MAX_CHUNK_COMP = 2^28 - 64 compressed bytes; bit positions inside a chunk are 32-bit).  The reference has no such
limits and no fixture of these sizes, so the pin here is the CPU oracle / the generator: a chunk AT each limit decodes
bit-exactly, one byte more is refused with XFG_E_LIMIT before anything is launched."""
import ctypes
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _decode_one(ctx, comp, raw_size, flags):
    from compress_amd import capi
    L = capi.lib()
    arr = (capi.Chunk * 1)(capi.Chunk(0, len(comp), 0, raw_size, flags, 0))
    res = (capi.Result * 1)()
    out = np.empty(raw_size + 64, dtype=np.uint8)
    out[raw_size:] = 0xEE
    rc = L.xfg_decode(ctx.handle, comp, len(comp), arr, 1, out.ctypes.data_as(ctypes.c_void_p), raw_size, res)
    return rc, res[0], out


def test_chunk_of_maximum_raw_size(ctx):
    """2^31 - 64 bytes out of one chunk (one wave; the last bit positions of the output counters)."""
    from compress_amd import capi
    n = capi.MAX_CHUNK_RAW
    z = zlib.compressobj(1, zlib.DEFLATED, -15)
    zeros = bytes(n)
    comp = z.compress(zeros) + z.flush(zlib.Z_SYNC_FLUSH)
    want_crc = zlib.crc32(zeros)
    del zeros
    assert len(comp) < capi.MAX_CHUNK_COMP
    rc, res, out = _decode_one(ctx, comp, n, capi.CHUNK_XFLATE_DEFLATE)
    assert rc == 0
    assert (res.status, res.out_produced, res.in_consumed) == (capi.ST_OK, n, len(comp) + 5)
    assert zlib.crc32(memoryview(out)[:n]) == want_crc
    assert bytes(out[n:]) == b"\xEE" * 64            # nothing beyond the chunk's range
    # one byte more is beyond the limit: refused before any launch, whatever the data
    arr = (capi.Chunk * 1)(capi.Chunk(0, len(comp), 0, n + 1, capi.CHUNK_XFLATE_DEFLATE, 0))
    resv = (capi.Result * 1)()
    buf = ctypes.create_string_buffer(16)
    assert capi.lib().xfg_decode(ctx.handle, comp, len(comp), arr, 1, buf, n + 1, resv) == capi.XFG_E_LIMIT


def test_chunk_of_maximum_compressed_size(ctx):
    """2^28 - 64 compressed bytes in one chunk: stored blocks of random bytes (the only way to get there quickly)."""
    from compress_amd import capi
    limit = capi.MAX_CHUNK_COMP
    # stored coding: 5 bytes per block of <= 65535, then the 5-byte sync marker; the last block makes the size exact
    full = (limit - 5) // 65540
    rest = (limit - 5) - full * 65540
    raw_size = full * 65535 + (rest - 5 if rest > 5 else 0)
    rng = np.random.Generator(np.random.PCG64(11))
    raw = rng.integers(0, 256, raw_size, dtype=np.uint8).tobytes()
    parts = []
    for o in range(0, raw_size, 65535):
        k = min(65535, raw_size - o)
        parts.append(bytes([0, k & 255, k >> 8, (~k) & 255, ((~k) >> 8) & 255]) + raw[o:o + k])
    comp = b"".join(parts) + b"\x00\x00\x00\xff\xff"
    assert limit - 5 <= len(comp) <= limit
    rc, res, out = _decode_one(ctx, comp, raw_size, capi.CHUNK_XFLATE_DEFLATE)
    assert rc == 0
    assert (res.status, res.out_produced, res.in_consumed) == (capi.ST_OK, raw_size, len(comp) + 5)
    assert memoryview(out)[:raw_size] == raw
    arr = (capi.Chunk * 1)(capi.Chunk(0, limit + 1, 0, raw_size, capi.CHUNK_XFLATE_DEFLATE, 0))
    resv = (capi.Result * 1)()
    big = np.zeros(limit + 1, dtype=np.uint8)
    buf = ctypes.create_string_buffer(16)
    assert capi.lib().xfg_decode(ctx.handle, big.ctypes.data_as(ctypes.c_void_p), limit + 1, arr, 1, buf, raw_size, resv) == capi.XFG_E_LIMIT
