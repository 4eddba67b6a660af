"""Large Reads through xfg_reader: a request of 1 MiB or more takes a run of clean chunks out of the decoded window in
one copy (host_reader.cpp, read()).  It must be indistinguishable from chunk-by-chunk Reads (xflate/reader.go:162-215)
except for how many bytes one call returns: same bytes, same error class after the same number of bytes, same
positions after Seek.  Also xfg_reader_new_mem / xfg_reader_reset_mem (the stream read where it lies)."""
import ctypes
import io
import os
import sys

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


@pytest.fixture(scope="module")
def stream_and_raw():
    from compress_amd import synth
    stream, raw = synth.build_mix_stream(96, 40000, seed=77, index_size=20)      # 3.8 MB, several indexes inside
    assert raw is not None
    return stream, raw


def _drain(L, h, req):
    """Read to the end (or the first error) with requests of `req` bytes -> (bytes, error class, largest return)."""
    from compress_amd import capi
    buf = ctypes.create_string_buffer(req)
    err = ctypes.c_int()
    out = bytearray()
    biggest = 0
    while True:
        k = L.xfg_reader_read(h, buf, req, ctypes.byref(err))
        if k <= 0:
            return bytes(out), err.value, biggest
        assert err.value == capi.ERR_NONE
        out += buf.raw[:k]
        biggest = max(biggest, k)


def _mem_reader(L, ctx, stream):
    h = ctypes.c_void_p()
    cls = L.xfg_reader_new_mem(ctx.handle, stream, len(stream), ctypes.byref(h))
    return h, cls


def test_bulk_equals_chunkwise(ctx, stream_and_raw):
    from compress_amd import capi, xflate
    stream, raw = stream_and_raw
    L = capi.lib()
    h, cls = _mem_reader(L, ctx, stream)
    assert cls == capi.ERR_NONE
    small, e1, big1 = _drain(L, h, 3000)
    assert L.xfg_reader_reset_mem(h, stream, len(stream)) == capi.ERR_NONE
    large, e2, big2 = _drain(L, h, 2 << 20)
    L.xfg_reader_free(h)
    assert small == raw and large == raw
    assert e1 == e2 == capi.ERR_EOF
    assert big1 <= 3000 and big2 > 40000            # the large requests did span chunks
    # and through the callback reader
    xr, err = xflate.NewReader(io.BytesIO(stream), ctx=ctx)
    assert err is None
    h2 = xr._h
    cb, e3, big3 = _drain(L, h2, 2 << 20)
    assert cb == raw and e3 == capi.ERR_EOF and big3 > 40000
    assert xr.bytes_read() <= len(stream) + 4096
    xr.Close()


@pytest.mark.parametrize("victim,where", [(0, 0.5), (17, 0.5), (17, 0.02), (40, 0.9), (95, 0.5)])
def test_bulk_error_position(ctx, stream_and_raw, victim, where):
    """A damaged chunk in the run: the bytes before it (and what it produced before the error) arrive, then the error."""
    from compress_amd import capi
    from oracle import xflate_oracle
    stream, raw = stream_and_raw
    L = capi.lib()
    h, _ = _mem_reader(L, ctx, stream)
    n = L.xfg_reader_num_records(h)
    recs = (capi.Record * n)()
    L.xfg_reader_get_records(h, recs, n)
    L.xfg_reader_free(h)
    deflate = [i for i in range(n) if recs[i].type == 1]
    i = deflate[victim]
    lo = recs[i - 1].comp_offset if i else 0
    hi = recs[i].comp_offset
    bad = bytearray(stream)
    bad[lo + int((hi - lo - 6) * where)] ^= 0x41
    bad = bytes(bad)
    results = []
    for req in (2500, 1 << 20, 3 << 20):
        h, cls = _mem_reader(L, ctx, bad)
        assert cls == capi.ERR_NONE                # index and footer are intact
        results.append(_drain(L, h, req)[:2])
        L.xfg_reader_free(h)
    assert results[0] == results[1] == results[2]
    xr = xflate_oracle.XflateReader(bad)
    assert xr.err is None
    want, werr = xr.read_all()
    got, cls = results[0]
    assert got == want
    if werr is None:
        assert cls == capi.ERR_EOF                 # the flip did not change what decodes
    else:
        assert cls in (capi.ERR_CORRUPTED, capi.ERR_UNEXPECTED_EOF)


def test_bulk_after_seek(ctx, stream_and_raw):
    from compress_amd import capi
    stream, raw = stream_and_raw
    L = capi.lib()
    h, _ = _mem_reader(L, ctx, stream)
    err = ctypes.c_int()
    buf = ctypes.create_string_buffer(2 << 20)
    for pos in (0, 1, 39999, 40000, 40001, 1234567, len(raw) - 5, len(raw)):
        assert L.xfg_reader_seek(h, pos, 0, ctypes.byref(err)) == pos and err.value == capi.ERR_NONE
        got = bytearray()
        while len(got) < (2 << 20):
            k = L.xfg_reader_read(h, buf, (2 << 20) - len(got), ctypes.byref(err))
            if k <= 0:
                break
            got += buf.raw[:k]
        assert bytes(got) == raw[pos:pos + (2 << 20)], pos
        # the position the Reader reports afterwards
        assert L.xfg_reader_seek(h, 0, 1, ctypes.byref(err)) == min(pos + (2 << 20), len(raw))
    L.xfg_reader_free(h)


def test_mem_reader_bad_stream(ctx):
    from compress_amd import capi
    L = capi.lib()
    h, cls = _mem_reader(L, ctx, b"not an xflate stream at all")
    assert cls == capi.ERR_CORRUPTED and h           # like the reference: a reader together with the error
    err = ctypes.c_int()
    buf = ctypes.create_string_buffer(16)
    assert L.xfg_reader_read(h, buf, 16, ctypes.byref(err)) == 0 and err.value == capi.ERR_CORRUPTED
    L.xfg_reader_free(h)


def test_direct_into_callers_buffer(ctx):
    """Requests of 32 MiB and more are decoded straight into the caller's buffer (no window): same bytes, same error
    position as small reads, with and without a damaged chunk, and after a Seek into the middle of a chunk."""
    from compress_amd import capi, synth
    stream, raw = synth.build_mix_stream(1200, 65536, seed=5, n_distinct=48)     # 75 MiB
    assert raw is not None
    L = capi.lib()
    h, cls = _mem_reader(L, ctx, stream)
    assert cls == capi.ERR_NONE
    got, e, biggest = _drain(L, h, 40 << 20)
    assert got == raw and e == capi.ERR_EOF and biggest > (30 << 20)
    # seek into a chunk, then large requests
    err = ctypes.c_int()
    pos = 10 * 65536 + 777
    assert L.xfg_reader_seek(h, pos, 0, ctypes.byref(err)) == pos
    got, e, _ = _drain(L, h, 40 << 20)
    assert got == raw[pos:] and e == capi.ERR_EOF
    n = L.xfg_reader_num_records(h)
    recs = (capi.Record * n)()
    L.xfg_reader_get_records(h, recs, n)
    L.xfg_reader_free(h)
    # a damaged chunk deep inside the first large request, and one at its very start
    deflate = [i for i in range(n) if recs[i].type == 1]
    for victim in (700, 0):
        i = deflate[victim]
        lo = recs[i - 1].comp_offset if i else 0
        hi = recs[i].comp_offset
        bad = bytearray(stream)
        bad[(lo + hi) // 2] ^= 0x7f
        bad = bytes(bad)
        results = []
        for req in (60000, 40 << 20):
            h, cls = _mem_reader(L, ctx, bad)
            assert cls == capi.ERR_NONE
            results.append(_drain(L, h, req)[:2])
            L.xfg_reader_free(h)
        assert results[0] == results[1]
        assert results[0][1] in (capi.ERR_CORRUPTED, capi.ERR_UNEXPECTED_EOF, capi.ERR_EOF)
        # the chunks before the damaged one are intact; what the damaged one produced before its error is the decoder's
        assert len(results[0][0]) >= victim * 65536 and results[0][0][:victim * 65536] == raw[:victim * 65536]
