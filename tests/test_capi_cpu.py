"""CPU-only checks of the C ABI: the library loads, exports every symbol that
include/xflate_amd.h declares, and its host-only entry points (index parsing,
argument validation) behave like the reference.  No compute call is made."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, load_golden
from compress_amd import capi
from oracle import xflate_oracle as xo

XFL = load_golden("xflate_reader.json")


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    header = open(os.path.join(ROOT, "include", "xflate_amd.h")).read()
    declared = set(re.findall(r"\b(xfg_[a-z_0-9]+)\s*\(", header))
    declared -= {"xfg_read_fn", "xfg_seek_fn"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in sorted(declared):
        assert getattr(L, name) is not None


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Chunk) == 40
    assert ctypes.sizeof(capi.Result) == 24
    assert ctypes.sizeof(capi.Record) == 24


def test_strerror():
    L = capi.lib()
    assert L.xfg_strerror(0) == b"ok"
    assert L.xfg_strerror(5)


@pytest.mark.parametrize("i", range(len(XFL["vectors"])))
def test_index_parse_matches_oracle(i):
    """decodeFooter/decodeIndexes of the C++ host layer vs. the Python oracle on every
    xflate/reader_test.go vector: same error class, same merged record table."""
    v = XFL["vectors"][i]
    stream = bytes.fromhex(v["input"])
    err, chunks = capi.index_parse(stream)
    xr = xo.XflateReader(stream)
    if xr.err is not None:
        assert err == capi.ERR_CORRUPTED, v["desc"]
        return
    assert err == capi.ERR_NONE, v["desc"]
    assert len(chunks) == len(xr.idx.records)
    pc = pr = 0
    # index and footer records carry HOST_META: the host decoded their meta blocks while parsing
    flags = {xo.DEFLATE_TYPE: capi.CHUNK_XFLATE_DEFLATE, xo.INDEX_TYPE: capi.CHUNK_XFLATE_INDEX | capi.CHUNK_HOST_META,
             xo.FOOTER_TYPE: capi.CHUNK_XFLATE_FOOTER | capi.CHUNK_HOST_META}
    for ch, (c, r, t) in zip(chunks, xr.idx.records):
        assert (ch.comp_off, ch.comp_size, ch.raw_off, ch.raw_size, ch.flags) == (pc, c - pc, pr, r - pr, flags[t])
        pc, pr = c, r


def test_index_parse_large_synthetic():
    from compress_amd import synth
    stream, raw = synth.build_mix_stream(40, 3000, index_size=7)
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    xr = xo.XflateReader(stream)
    assert len(chunks) == len(xr.idx.records)
    assert sum(c.raw_size for c in chunks) == len(raw)
    assert chunks[-1].comp_off + chunks[-1].comp_size == len(stream)


def test_crc32_combine_matches_zlib():
    """xfg_crc32_combine (host only) against C zlib: hash/crc32.IEEE is the same polynomial
    (xflate/reader.go:386, xflate/example_test.go:171)."""
    import random
    import zlib
    rng = random.Random(4)
    for _ in range(50):
        a = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 300)))
        b = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 2, 255, 256, 1000, 70000])))
        assert capi.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
