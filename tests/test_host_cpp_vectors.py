"""The PRODUCT's host-side C++ (compress_amd/csrc/host_reader.cpp: meta::Reader, Index, decode_index) against the
reference's own unit-test vectors, through the test hooks of include/xflate_amd_testhooks.h.  CPU only.

  xflate/internal/meta/reader_test.go:38-452   TestReader        -> test_meta_vector
  xflate/internal/meta/reader_test.go:455-610  TestReaderReset   -> test_meta_streams, test_meta_corrupt_stream
  xflate/internal/meta/meta_test.go:210-246    TestReverseSearch -> test_reverse_search
  xflate/index_test.go:14-55                   TestIndexRoundTrip-> test_index_round_trip
  xflate/index_test.go:57-117                  TestIndex         -> test_index_append
  xflate/index_test.go:119-184                 TestIndexSearch   -> test_index_search
"""
import ctypes
import io
import os
import re

import pytest

from conftest import ROOT, load_golden
from compress_amd import capi

META = load_golden("meta_reader.json")
IDX = load_golden("index_tables.json")
ERRF = {"": capi.ERR_NONE, "IsEOF": capi.ERR_EOF, "IsUnexpectedEOF": capi.ERR_UNEXPECTED_EOF, "IsCorrupted": capi.ERR_CORRUPTED}


@pytest.fixture(scope="module")
def L():
    lib = capi.testhooks_lib()
    vp, sz, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64
    pi, psz, pi64 = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(sz), ctypes.POINTER(i64)
    lib.xfg_test_meta_block.argtypes = [vp, sz, vp, psz, pi, pi64]
    lib.xfg_test_meta_stream.argtypes = [vp, sz, vp, sz, psz, pi64, pi64, pi64, pi]
    lib.xfg_test_meta_reverse_search.argtypes = [vp, sz]
    lib.xfg_test_index_new.restype = vp
    lib.xfg_test_index_free.argtypes = [vp]
    lib.xfg_test_index_free.restype = None
    lib.xfg_test_index_append_record.argtypes = [vp, i64, i64, ctypes.c_int]
    lib.xfg_test_index_append_index.argtypes = [vp, vp]
    lib.xfg_test_index_push_raw.argtypes = [vp, i64, i64, ctypes.c_int]
    lib.xfg_test_index_push_raw.restype = None
    lib.xfg_test_index_len.argtypes = [vp]
    lib.xfg_test_index_len.restype = i64
    lib.xfg_test_index_record.argtypes = [vp, i64, vp]
    lib.xfg_test_index_record.restype = None
    lib.xfg_test_index_last_record.argtypes = [vp, vp]
    lib.xfg_test_index_last_record.restype = None
    lib.xfg_test_index_search.argtypes = [vp, i64]
    lib.xfg_test_index_search.restype = i64
    lib.xfg_test_index_get_records.argtypes = [vp, i64, vp, vp]
    lib.xfg_test_index_get_records.restype = None
    lib.xfg_test_decode_index.argtypes = [vp, sz, vp, pi64]
    return lib


def test_hooks_header_matches_exports(L):
    header = open(os.path.join(ROOT, "include", "xflate_amd_testhooks.h")).read()
    shipped = capi.lib()
    names = set(re.findall(r"\b(xfg_test_[a-z_0-9]+)\s*\(", header))
    assert len(names) >= 10
    for name in names:
        assert getattr(L, name) is not None
        assert not hasattr(shipped, name), "%s is exported by the shipped library" % name


# ------------------------------------------------------------------ meta
@pytest.mark.parametrize("i", range(len(META["vectors"])))
def test_meta_vector(L, i):
    v = META["vectors"][i]
    inp = bytes.fromhex(v["input"])
    out = ctypes.create_string_buffer(64)
    n, fin, off = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int64()
    e = L.xfg_test_meta_block(inp, len(inp), out, ctypes.byref(n), ctypes.byref(fin), ctypes.byref(off))
    assert out.raw[:n.value].hex() == v["output"], v["desc"]
    if e == capi.ERR_NONE:
        assert off.value == len(inp), v["desc"]
    assert fin.value == v["final"], v["desc"]
    assert e == ERRF[v["errf"]], (v["desc"], e)


def test_meta_streams(L):
    data = bytes.fromhex(META["streams_input"])
    pos = 0
    for v in META["streams"]:
        out = ctypes.create_string_buffer(4096)
        n = ctypes.c_size_t()
        io_, oo, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        fm = ctypes.c_int()
        rest = data[pos:]
        e = L.xfg_test_meta_stream(rest, len(rest), out, 4096, ctypes.byref(n), ctypes.byref(io_), ctypes.byref(oo),
                                   ctypes.byref(nb), ctypes.byref(fm))
        assert e == capi.ERR_NONE
        assert out.raw[:n.value].hex() == v["data"]
        assert (io_.value, oo.value, nb.value, fm.value) == (v["inOff"], v["outOff"], v["numBlks"], v["final"])
        pos += io_.value
    assert pos == len(data)


def test_meta_corrupt_stream(L):
    # meta/reader_test.go:470-477
    n = ctypes.c_size_t()
    e = L.xfg_test_meta_stream(b"corrupt", 7, None, 0, ctypes.byref(n), None, None, None, None)
    assert e == capi.ERR_CORRUPTED


def test_reverse_search(L):
    assert L.xfg_test_meta_reverse_search(b"", 0) == -1
    footer = bytes.fromhex("0d008705000048c82a51e8ff37dbf1")
    assert L.xfg_test_meta_reverse_search(footer, len(footer)) == 0
    g = b"garbage" + footer
    assert L.xfg_test_meta_reverse_search(g, len(g)) == 7
    # the LAST magic wins
    g2 = footer + b"xx" + footer
    assert L.xfg_test_meta_reverse_search(g2, len(g2)) == len(footer) + 2


def test_meta_round_trip_every_size(L):
    """meta/meta_test.go TestRoundTrip's shape: what the (vector-pinned) Python encoder writes for every payload
    size and final mode, the C++ decoder reads back, block by block."""
    from compress_amd import meta
    import random
    rng = random.Random(7)
    for size in range(0, 200, 7):
        for final in (meta.FinalNil, meta.FinalMeta, meta.FinalStream):
            payload = bytes(rng.getrandbits(8) for _ in range(size))
            enc = meta.encode(payload, final)
            out = ctypes.create_string_buffer(max(size, 1))
            n = ctypes.c_size_t()
            io_, oo, nb = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            fm = ctypes.c_int()
            e = L.xfg_test_meta_stream(enc, len(enc), out, max(size, 1), ctypes.byref(n), ctypes.byref(io_), ctypes.byref(oo),
                                       ctypes.byref(nb), ctypes.byref(fm))
            assert e == capi.ERR_NONE
            assert out.raw[:n.value] == payload and io_.value == len(enc) and oo.value == size and fm.value == final


# ----------------------------------------------------------------- index
class _Idx(object):
    """index{Records: ...} as the Go literals of index_test.go write it: records verbatim."""
    def __init__(self, L, records=()):
        self.L = L
        self.h = L.xfg_test_index_new()
        for c, r, t in records:
            L.xfg_test_index_push_raw(self.h, c, r, t)

    def last(self):
        rec = capi.Record()
        self.L.xfg_test_index_last_record(self.h, ctypes.byref(rec))
        return (rec.comp_offset, rec.raw_offset, rec.type)

    def records(self):
        out = []
        for i in range(self.L.xfg_test_index_len(self.h)):
            rec = capi.Record()
            self.L.xfg_test_index_record(self.h, i, ctypes.byref(rec))
            out.append([rec.comp_offset, rec.raw_offset, rec.type])
        return out

    def __del__(self):
        self.L.xfg_test_index_free(self.h)


def test_index_append(L):
    idx = _Idx(L)
    assert idx.last() == (0, 0, 0)
    for c, r, ok in IDX["append_records"]:
        assert L.xfg_test_index_append_record(idx.h, c, r, 0) == (1 if ok else 0), (c, r)
    assert list(idx.last()) == IDX["last_after_records"]
    for v in IDX["append_indexes"]:
        other = _Idx(L, v["records"])
        assert L.xfg_test_index_append_index(idx.h, other.h) == (1 if v["ok"] else 0), v
    assert list(idx.last()) == IDX["last_after_indexes"]
    assert idx.records() == IDX["final"]


def test_index_search(L):
    for v in IDX["search"]:
        idx = _Idx(L, v["records"])
        for off, prev, curr in v["queries"]:
            p, c = capi.Record(), capi.Record()
            L.xfg_test_index_get_records(idx.h, L.xfg_test_index_search(idx.h, off), ctypes.byref(p), ctypes.byref(c))
            assert [p.comp_offset, p.raw_offset, p.type] == prev, (v["records"], off)
            assert [c.comp_offset, c.raw_offset, c.type] == curr, (v["records"], off)


def test_index_round_trip(L):
    """encodeIndex (the Python writer, pinned byte for byte by xflate/writer_test.go's vectors) -> the C++ decodeIndex."""
    from compress_amd import xflate
    for v in IDX["round_trip"]:
        out = io.BytesIO()
        xw = xflate.Writer(out)
        index = xflate._Index()
        index.Records = [(c, r) for c, r, _t in v["records"]]
        index.BackSize = v["back_size"]
        out.seek(0)
        out.truncate()
        xw._encode_index(index)
        enc = out.getvalue()
        assert len(enc) == index.IndexSize
        h = L.xfg_test_index_new()
        back = ctypes.c_int64()
        e = L.xfg_test_decode_index(enc, len(enc), h, ctypes.byref(back))
        # (decodeIndex rejects csize <= 4, reader.go:412: the table's {5,0},{10,1},{15,2} rows sit exactly on the boundary)
        assert e == capi.ERR_NONE, v
        assert back.value == v["back_size"]
        got = []
        for i in range(L.xfg_test_index_len(h)):
            rec = capi.Record()
            L.xfg_test_index_record(h, i, ctypes.byref(rec))
            got.append([rec.comp_offset, rec.raw_offset, rec.type])
        assert got == v["records"]
        L.xfg_test_index_free(h)
