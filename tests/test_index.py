"""index semantics, restating xflate/index_test.go:14-184 against the oracle's Index
(the product's C++ Index is exercised through xfg_index_parse / the Reader tests)."""
from oracle import xflate_oracle as xo

MAX = (1 << 63) - 1


def test_append_record_overflow_rules():
    # index_test.go TestIndexAppendRecord
    idx = xo.Index()
    assert idx.append_record(1, 1, xo.DEFLATE_TYPE)
    assert not idx.append_record(-1, 0, xo.DEFLATE_TYPE)
    assert not idx.append_record(0, -1, xo.DEFLATE_TYPE)
    assert not idx.append_record(MAX, 0, xo.DEFLATE_TYPE)       # 1 + MaxInt64 overflows
    assert not idx.append_record(0, MAX, xo.DEFLATE_TYPE)
    assert idx.append_record(MAX - 1, MAX - 1, xo.DEFLATE_TYPE)
    assert idx.records == [(1, 1, 1), (MAX, MAX, 1)]


def test_append_index_is_atomic():
    # index_test.go TestIndexAppendIndex
    a = xo.Index()
    a.append_record(10, 20, 1)
    b = xo.Index()
    b.append_record(5, 5, 1)
    b.append_record(MAX - 10, 0, 1)                              # fits in b, overflows once appended to a
    before = list(a.records)
    assert not a.append_index(b)
    assert a.records == before
    c = xo.Index()
    c.append_record(5, 5, 1)
    c.append_record(5, 0, 2)
    assert a.append_index(c)
    assert a.records == [(10, 20, 1), (15, 25, 1), (20, 25, 2)]


def test_search_and_get_records():
    # index_test.go TestIndexSearch: duplicate RawOffsets pick the LAST record at that offset
    idx = xo.Index()
    for c, r in [(1, 1), (2, 2), (3, 2), (4, 2), (5, 5), (6, 5), (7, 10)]:
        idx.append_record(c - idx.last_record()[0], r - idx.last_record()[1], 1)
    assert idx.search(0) == 0
    assert idx.search(1) == 1
    assert idx.search(2) == 4
    assert idx.search(4) == 4
    assert idx.search(5) == 6
    assert idx.search(9) == 6
    assert idx.search(10) == 7
    assert idx.search(1000) == 7
    prev, curr = idx.get_records(0)
    assert prev == (0, 0, 0) and curr == (1, 1, 1)
    prev, curr = idx.get_records(7)
    assert prev == (7, 10, 1) and curr == (7, 10, 0)            # past the end: unknownType
    prev, curr = idx.get_records(99)
    assert prev == (7, 10, 1) and curr[2] == xo.UNKNOWN_TYPE
    prev, curr = idx.get_records(-1)
    assert prev == (0, 0, 0) and curr[2] == xo.UNKNOWN_TYPE


# ---- the reference's own tables (xflate/index_test.go:57-184), from tests/golden/index_tables.json
def _literal(records):
    idx = xo.Index()
    idx.records = [tuple(r) for r in records]
    return idx


def test_reference_tables_append():
    from conftest import load_golden
    T = load_golden("index_tables.json")
    idx = xo.Index()
    assert tuple(idx.last_record()) == (0, 0, 0)
    for c, r, ok in T["append_records"]:
        assert idx.append_record(c, r, 0) == ok
    assert list(idx.last_record()) == T["last_after_records"]
    for v in T["append_indexes"]:
        assert idx.append_index(_literal(v["records"])) == v["ok"]
    assert list(idx.last_record()) == T["last_after_indexes"]
    assert [list(r) for r in idx.records] == T["final"]


def test_reference_tables_search():
    from conftest import load_golden
    T = load_golden("index_tables.json")
    for v in T["search"]:
        idx = _literal(v["records"])
        for off, prev, curr in v["queries"]:
            p, c = idx.get_records(idx.search(off))
            assert list(p) == prev and list(c) == curr, (v["records"], off)
