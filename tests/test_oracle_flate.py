"""Pin the C oracle (oracle/inflate_oracle.c) to the reference's flate.Reader vectors.

Mirrors flate/reader_test.go:835-874: output bytes, InputOffset, OutputOffset and
error class for all 60 vectors, plus the reference's own -zcheck cross-check
against C zlib (flate/reader_test.go:861-871).
"""
import zlib

import pytest

from conftest import load_golden

VECS = load_golden("flate_reader.json")["vectors"]
ERR = {"": 0, "IsCorrupted": 1, "IsUnexpectedEOF": 2}


@pytest.mark.parametrize("v", VECS, ids=[v["name"] for v in VECS])
def test_flate_vector(oracle_lib, v):
    st, out, in_off, out_off = oracle_lib.inflate(bytes.fromhex(v["input"]))
    assert out.hex() == v["output"]
    assert in_off == v["inIdx"]
    assert out_off == v["outIdx"]
    assert st == ERR[v["errf"]]


@pytest.mark.parametrize("v", VECS, ids=[v["name"] for v in VECS])
def test_flate_vector_zcheck(v):
    inp = bytes.fromhex(v["input"])
    try:
        z = zlib.decompressobj(-15)
        out = z.decompress(inp)
        ok = z.eof
    except zlib.error:
        out, ok = b"", False
    assert ok == (v["errf"] == "")
    if ok:
        assert out.hex() == v["output"]


def test_flate_reset_stream(oracle_lib):
    # flate/reader_test.go:956-979
    st, out, _, _ = oracle_lib.inflate(b"garbage")
    assert st == oracle_lib.CORRUPTED
    data = b"\x00\x0c\x00\xf3\xffhello, world\x01\x00\x00\xff\xff"
    st, out, i, o = oracle_lib.inflate(data)
    assert (st, out, i, o) == (oracle_lib.OK, b"hello, world", len(data), 12)


@pytest.mark.parametrize("level", [1, 6, 9])
def test_oracle_matches_zlib_on_valid_streams(oracle_lib, level):
    import random
    rng = random.Random(level)
    for n in (0, 1, 100, 5000, 70000, 300000):
        words = [bytes(rng.choices(b"abcdefgh \n", k=rng.randint(1, 9))) for _ in range(200)]
        raw = b"".join(rng.choice(words) for _ in range(n // 5 + 1))[:n]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        st, out, i, o = oracle_lib.inflate(comp, len(raw) + 16)
        assert st == oracle_lib.OK and out == raw and i == len(comp) and o == len(raw)
