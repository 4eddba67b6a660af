"""GPU compressor through the C ABI (xfg_deflate).

A DEFLATE compressor's bytes are its own: what is pinned is that every chunk is what xflate.Writer's
chunks are (a DEFLATE stream without a final block that ends in the sync marker, xflate/writer.go:169-204,
xflate/common.go:52-57) and that it decodes to the input -- checked with zlib (the reference's own second
opinion, flate/reader_test.go:861-871), with the CPU oracle, and by writing the chunks into an XFLATE
stream (index and footer from the host writer) that the oracle's xflate.Reader and the GPU reader read
back, as xflate/xflate_test.go:25-83 round-trips its corpora."""
import io
import os
import zlib

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.lib()
    return oracle


def _check_chunks(orc, raw, chunk_size, chunks):
    pieces = [raw[o:o + chunk_size] for o in range(0, len(raw), chunk_size)] or [b""]
    assert len(chunks) == len(pieces)
    for piece, c in zip(pieces, chunks):
        assert c[-4:] == b"\x00\x00\xff\xff"
        d = zlib.decompressobj(-15)
        assert d.decompress(c) == piece
        assert d.unused_data == b"" and not d.eof               # no final block: the stream goes on
        st, out, n_in, n_out = orc.inflate_chunk(c, len(piece) + 1)
        assert (st, out, n_out) == (orc.OK, piece, len(piece))


@pytest.mark.parametrize("kind", ["huffman", "repeats", "digits", "zeros", "random"])
@pytest.mark.parametrize("size,chunk", [(0, 65536), (1, 65536), (3, 65536), (4, 65536), (5, 65536), (63, 65536),
                                        (64, 65536), (65, 65536), (300, 100), (65535, 65536), (65536, 65536),
                                        (70001, 65536), (200000, 16384), (300000, 262144)])
@pytest.mark.parametrize("dynamic", [False, True])
def test_chunks_decode_to_the_input(ctx, orc, kind, size, chunk, dynamic):
    from compress_amd import synth
    raw = synth.GENERATORS[kind](size, 7 + size)
    chunks = ctx.deflate(raw, chunk, dynamic)
    _check_chunks(orc, raw, chunk, chunks)
    from compress_amd import capi
    for piece_len, c in zip([min(chunk, size - o) for o in range(0, size, chunk)] or [0], chunks):
        assert len(c) <= capi.lib().xfg_deflate_bound(piece_len)


def test_compresses(ctx):
    """Level-1 class: matches are found (repeats, zeros, text) and incompressible data falls back to stored."""
    from compress_amd import synth
    n = 1 << 18
    def ratio(raw):
        return len(raw) / sum(len(c) for c in ctx.deflate(raw, 65536))
    assert ratio(synth.GENERATORS["zeros"](n, 0)) > 50
    assert ratio(synth.GENERATORS["repeats"](n, 3)) > 2.5
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    assert ratio(twain) > 1.4
    assert ratio(synth.GENERATORS["random"](n, 5)) > 0.999 * n / (n + 5 * 4 + 5 * 4 * 2)


def test_dynamic_codes(ctx, orc):
    """XFG_DEFLATE_DYNAMIC: Huffman codes built per chunk (RFC 1951 3.2.7).  Never larger than the fixed coding of the
    same tokens, much smaller where the bytes are unevenly distributed; text, the reference's corpora kinds, skewed
    two- and one-symbol alphabets (the degenerate codes of flate/prefix.go:196-208), long runs of one length in the
    header, every chunk decoded by zlib and the oracle."""
    import random
    from compress_amd import synth
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    rng = random.Random(4)
    cases = {
        "twain": twain,
        "digits": synth.GENERATORS["digits"](300000, 1),
        "huffman": synth.GENERATORS["huffman"](300000, 2),
        "repeats": synth.GENERATORS["repeats"](300000, 3),
        "one byte": b"q",
        "two bytes": b"ab",
        "no match, two symbols": bytes(rng.choice(b"xy") for _ in range(4000)),
        "all 256 bytes once": bytes(range(256)),
        "all 256 bytes, skewed": bytes(min(255, int(rng.expovariate(0.05))) for _ in range(100000)),
        "one distance only": b"abcdefgh" * 5000,
        "fibonacci frequencies": b"".join(bytes([i]) * f for i, f in enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987,
                                                                                   1597, 2584, 4181, 6765, 10946, 17711])),
    }
    for name, raw in cases.items():
        raw = bytes(raw)
        for chunk in (65536, 4096):
            fixed = ctx.deflate(raw, chunk, False)
            dyn = ctx.deflate(raw, chunk, True)
            _check_chunks(orc, raw, chunk, dyn)
            for f, d in zip(fixed, dyn):
                assert len(d) <= len(f), name
    def total(raw, dynamic):
        return sum(len(c) for c in ctx.deflate(raw, 65536, dynamic))
    assert total(cases["digits"], True) < 0.75 * total(cases["digits"], False)
    assert total(cases["huffman"], True) < 0.97 * len(cases["huffman"])          # the fixed codes cannot shrink this kind
    assert total(twain, True) < 0.9 * total(twain, False)
    # the fibonacci case needs codes deeper than 15 bits before they are limited
    assert len(ctx.deflate(cases["fibonacci frequencies"], 65536, True)[0]) < len(cases["fibonacci frequencies"]) // 2


@pytest.mark.parametrize("seed", [226, 408, 474, 622, 1005, 1102, 7, 8, 9])
def test_dynamic_codes_fuzz_cases(ctx, seed):
    """Cases of tools/deflate_fuzz.py; the first six need the 15-bit / 7-bit length limit's repair (they came out
    over-subscribed before it counted the Kraft excess)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import deflate_fuzz
    deflate_fuzz.check(ctx, deflate_fuzz.make_pieces(seed))


def test_xflate_stream_of_gpu_chunks(ctx, orc):
    """The chunks in an XFLATE stream (host index + footer): read back by the oracle's Reader and the GPU Reader,
    front to back and after a seek into the middle."""
    from compress_amd import synth, xflate
    from oracle import xflate_oracle
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    raw = twain + synth.GENERATORS["repeats"](150000, 11) + synth.GENERATORS["digits"](70000, 12)
    chunk = 32768
    chunks = ctx.deflate(raw, chunk)
    sizes = [min(chunk, len(raw) - o) for o in range(0, len(raw), chunk)]
    stream = synth.assemble_stream(chunks, sizes, index_size=5)
    xr = xflate_oracle.XflateReader(stream)
    assert xr.err is None
    buf, err = xr.read_all()
    assert err is None and buf == raw
    rd, err = xflate.NewReader(io.BytesIO(stream), ctx=ctx)
    assert err is None
    assert rd.Read(-1) == raw
    rd.Seek(len(twain) - 100, xflate.SeekStart)
    assert rd.Read(300) == raw[len(twain) - 100:len(twain) + 200]
    rd.Close()
