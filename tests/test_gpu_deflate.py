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
def test_chunks_decode_to_the_input(ctx, orc, kind, size, chunk):
    from compress_amd import synth
    raw = synth.GENERATORS[kind](size, 7 + size)
    chunks = ctx.deflate(raw, chunk)
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
