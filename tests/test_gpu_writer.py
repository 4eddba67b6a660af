"""xflate.Writer driving the GPU compressor (GpuWriter -> xfg_deflate): the chunker, flush modes, index and footer of
xflate/writer.go:138-298 over device-compressed chunks.  A compressor's bytes are its own, so the pin is what the
readers make of the stream: the oracle's xflate.Reader and the GPU Reader must both give the input back, the records
must describe the chunks the Writer's configuration asks for, and index/footer bytes must be what the zlib-backed
Writer (pinned byte-for-byte by xflate/writer_test.go in tests/test_writer.py) emits for the same records."""
import io
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _oracle_read(stream):
    from oracle import xflate_oracle
    xr = xflate_oracle.XflateReader(stream)
    assert xr.err is None, xr.err
    buf, err = xr.read_all()
    assert err is None, err
    return buf


def _gpu_read(stream, ctx):
    from compress_amd import xflate
    xr, err = xflate.NewReader(io.BytesIO(stream), ctx=ctx)
    assert err is None, err
    out = xr.Read(-1)
    recs = xr.records()
    xr.Close()
    return out, recs


def _data(n, seed):
    from compress_amd import synth
    kinds = [synth.gen_huffman, synth.gen_repeats, synth.gen_digits, synth.gen_random]
    rng = random.Random(seed)
    out = bytearray()
    while len(out) < n:
        k = min(n - len(out), rng.randrange(1, 70000))
        out += kinds[rng.randrange(4)](k, seed + len(out))
    return bytes(out[:n])


@pytest.mark.parametrize("size,chunk,index,batch", [
    (0, 1 << 16, 0, 1024), (1, 1 << 16, 0, 1024), (65536, 1 << 16, 0, 1024), (65537, 1 << 16, 0, 1024),
    (300000, 1 << 12, 7, 5), (1 << 20, 1 << 16, 3, 2), (5 << 20, 1 << 18, 0, 1024), (777777, 1000, 64, 100),
])
def test_round_trip(ctx, size, chunk, index, batch):
    from compress_amd import xflate
    data = _data(size, size + chunk)
    out = bytearray()
    xw = xflate.GpuWriter(out, xflate.WriterConfig(ChunkSize=chunk, IndexSize=index), ctx=ctx, batch_chunks=batch)
    rng = random.Random(size)
    pos = 0
    while pos < len(data):                       # ragged writes
        k = min(len(data) - pos, rng.randrange(1, 3 * chunk))
        assert xw.Write(data[pos:pos + k]) == k
        pos += k
    xw.Close()
    xw.Close()                                   # idempotent (writer.go:210-212)
    assert xw.InputOffset == len(data) and xw.OutputOffset == len(out)
    stream = bytes(out)
    got, recs = _gpu_read(stream, ctx)
    assert got == data
    assert _oracle_read(stream) == data
    # the chunker: every chunk but the last holds ChunkSize bytes
    deflate = [r for r in recs if r[2] == 1]
    raws = [b[1] - a[1] for a, b in zip([(0, 0, 0)] + recs[:-1], recs) if b[2] == 1]
    assert len(deflate) == (len(data) + chunk - 1) // chunk
    assert all(r == chunk for r in raws[:-1]) and sum(raws) == len(data)


def test_flush_modes(ctx):
    """FlushSync inside a chunk, FlushFull ending chunks early, FlushIndex in the middle (writer.go:169-204)."""
    from compress_amd import xflate
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    out = bytearray()
    xw = xflate.GpuWriter(out, xflate.WriterConfig(ChunkSize=1 << 15), ctx=ctx, batch_chunks=3)
    xw.Write(twain[:1000])
    xw.Flush(xflate.FlushSync)                   # what is written so far is decodable
    n_sync = len(out)
    assert n_sync > 0 and bytes(out[-4:]) == b"\x00\x00\xff\xff"
    xw.Flush(xflate.FlushSync)                   # nothing new: the empty stored block alone
    assert len(out) == n_sync + 5 and bytes(out[-5:]) == b"\x00\x00\x00\xff\xff"
    xw.Write(twain[1000:5000])
    xw.Flush(xflate.FlushFull)                   # a short chunk
    xw.Write(twain[5000:200000])
    xw.Flush(xflate.FlushIndex)
    xw.Write(twain[200000:])
    xw.Flush(xflate.FlushFull)
    xw.Flush(xflate.FlushFull)                   # an empty chunk: its sync marker only
    xw.Close()
    with pytest.raises(Exception):
        xw.Write(b"x")
    stream = bytes(out)
    got, recs = _gpu_read(stream, ctx)
    assert got == twain
    assert _oracle_read(stream) == twain
    raws = [b[1] - a[1] for a, b in zip([(0, 0, 0)] + recs[:-1], recs) if b[2] == 1]
    assert raws[0] == 5000 and raws[-1] == 0 and sum(raws) == len(twain)
    assert sum(1 for r in recs if r[2] == 2) == 2        # two indexes


def test_index_and_footer_bytes_match_the_host_writer(ctx):
    """Same records -> same index and footer bytes as the zlib-backed Writer, whose encoders the reference's vectors pin."""
    from compress_amd import xflate, synth
    data = _data(200000, 9)
    out = bytearray()
    xw = xflate.GpuWriter(out, xflate.WriterConfig(ChunkSize=1 << 14, IndexSize=5), ctx=ctx)
    xw.Write(data)
    xw.Close()
    stream = bytes(out)
    _, recs = _gpu_read(stream, ctx)
    # rebuild the stream around the GPU's chunks with the host writer's assembler
    chunks, raws, pc, pr = [], [], 0, 0
    for c, r, t in recs:
        if t == 1:
            chunks.append(stream[pc:c])
            raws.append(r - pr)
        pc, pr = c, r
    assert synth.assemble_stream(chunks, raws, index_size=5) == stream


def test_unknown_compressor():
    from compress_amd import xflate, errors
    with pytest.raises(errors.Error):
        xflate.NewWriter(bytearray(), None, compressor="lz4")
