"""GPU parity tests: the HIP path, called through the C ABI, against the reference's golden
vectors and the CPU oracle on the same inputs.  Bit-exact: bytes, offsets, error class."""
import hashlib
import io
import os
import random
import zlib

import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

FLATE = load_golden("flate_reader.json")["vectors"]
XFL = load_golden("xflate_reader.json")
ERR = {"": 0, "IsCorrupted": 1, "IsUnexpectedEOF": 2}


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


# ---------------------------------------------------------------- flate.Reader vectors
@pytest.mark.parametrize("v", FLATE, ids=[v["name"] for v in FLATE])
def test_flate_vector(ctx, v):
    """flate/reader_test.go:835-859: output, InputOffset, OutputOffset, error class."""
    from compress_amd import capi
    inp = bytes.fromhex(v["input"])
    cap = max(64, len(bytes.fromhex(v["output"])) + 64)
    raw, res = ctx.decode(inp, [(0, len(inp), 0, cap, 0)], cap)
    r = res[0]
    assert raw.raw[:r.out_produced].hex() == v["output"]
    assert r.in_consumed == v["inIdx"]
    assert r.out_produced == v["outIdx"]
    assert r.status == ERR[v["errf"]], capi.ST_NAMES[r.status]


def test_flate_vectors_one_launch(ctx):
    """All 60 vectors as one batch: exercises the multi-wave/multi-block grid and unaligned comp offsets."""
    comp = bytearray()
    chunks = []
    raw_off = 0
    for v in FLATE:
        inp = bytes.fromhex(v["input"])
        cap = len(v["output"]) // 2 + 7
        chunks.append((len(comp), len(inp), raw_off, cap, 0))
        comp += inp
        raw_off += cap
    raw, res = ctx.decode(bytes(comp), chunks, raw_off)
    for v, c, r in zip(FLATE, chunks, res):
        assert raw.raw[c[2]:c[2] + r.out_produced].hex() == v["output"], v["name"]
        assert (r.in_consumed, r.out_produced, r.status) == (v["inIdx"], v["outIdx"], ERR[v["errf"]]), v["name"]


def test_flate_reader_api(ctx):
    """flate/reader_test.go:956-979 (TestReaderReset) through compress_amd.flate."""
    from compress_amd import flate, errors
    zr, err = flate.NewReader(b"garbage", ctx=ctx)
    assert err is None
    with pytest.raises(errors.Error) as ei:
        zr.Read()
    assert errors.IsCorrupted(ei.value)
    with pytest.raises(errors.Error) as ei:
        zr.Close()
    assert errors.IsCorrupted(ei.value)
    data = b"\x00\x0c\x00\xf3\xffhello, world\x01\x00\x00\xff\xff"
    zr.Reset(data, ctx)
    assert zr.Read() == b"hello, world"
    assert (zr.InputOffset, zr.OutputOffset) == (len(data), 12)
    zr.Close()


# ---------------------------------------------------------------- xflate.Reader vectors
@pytest.mark.parametrize("i", range(len(XFL["vectors"])))
def test_xflate_vector(ctx, i):
    """xflate/reader_test.go:277-301."""
    from compress_amd import xflate, errors
    v = XFL["vectors"][i]
    xr, err = xflate.NewReader(bytes.fromhex(v["input"]), ctx=ctx)
    buf = None
    if err is None:
        try:
            buf = xr.Read()
        except (errors.Error, errors.UnexpectedEOF) as e:
            err = e
    if v["errf"]:
        assert errors.IsCorrupted(err), (v["desc"], err)
    else:
        assert err is None, (v["desc"], err)
        assert buf.hex() == v["output"], v["desc"]


def test_xflate_reader_reset(ctx):
    """xflate/reader_test.go:304-365 (TestReaderReset)."""
    from compress_amd import xflate, errors
    r = {k: bytes.fromhex(h) for k, h in XFL["reset"].items()}
    xr = xflate.Reader()
    xr.Reset(r["empty"], ctx)
    assert xr.Read() == b""
    xr.Close()
    xr.Close()
    with pytest.raises(errors.Error) as ei:
        xr.Read()
    assert errors.IsClosed(ei.value)
    with pytest.raises(errors.Error) as ei:
        xr.Reset(b"garbage" + r["empty"], ctx)
    assert errors.IsCorrupted(ei.value)
    with pytest.raises(errors.Error) as ei:
        xr.Seek(0, xflate.SeekStart)
    assert errors.IsCorrupted(ei.value)
    with pytest.raises(errors.Error) as ei:
        xr.Close()
    assert errors.IsCorrupted(ei.value)
    for key in ("badData", "badSize"):
        xr.Reset(r[key], ctx)
        xr.Seek(-1, xflate.SeekEnd)
        with pytest.raises(errors.Error) as ei:
            xr.Read()
        assert errors.IsCorrupted(ei.value), key


class CountRS(io.BytesIO):
    """countReadSeeker of xflate/reader_test.go:19-28."""
    N = 0

    def read(self, n=-1):
        b = io.BytesIO.read(self, n)
        self.N += len(b)
        return b


def test_xflate_reader_seek(ctx):
    """xflate/reader_test.go:367-507 (TestReaderSeek): twain.txt, 1 KiB chunks, scripted + random
    seeks + full reverse read, and the I/O budget heuristics."""
    from compress_amd import xflate, errors
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    out = bytearray()
    xw = xflate.NewWriter(out, xflate.WriterConfig(ChunkSize=1 << 10))
    xw.Write(twain)
    xw.Close()
    rs = CountRS(bytes(out))
    xr, err = xflate.NewReader(rs, ctx=ctx)
    assert err is None
    assert rs.N <= len(twain) // 100
    rs.N = 0
    n = len(twain)
    S, C, E = xflate.SeekStart, xflate.SeekCurrent, xflate.SeekEnd
    vectors = [(40, n - 1, S), (40, n, S), (40, n + 1, S), (40, (1 << 63) - 1, S), (0, 0, C), (13, 15, S),
               (32, 23, C), (32, -23, C), (13, -15, S), (100, -15, E), (0, 0, C), (0, 0, C), (32, -34, C),
               (32, -34, C), (2000, 53, S), (2000, n - 1000, S), (0, 0, C), (100, -n, E), (100, -n - 1, E),
               (0, 0, S), (10, 10, C), (10, 10, C), (10, 10, C), (10, 10, C), (0, 0, -1)]
    rng = random.Random(0)
    for _ in range(100):
        length, offset = rng.randrange(1 << 11), rng.randrange(n)
        if length + offset <= n:
            vectors.append((length, offset, S))
    vectors.append((0, 0, E))
    pos = n
    while pos > 0:
        k = min(rng.randrange(1 << 11), pos)
        pos -= k
        vectors.append((k, pos, S))
    pos, total = 0, 0
    for i, (length, offset, whence) in enumerate(vectors):
        want_pos = {S: offset, C: offset + pos, E: offset + n}.get(whence, -1)
        try:
            got_pos = xr.Seek(offset, whence)
            failed = False
        except errors.Error as e:
            assert errors.IsInvalid(e)
            failed = True
        assert failed == (want_pos < 0), i
        if failed:
            continue
        assert got_pos == want_pos
        want = twain[want_pos:want_pos + length] if want_pos < n else b""
        got = bytearray()
        while len(got) < length:
            try:
                b = xr.Read(length - len(got))
            except errors.EOF:
                break
            got += b
        assert bytes(got) == want, i
        pos = got_pos + len(got)
        total += length
    assert rs.N <= 2 * total


def test_recursive_reader(ctx):
    """xflate/reader_test.go:509-563: 5 nested xflate layers, each Reader is the next one's ReadSeeker."""
    from compress_amd import xflate
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    data = twain
    for i in reversed(range(5)):
        out = bytearray()
        xw = xflate.NewWriter(out, xflate.WriterConfig(ChunkSize=1 << (10 + i)))
        xw.Write(data)
        xw.Close()
        data = bytes(out)

    class RS(object):
        def __init__(self, xr):
            self.xr = xr

        def read(self, n):
            from compress_amd import errors
            try:
                return self.xr.Read(n)
            except errors.EOF:
                return b""

        def seek(self, off, whence):
            return self.xr.Seek(off, whence)

    rlast = io.BytesIO(data)
    readers = []
    for i in range(5):
        xr, err = xflate.NewReader(rlast, ctx=ctx)
        assert err is None
        readers.append(xr)
        rlast = RS(xr)
    rlast.seek(len(twain) // 2, 0)
    got = bytearray()
    while len(got) < 321:
        got += rlast.read(321 - len(got))
    assert bytes(got) == twain[len(twain) // 2:][:321]
    for xr in reversed(readers):
        xr.Close()


# ---------------------------------------------------------------- oracle parity on generated streams
def _roundtrip(ctx, orc, stream, raw):
    """Decode every record of `stream` on the GPU in one launch and compare, chunk by chunk, with the
    CPU oracle fed the same bytes: status, in_consumed, out_produced and output bytes."""
    from compress_amd import capi
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    descs = [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks]
    raw_len = sum(c.raw_size for c in chunks)
    out, res = ctx.decode(stream, descs, raw_len)
    for c, r in zip(chunks, res):
        comp = stream[c.comp_off:c.comp_off + c.comp_size]
        st, o_out, o_in, o_on = orc.inflate_chunk(comp, c.raw_size + 1)
        assert st == orc.OK
        assert r.status == capi.ST_OK, (capi.ST_NAMES[r.status], c.comp_off)
        want_in = o_in if not (c.flags & capi.CHUNK_FOOTER) else c.comp_size
        assert (r.in_consumed, r.out_produced) == (want_in, o_on)
        assert out.raw[c.raw_off:c.raw_off + c.raw_size] == o_out
    if raw is not None:
        assert out.raw[:raw_len] == raw
    return out.raw[:raw_len]


@pytest.mark.parametrize("kind", ["huffman", "repeats", "digits", "zeros", "random"])
@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY),
                                           (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)])
def test_generated_streams_match_oracle(ctx, orc, kind, level, strategy):
    """xflate/xflate_test.go:25-83 (TestRoundTrip) shape: every corpus kind, dynamic / fixed / stored blocks,
    ragged chunk sizes including empty and 1-byte chunks."""
    from compress_amd import synth
    sizes = [0, 1, 2, 3, 64, 257, 258, 259, 1000, 4096, 32768, 32769, 65536, 70001, 150000]
    raws = [synth.GENERATORS[kind](n, 17 + i) for i, n in enumerate(sizes)]
    comps = [synth.deflate_chunk(r, level, strategy) for r in raws]
    stream = synth.assemble_stream(comps, [len(r) for r in raws], index_size=4)
    _roundtrip(ctx, orc, stream, b"".join(raws))


@pytest.mark.parametrize("level,strategy", [(6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)])
def test_chunks_that_change_character(ctx, orc, level, strategy):
    """The kernel picks its segment size (208 / 160 / 96 bits, 128 for runs) per batch from the expansion of the batch
    before it.  These chunks switch between literal-heavy, mixed and match-heavy data every few KB, also
    inside one DEFLATE block, so that every transition between sizes is taken; results must not depend on it."""
    from compress_amd import synth
    rng = random.Random(level * 31 + strategy)
    raws = []
    for i in range(24):
        parts = []
        while sum(map(len, parts)) < 40000 + 9000 * (i % 5):
            kind = rng.choice(["huffman", "repeats", "digits", "zeros", "random", "huffman", "digits"])
            parts.append(synth.GENERATORS[kind](rng.choice([700, 2000, 5000, 12000, 30000]), 1000 * i + len(parts)))
        raws.append(b"".join(parts))
    comps = [synth.deflate_chunk(r, level, strategy) for r in raws]
    stream = synth.assemble_stream(comps, [len(r) for r in raws], index_size=7)
    _roundtrip(ctx, orc, stream, b"".join(raws))


@pytest.mark.parametrize("level,strategy", [(9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)])
def test_periodic_long_matches_every_distance(ctx, orc, level, strategy):
    """dictDecoder's overlapping forward copy (flate/dict_decoder.go:93-100) for every period: a pattern of p bytes
    repeated for 70 to 600 bytes is one literal run and matches of distance p, most of them long ones (which the
    whole wave copies, byte j from source byte j mod p), for every p from 1 to 300, at every offset of the
    window, sources inside and before it; a second pass uses periods up to 32768."""
    rng = random.Random(level)
    raws = []
    for rep in range(3):
        out = bytearray()
        for p_ in list(range(1, 301)) + [rng.randrange(301, 32769) for _ in range(20)]:
            pat = bytes(rng.randrange(256) for _ in range(min(p_, 4096)))
            if p_ > len(pat):
                pat = pat + bytes(rng.randrange(256) for _ in range(p_ - len(pat)))
            n = rng.randrange(70, 600) + p_
            out += (pat * (n // p_ + 1))[:n]
            out += bytes(rng.randrange(256) for _ in range(rng.randrange(0, 9)))
        raws.append(bytes(out))
    comps = [synth_deflate(r, level, strategy) for r in raws]
    from compress_amd import synth
    stream = synth.assemble_stream(comps, [len(r) for r in raws], index_size=2)
    _roundtrip(ctx, orc, stream, b"".join(raws))


def synth_deflate(raw, level, strategy):
    from compress_amd import synth
    return synth.deflate_chunk(raw, level, strategy)


def test_twain_default_chunks(ctx, orc):
    """BASELINE config C1: twain.txt, default 256 KiB chunk size (chunks of 262144 + 125825 bytes)."""
    from compress_amd import xflate, capi
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    out = bytearray()
    xw = xflate.NewWriter(out)
    xw.Write(twain)
    xw.Close()
    err, chunks = capi.index_parse(bytes(out))
    assert [c.raw_size for c in chunks if c.flags & capi.CHUNK_SYNC] == [262144, 125825]
    _roundtrip(ctx, orc, bytes(out), twain)
    xr, err = xflate.NewReader(bytes(out), ctx=ctx)
    assert err is None and xr.Read() == twain


def test_mix_stream_64k_chunks(ctx, orc):
    """BASELINE config C2 at 1/64 scale: 256 x 64 KiB chunks, huffman/repeats/digits mix, level 6."""
    from compress_amd import synth
    stream, raw = synth.build_mix_stream(256, 65536)
    got = _roundtrip(ctx, orc, stream, raw)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(raw).hexdigest()


def test_corrupted_chunks_match_oracle(ctx, orc):
    """Bit flips and truncations inside chunks: the GPU must report the oracle's status, offsets and
    the bytes produced before the error for every damaged chunk."""
    from compress_amd import synth, capi
    rng = random.Random(5)
    raws = [synth.gen_huffman(20000, 3), synth.gen_repeats(20000, 4), synth.gen_digits(20000, 5)]
    comp = bytearray()
    chunks = []
    cap = 40000
    for t in range(120):
        c = bytearray(synth.deflate_chunk(raws[t % 3], 6))
        mode = t % 4
        if mode == 0:
            c[rng.randrange(len(c))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            c = c[:rng.randrange(1, len(c))]
        elif mode == 2:
            for _ in range(3):
                c[rng.randrange(len(c))] = rng.randrange(256)
        chunks.append((len(comp), len(c), t * cap, cap, capi.CHUNK_END_BLOCK | capi.CHUNK_RFC_REP16))
        comp += c
        comp += bytes(rng.randrange(0, 4))          # ragged, unaligned chunk starts
    out, res = ctx.decode(bytes(comp), chunks, cap * len(chunks))
    for c, r in zip(chunks, res):
        st, o_out, o_in, o_on = orc.inflate_chunk(bytes(comp[c[0]:c[0] + c[1]]), cap)
        want = {orc.OK: capi.ST_OK, orc.CORRUPTED: capi.ST_CORRUPTED, orc.UNEXPECTED_EOF: capi.ST_UNEXPECTED_EOF,
                orc.OUT_FULL: capi.ST_OUT_FULL}[st]
        assert r.status == want, (capi.ST_NAMES[r.status], capi.ST_NAMES[want], c)
        assert r.out_produced == o_on
        assert out.raw[c[2]:c[2] + o_on] == o_out
        if st != orc.OUT_FULL:
            assert r.in_consumed == o_in


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_damaged_chunks_fuzz(ctx, orc, seed):
    """A wider sweep of damage than test_corrupted_chunks_match_oracle: every block type (stored, fixed,
    dynamic, token-written match chunks), bit flips, byte splices, truncations and garbage tails, with
    and without the appended endBlock.  Status, offsets and the bytes before the error must equal the oracle's."""
    from compress_amd import synth, capi
    rng = random.Random(seed)
    sources = [
        synth.deflate_chunk(synth.gen_huffman(30000, seed), 6),
        synth.deflate_chunk(synth.gen_repeats(30000, seed + 1), 9),
        synth.deflate_chunk(synth.gen_digits(30000, seed + 2), 1),
        synth.deflate_chunk(synth.gen_digits(9000, seed + 3), 6, zlib.Z_FIXED),
        synth.deflate_chunk(synth.gen_random(20000, seed + 4), 0),
        synth.match_chunk(40000, seed + 5)[1],
        synth.run_chunk(33000, 7, False)[1],
        synth.deflate_chunk(synth.gen_huffman(700, seed + 6), 6),
    ]
    comp = bytearray()
    chunks = []
    cap = 48000
    for t in range(400):
        c = bytearray(sources[t % len(sources)])
        mode = rng.randrange(6)
        if mode == 0:
            c[rng.randrange(len(c))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            c = c[:rng.randrange(1, len(c))]
        elif mode == 2:
            for _ in range(rng.randrange(1, 6)):
                c[rng.randrange(len(c))] = rng.randrange(256)
        elif mode == 3:
            p = rng.randrange(len(c))
            c[p:p + rng.randrange(1, 40)] = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40)))
        elif mode == 4:
            c = c[:rng.randrange(1, min(len(c), 400))]          # damage inside the first block header
            c[rng.randrange(len(c))] ^= 0xff
        else:
            c += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 30)))
        flags = rng.choice([0, capi.CHUNK_END_BLOCK, capi.CHUNK_END_BLOCK | capi.CHUNK_RFC_REP16, capi.CHUNK_RFC_REP16])
        chunks.append((len(comp), len(c), t * cap, cap, flags))
        comp += c
        comp += bytes(rng.randrange(0, 4))
    out, res = ctx.decode(bytes(comp), chunks, cap * len(chunks))
    for c, r in zip(chunks, res):
        data = bytes(comp[c[0]:c[0] + c[1]]) + (b"\x01\x00\x00\xff\xff" if c[4] & capi.CHUNK_END_BLOCK else b"")
        st, o_out, o_in, o_on = orc.inflate(data, cap, rfc_rep16=bool(c[4] & capi.CHUNK_RFC_REP16))
        want = {orc.OK: capi.ST_OK, orc.CORRUPTED: capi.ST_CORRUPTED, orc.UNEXPECTED_EOF: capi.ST_UNEXPECTED_EOF,
                orc.OUT_FULL: capi.ST_OUT_FULL}[st]
        assert r.status == want, (capi.ST_NAMES[r.status], capi.ST_NAMES[want], c)
        assert r.out_produced == o_on, c
        assert out.raw[c[2]:c[2] + o_on] == o_out, c
        if st != orc.OUT_FULL:
            assert r.in_consumed == o_in, c


def test_reader_streams_a_large_file(ctx):
    """xflate.Reader over a 48 MiB stream read front to back in 1 MiB pieces, then a seek into the middle:
    the read-ahead window grows to large launches and the bytes stay those of the generator."""
    from compress_amd import xflate, synth, errors
    stream, _ = synth.build_mix_stream(768, 65536, n_distinct=96)
    made = [synth.GENERATORS[k](65536, 1 + 1000003 * (i % 3) + i) for i, k in zip(range(96), ["huffman", "repeats", "digits"] * 32)]
    want = hashlib.sha256()
    for i in range(768):
        want.update(made[i % 96])
    xr, err = xflate.NewReader(io.BytesIO(stream), ctx=ctx)
    assert err is None
    got, total = hashlib.sha256(), 0
    while True:
        try:
            b = xr.Read(1 << 20)
        except errors.EOF:
            break
        got.update(b)
        total += len(b)
    assert total == 768 * 65536 and got.hexdigest() == want.hexdigest()
    pos = 300 * 65536 + 12345
    assert xr.Seek(pos, xflate.SeekStart) == pos
    b = bytearray()
    while len(b) < 100000:
        b += xr.Read(100000 - len(b))
    ref = b"".join(made[i % 96] for i in range(300, 303))
    assert bytes(b) == ref[12345:12345 + 100000]
    xr.Close()
