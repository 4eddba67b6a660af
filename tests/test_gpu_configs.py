"""GPU parity on the shapes of BASELINE.json configs C3, C4 and C5 (C1 and C2 are in
test_gpu_parity.py): 256 KiB chunks, match-copy-bound 1 MiB chunks, and the chunk-size x
block-type sweep.  Every chunk is compared with the generator's bytes and with the CPU oracle
(status, consumed, produced)."""
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


def _check(ctx, orc, comps, raws, index_size=4096):
    from compress_amd import capi, synth
    stream = synth.assemble_stream(comps, [len(r) for r in raws], index_size=index_size)
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    descs = [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks]
    raw_len = sum(c.raw_size for c in chunks)
    out, res = ctx.decode(stream, descs, raw_len)
    data = [c for c in chunks if c.flags & capi.CHUNK_SYNC]
    assert len(data) == len(raws)
    for c, r in zip(chunks, res):
        assert r.status == capi.ST_OK, (capi.ST_NAMES[r.status], c.comp_off, c.comp_size)
        assert r.out_produced == c.raw_size
    for c, raw in zip(data, raws):
        assert out.raw[c.raw_off:c.raw_off + c.raw_size] == raw
    # the oracle on a sample of the chunks (all of them when there are few)
    step = max(1, len(data) // 16)
    for c, r in list(zip(chunks, res))[::step]:
        st, o_out, o_in, o_on = orc.inflate_chunk(stream[c.comp_off:c.comp_off + c.comp_size], c.raw_size + 1)
        assert st == orc.OK
        want_in = o_in if not (c.flags & capi.CHUNK_FOOTER) else c.comp_size
        assert (r.in_consumed, r.out_produced) == (want_in, o_on)


def test_c3_256k_chunks(ctx, orc):
    """C3 shape: 256 KiB chunks of the huffman/repeats/digits mix, several indexes."""
    from compress_amd import synth
    raws = [synth.GENERATORS[k](1 << 18, 100 + i) for i, k in enumerate(["huffman", "repeats", "digits"] * 8)]
    comps = [synth.deflate_chunk(r, 6) for r in raws]
    _check(ctx, orc, comps, raws, index_size=5)


@pytest.mark.parametrize("dynamic", [True, False])
def test_c4_match_bound_1m_chunks(ctx, orc, dynamic):
    """C4: 1 MiB chunks of 258-byte matches at 1-32 KiB distances, the dist-1 run, and zeros through zlib."""
    from compress_amd import synth
    made = [synth.match_chunk(1 << 20, 1, dynamic=dynamic),
            synth.match_chunk(1 << 20, 2, dists=(32768,), dynamic=dynamic),
            synth.match_chunk(1 << 20, 3, dists=(1, 2, 3, 5, 7, 16, 257, 258, 259), dynamic=dynamic),
            synth.match_chunk((1 << 20) - 77, 4, dists=(1024, 32768, 4096), dynamic=dynamic),
            synth.run_chunk(1 << 20, 0, dynamic),
            synth.run_chunk((1 << 20) + 1, 0xA5, dynamic)]
    raws = [m[0] for m in made] + [bytes(1 << 20), synth.gen_repeats(1 << 20, 9)]
    comps = [m[1] for m in made] + [synth.deflate_chunk(r, 6) for r in raws[-2:]]
    _check(ctx, orc, comps, raws)


@pytest.mark.parametrize("kind", ["stored", "fixed", "dynamic"])
def test_c5_chunk_size_sweep(ctx, orc, kind):
    """C5: chunk sizes 16 KiB - 4 MiB x {stored (random bytes, level 0), fixed (Z_FIXED), dynamic (level 6)}."""
    from compress_amd import synth
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    text = twain * 11
    raws, comps = [], []
    for i, size in enumerate([16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20]):
        if kind == "stored":
            raw = synth.gen_random(size, 40 + i)
            comp = synth.deflate_chunk(raw, 0)
        elif kind == "fixed":
            raw = text[i * 1000:i * 1000 + size]
            comp = synth.deflate_chunk(raw, 6, zlib.Z_FIXED)
        else:
            raw = text[i * 777:i * 777 + size]
            comp = synth.deflate_chunk(raw, 6)
        raws.append(raw)
        comps.append(comp)
    _check(ctx, orc, comps, raws)
