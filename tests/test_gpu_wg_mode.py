"""The workgroup-per-chunk decode mode (compress_amd/csrc/inflate_wg_kernel.inc), forced with xfg_set_decode_waves:
the same golden vectors, generated streams and damaged chunks as tests/test_gpu_parity.py must come out bit-identical
(bytes, InputOffset, OutputOffset, status) with 2, 4, 5, 12 and 16 wavefronts per chunk (12 and 16: with the LDS history ring), and identical to what one
wavefront per chunk gives.  Reference shapes: flate/reader_test.go:835-859 and :981-1009 (one stream, one reader),
xflate/reader_test.go:277-301 and :568-609 (a Reader that decodes a few chunks per call)."""
import random
import zlib

import pytest

import test_gpu_parity as P
from conftest import load_golden

pytestmark = pytest.mark.gpu

FLATE = load_golden("flate_reader.json")["vectors"]
WAVES = [2, 4, 5, 12, 16]


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


@pytest.fixture(params=WAVES, ids=["w%d" % w for w in WAVES])
def waves(request):
    from compress_amd import capi
    capi.set_decode_waves(request.param)
    yield request.param
    capi.set_decode_waves(0)


def test_setting_is_checked():
    from compress_amd import capi
    L = capi.lib()
    assert L.xfg_set_decode_waves(17) == capi.XFG_E_ARG
    assert L.xfg_set_decode_waves(-1) == capi.XFG_E_ARG
    assert L.xfg_set_decode_waves(0) == capi.XFG_OK
    # by chunk count: one wavefront per chunk from half the wave slots on, a workgroup below
    assert L.xfg_decode_waves_for(1, 256) == 12
    assert L.xfg_decode_waves_for(1024, 256) in (2, 3, 4)
    assert L.xfg_decode_waves_for(1 << 14, 256) == 1
    # two wavefronts per chunk up to 1.25 chunks per wave slot (profiles/r3/wg_wave_policy.txt), one beyond
    import ctypes
    L.xfg_grid_blocks.argtypes, L.xfg_grid_blocks.restype = [ctypes.c_uint32, ctypes.c_int], ctypes.c_uint32
    slots = int(L.xfg_grid_blocks(1 << 20, 256))                # the persistent grid of the one-wave kernel: the chip's wave slots (4096)
    assert L.xfg_decode_waves_for(slots, 256) == 2
    assert L.xfg_decode_waves_for(slots * 5 // 4, 256) == 2
    assert L.xfg_decode_waves_for(slots * 5 // 4 + 1, 256) == 1
    assert L.xfg_decode_waves_for(slots // 4, 256) == 4
    assert L.xfg_decode_waves_for(slots // 12, 256) == 12
    assert L.xfg_set_decode_waves(3) == capi.XFG_OK
    assert L.xfg_decode_waves_for(1 << 14, 256) == 3
    assert L.xfg_set_decode_waves(0) == capi.XFG_OK


def test_flate_vectors(ctx, waves):
    """All 60 flate/reader_test.go vectors, one by one and as one launch."""
    for v in FLATE:
        P.test_flate_vector(ctx, v)
    P.test_flate_vectors_one_launch(ctx)


def test_xflate_vectors(ctx, waves):
    for i in range(len(P.XFL["vectors"])):
        P.test_xflate_vector(ctx, i)
    P.test_xflate_reader_reset(ctx)


def test_reader_seek_and_recursion(ctx, waves):
    if waves in (2, 16):
        P.test_xflate_reader_seek(ctx)
    P.test_recursive_reader(ctx)


@pytest.mark.parametrize("kind", ["huffman", "repeats", "digits", "zeros", "random"])
def test_generated_streams(ctx, orc, waves, kind):
    for level, strategy in [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)]:
        P.test_generated_streams_match_oracle(ctx, orc, kind, level, strategy)


def test_changing_character_and_periods(ctx, orc, waves):
    P.test_chunks_that_change_character(ctx, orc, 6, zlib.Z_DEFAULT_STRATEGY)
    P.test_chunks_that_change_character(ctx, orc, 6, zlib.Z_FIXED)
    P.test_periodic_long_matches_every_distance(ctx, orc, 9, zlib.Z_DEFAULT_STRATEGY)


def test_twain_and_mix(ctx, orc, waves):
    P.test_twain_default_chunks(ctx, orc)
    P.test_mix_stream_64k_chunks(ctx, orc)


def test_damaged_chunks(ctx, orc, waves):
    P.test_corrupted_chunks_match_oracle(ctx, orc)
    P.test_damaged_chunks_fuzz(ctx, orc, 11 + waves)


def test_large_chunks_equal_one_wave(ctx, waves):
    """A few large chunks (text at level 6 and 1, fixed codes, stored, match-bound, a run): the shape the mode is for.
    Output, offsets and status must equal the one-wavefront kernel's, chunk by chunk."""
    import os
    from compress_amd import capi, synth
    twain = open(os.path.join(P.GOLDEN, "twain.txt"), "rb").read()
    rng = random.Random(waves)
    raws = [(twain * 6)[rng.randrange(1000):][:1500000],
            synth.gen_huffman(700000, 5), synth.gen_digits(500000, 6), synth.gen_repeats(900000, 7),
            synth.gen_random(300000, 8)]
    comps = [synth.deflate_chunk(raws[0], 6), synth.deflate_chunk(raws[1], 6), synth.deflate_chunk(raws[2], 1),
             synth.deflate_chunk(raws[3], 9), synth.deflate_chunk(raws[4], 0),
             synth.deflate_chunk(raws[0][:400000], 6, zlib.Z_FIXED)]
    raws.append(raws[0][:400000])
    m_raw, m_comp = synth.match_chunk(1 << 20, 3)
    r_raw, r_comp = synth.run_chunk(600000, 9)
    raws += [m_raw, r_raw]
    comps += [m_comp, r_comp]
    comp = bytearray()
    chunks = []
    off = 0
    for raw, c in zip(raws, comps):
        chunks.append((len(comp), len(c), off, len(raw), capi.CHUNK_XFLATE_DEFLATE))
        comp += c
        comp += bytes(rng.randrange(0, 7))
        off += len(raw)
    out, res = ctx.decode(bytes(comp), chunks, off)
    capi.set_decode_waves(1)
    out1, res1 = ctx.decode(bytes(comp), chunks, off)
    capi.set_decode_waves(waves)
    for c, r, r1, raw in zip(chunks, res, res1, raws):
        assert (r.status, r.in_consumed, r.out_produced) == (r1.status, r1.in_consumed, r1.out_produced) == (capi.ST_OK, c[1] + 5, len(raw))
        assert out.raw[c[2]:c[2] + c[3]] == raw
    assert out.raw[:off] == out1.raw[:off]
