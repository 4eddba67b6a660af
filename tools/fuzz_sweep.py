#!/usr/bin/env python3
"""One-off wider run of tests/test_gpu_parity.py::test_damaged_chunks_fuzz over many seeds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from compress_amd import capi  # noqa: E402
import test_gpu_parity as t  # noqa: E402

oracle.lib()
ctx = capi.Context(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    try:
        t.test_damaged_chunks_fuzz.__wrapped__(ctx, oracle, seed) if hasattr(t.test_damaged_chunks_fuzz, "__wrapped__") else t.test_damaged_chunks_fuzz(ctx, oracle, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED:", str(e)[:300], flush=True)
    if seed % 10 == 0:
        print("seed", seed, "done", flush=True)
print("failures:", bad)
ctx.close()

# second part: valid streams over random generator / level / strategy / size combinations
import random  # noqa: E402
import zlib  # noqa: E402
from compress_amd import synth  # noqa: E402

ctx = capi.Context(0)
rng = random.Random(lo)
bad = 0
for it in range((hi - lo) * 2):
    raws, comps = [], []
    for _ in range(rng.randrange(1, 24)):
        kind = rng.choice(["huffman", "repeats", "digits", "zeros", "random"])
        size = rng.choice([0, 1, 5, 100, 257, 258, 259, 4095, 32768, 65536, rng.randrange(1, 300000)])
        raw = synth.GENERATORS[kind](size, rng.randrange(1 << 30))
        if rng.random() < 0.3 and size > 10:                      # mixed content inside one chunk
            other = synth.GENERATORS[rng.choice(["huffman", "repeats", "digits"])](size, rng.randrange(1 << 30))
            cut = rng.randrange(size)
            raw = raw[:cut] + other[cut:]
        level = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
        strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
        raws.append(raw)
        comps.append(synth.deflate_chunk(raw, level, strat))
    stream = synth.assemble_stream(comps, [len(r) for r in raws], index_size=rng.choice([1, 3, 4096]))
    try:
        t._roundtrip(ctx, oracle, stream, b"".join(raws))
    except AssertionError as e:
        bad += 1
        print("valid-stream iteration", it, "FAILED:", str(e)[:300], flush=True)
print("valid-stream failures:", bad)
ctx.close()
