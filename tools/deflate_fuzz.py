#!/usr/bin/env python3
"""Wider run of the GPU compressor's dynamic-code path: random alphabets (size, skew), sizes and match densities, every
chunk decoded by zlib and compared with its input; also checks dynamic <= fixed per chunk.
usage: tools/deflate_fuzz.py <first seed> <last seed>"""
import os
import random
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from compress_amd import capi  # noqa: E402


def make_pieces(seed):
    """The chunks of one fuzz case."""
    rng = random.Random(seed)
    pieces = []
    for _ in range(rng.randrange(1, 40)):
        size = rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 1000, 4096, 65535, 65536, rng.randrange(1, 200000)])
        nsym = rng.choice([1, 2, 3, 4, 16, 17, 30, 64, 137, 138, 139, 200, 256])
        alpha = rng.sample(range(256), nsym)
        skew = rng.choice([0.0, 0.5, 1.0, 2.0, 4.0])
        weights = [1.0 / (i + 1) ** skew for i in range(nsym)]
        data = bytearray(rng.choices(alpha, weights, k=size))
        # sprinkle copies of earlier stretches (matches at assorted distances)
        for _ in range(rng.randrange(0, 1 + size // 50)):
            if size < 8:
                break
            n = rng.randrange(4, min(300, size))
            src = rng.randrange(0, size - n)
            dst = rng.randrange(0, size - n)
            data[dst:dst + n] = data[src:src + n]
        pieces.append(bytes(data))
    return pieces


def check(ctx, pieces):
    dyn = ctx.deflate_chunks(pieces, True)
    fix = ctx.deflate_chunks(pieces, False)
    for i, (p, d, f) in enumerate(zip(pieces, dyn, fix)):
        z = zlib.decompressobj(-15)
        out = z.decompress(d)
        assert out == p and not z.eof and z.unused_data == b"", "chunk %d of %d bytes decodes wrong" % (i, len(p))
        assert d[-4:] == b"\x00\x00\xff\xff"
        assert len(d) <= len(f), "chunk %d: dynamic %d > fixed %d" % (i, len(d), len(f))
        assert zlib.decompressobj(-15).decompress(f) == p


def main():
    ctx = capi.Context(0)
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for seed in range(lo, hi):
        try:
            check(ctx, make_pieces(seed))
        except (AssertionError, zlib.error, capi.TransportError) as e:
            bad += 1
            print("seed", seed, "FAILED:", str(e)[:200], flush=True)
        if seed % 50 == 0:
            print("seed", seed, "done", flush=True)
    print("failures:", bad)
    ctx.close()


if __name__ == "__main__":
    main()
