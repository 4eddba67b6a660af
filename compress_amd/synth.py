"""Synthetic corpora and XFLATE streams for tests and bench.py.

The reference's testdata generators use Go's math/rand with seed 0
(testdata/huffman.go:35, repeats.go:26), whose stream cannot be reproduced
without Go, so the DISTRIBUTIONS are restated here over numpy's PCG64 with
stated seeds; the bytes are not identical to testdata/*.bin.

  huffman  testdata/huffman.go:37-75   segments of 16-79 symbols: 75 % base64-ish
                                       (10/60/30 % lower/letters/base64), 25 % UTF-8 2-/3-byte
  repeats  testdata/repeats.go:28-118  10 % random runs, 80 % long-distance copies
                                       (len 4-511, dist 1-32767 log-bucketed), 10 % short/overlapping
  digits   uniform ASCII '0'-'9' (the statistics of testdata/digits.txt)
  zeros    testdata/zeros.go           all zero bytes
  random   testdata/random.go          uniform bytes (incompressible -> stored blocks)
"""
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import xflate


def gen_digits(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.integers(0, 10, n, dtype=np.uint8) + ord("0")).tobytes()


def gen_random(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, n, dtype=np.uint8).tobytes()


def gen_zeros(n, seed=0):
    return bytes(n)


_LOWER = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", dtype=np.uint8)
_LETTERS = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
_BASE64 = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789+/", dtype=np.uint8)


def gen_huffman(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty(n + 256, dtype=np.uint8)
    pos = 0
    while pos < n:
        seg = int(rng.integers(16, 80))
        kind = rng.random()
        if kind < 0.75:
            sub = rng.random()
            alpha = _LOWER if sub < 0.10 else (_LETTERS if sub < 0.70 else _BASE64)
            out[pos:pos + seg] = alpha[rng.integers(0, len(alpha), seg)]
            pos += seg
        else:
            # UTF-8 encoded code points: half 2-byte (U+0080..U+07FF), half 3-byte (U+0800..U+FFFF sans surrogates)
            if rng.random() < 0.5:
                cp = rng.integers(0x80, 0x800, seg)
                b = np.empty((seg, 2), dtype=np.uint8)
                b[:, 0] = 0xC0 | (cp >> 6)
                b[:, 1] = 0x80 | (cp & 0x3F)
            else:
                cp = rng.integers(0x800, 0xD800, seg)
                b = np.empty((seg, 3), dtype=np.uint8)
                b[:, 0] = 0xE0 | (cp >> 12)
                b[:, 1] = 0x80 | ((cp >> 6) & 0x3F)
                b[:, 2] = 0x80 | (cp & 0x3F)
            flat = b.reshape(-1)
            k = min(len(flat), len(out) - pos)
            out[pos:pos + k] = flat[:k]
            pos += k
    return out[:n].tobytes()


def gen_repeats(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = bytearray(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
    while len(out) < n:
        p = rng.random()
        if p < 0.10:
            out += rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
            continue
        if p < 0.90:
            length = int(rng.integers(4, 512))
            bucket = int(rng.integers(0, 15))               # log-bucketed distance 1..32767
            dist = int(rng.integers(1 << bucket, 1 << (bucket + 1)))
        else:
            length = int(rng.integers(4, 64))
            dist = int(rng.integers(1, 16))                 # short, usually overlapping
        dist = min(dist, len(out))
        start = len(out) - dist
        if dist >= length:
            out += out[start:start + length]
        else:
            reps = -(-length // dist)
            out += (bytes(out[start:]) * reps)[:length]
    return bytes(out[:n])


GENERATORS = {"huffman": gen_huffman, "repeats": gen_repeats, "digits": gen_digits,
              "zeros": gen_zeros, "random": gen_random}


def deflate_chunk(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """One XFLATE chunk: raw DEFLATE of `raw` ended by a sync flush (00 00 ff ff), no final block."""
    z = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    out = z.compress(raw) + z.flush(zlib.Z_SYNC_FLUSH)
    if not out:
        out = b"\x00\x00\x00\xff\xff"
    return out


def assemble_stream(comp_chunks, raw_sizes, index_size=xflate.DefaultIndexSize):
    """Concatenate independently compressed chunks into an XFLATE stream with fresh
    indexes and footer (what xflate.Writer emits for FlushFull after every chunk)."""
    out = bytearray()
    w = xflate.Writer(out, xflate.WriterConfig(IndexSize=index_size))
    for c, r in zip(comp_chunks, raw_sizes):
        w._emit(c)
        lc, lr = w.idx.last()
        w.idx.Records.append((lc + len(c), lr + r))
        if len(w.idx.Records) == w.nidx:
            w.Flush(xflate.FlushIndex)
    w.Close()
    return bytes(out)


def build_mix_stream(n_chunks, chunk_size, kinds=("huffman", "repeats", "digits"), level=6,
                     strategy=zlib.Z_DEFAULT_STRATEGY, seed=1, n_distinct=None, threads=8,
                     index_size=xflate.DefaultIndexSize):
    """BASELINE config C2/C3 shape: chunk i is drawn round-robin from `kinds`, seed = seed + kind
    index + i, compressed at `level` with a sync flush per chunk.  If n_distinct < n_chunks only
    that many distinct chunks are generated and the stream tiles them (chunks are independent,
    so any concatenation plus a fresh index is a valid stream).
    -> (stream bytes, raw bytes or None when tiled)"""
    nd = n_chunks if n_distinct is None else min(n_distinct, n_chunks)

    def make(i):
        kind = kinds[i % len(kinds)]
        raw = GENERATORS[kind](chunk_size, seed + 1000003 * (i % len(kinds)) + i)
        return raw, deflate_chunk(raw, level, strategy)

    with ThreadPoolExecutor(threads) as ex:
        made = list(ex.map(make, range(nd)))
    comp = [made[i % nd][1] for i in range(n_chunks)]
    sizes = [len(made[i % nd][0]) for i in range(n_chunks)]
    stream = assemble_stream(comp, sizes, index_size)
    raw = b"".join(made[i % nd][0] for i in range(n_chunks)) if nd == n_chunks or n_chunks * chunk_size <= (256 << 20) else None
    return stream, raw
