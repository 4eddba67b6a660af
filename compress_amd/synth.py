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


class _SparseOut(object):
    """A writer that keeps what is written with its offset and can skip ahead: the index and footer of a
    stream whose chunk bytes are never materialised."""
    def __init__(self):
        self.pos = 0
        self.blobs = []          # (offset, bytes)

    def write(self, b):
        self.blobs.append((self.pos, bytes(b)))
        self.pos += len(b)

    def skip(self, n):
        self.pos += n


def assemble_sparse(comp_sizes, raw_sizes, index_size=xflate.DefaultIndexSize):
    """The layout of the XFLATE stream assemble_stream() would build from chunks of these sizes, without the
    chunks: -> (stream length, [(offset, bytes)...] for every index and the footer, in stream order).
    Chunk i's bytes belong at the offsets the records say; the caller places (or tiles) them."""
    out = _SparseOut()
    w = xflate.Writer(out, xflate.WriterConfig(IndexSize=index_size))
    for c, r in zip(comp_sizes, raw_sizes):
        out.skip(c)
        w.OutputOffset += c
        lc, lr = w.idx.last()
        w.idx.Records.append((lc + c, lr + r))
        if len(w.idx.Records) == w.nidx:
            w.Flush(xflate.FlushIndex)
    w.Close()
    return out.pos, out.blobs


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


# ---------------------------------------------------------------------------
# Token-level DEFLATE writer (BASELINE config C4).  zlib will not reliably choose
# maximum-length matches at prescribed distances, so chunks of that shape are
# written token by token: RFC 1951 section 3.2.5 length / distance codes, one fixed
# (3.2.6) or dynamic (3.2.7) Huffman block, then an empty stored block as the sync
# marker.  Tokens are ints (literal byte) or (length, distance) pairs.
# ---------------------------------------------------------------------------
_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
          4097, 6145, 8193, 12289, 16385, 24577]
_DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
_CLEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class _BitWriter(object):
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, v, n):                       # LSB first
        self.acc |= v << self.n
        self.n += n
        if self.n >= 64:
            self.flush_bytes()

    def flush_bytes(self):
        k = self.n >> 3
        self.out += (self.acc & ((1 << (8 * k)) - 1)).to_bytes(k, "little")
        self.acc >>= 8 * k
        self.n -= 8 * k

    def code(self, c, n):                       # Huffman codes go MSB first
        self.bits(int(format(c, "0%db" % n)[::-1], 2), n)

    def align(self):
        self.flush_bytes()
        if self.n:
            self.out.append(self.acc & 0xff)
            self.acc, self.n = 0, 0


def _canonical(lens):
    maxl = max(lens)
    cnt = [0] * (maxl + 2)
    for l in lens:
        if l:
            cnt[l] += 1
    code, nxt = 0, [0] * (maxl + 2)
    for l in range(1, maxl + 1):
        code = (code + cnt[l - 1]) << 1
        nxt[l] = code
    codes = [0] * len(lens)
    for s, l in enumerate(lens):
        if l:
            codes[s] = nxt[l]
            nxt[l] += 1
    return codes


def _limited_lengths(freq, maxbits):
    """Huffman code lengths for freq (0 = unused), limited to maxbits: plain Huffman, then the
    overlong codes are shortened and the Kraft sum repaired by lengthening the cheapest others."""
    import heapq
    used = [s for s, f in enumerate(freq) if f]
    lens = [0] * len(freq)
    if len(used) == 1:
        lens[used[0]] = 1
        return lens
    heap = [(freq[s], s, None, None) for s in used]
    heapq.heapify(heap)
    uid = len(freq)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], uid, a, b))
        uid += 1
    stack = [(heap[0], 0)]
    while stack:
        (f, s, a, b), d = stack.pop()
        if a is None:
            lens[s] = max(d, 1)
        else:
            stack.append((a, d + 1))
            stack.append((b, d + 1))
    for s in used:
        lens[s] = min(lens[s], maxbits)
    kraft = sum(1 << (maxbits - lens[s]) for s in used)
    order = sorted(used, key=lambda s: freq[s])
    while kraft > (1 << maxbits):               # over-subscribed: lengthen the rarest symbols that still can
        for s in order:
            if lens[s] < maxbits:
                kraft -= 1 << (maxbits - lens[s] - 1)
                lens[s] += 1
                break
    for s in reversed(order):                   # under-subscribed: shorten the most frequent while it fits
        while lens[s] > 1 and kraft + (1 << (maxbits - lens[s])) <= (1 << maxbits):
            kraft += 1 << (maxbits - lens[s])
            lens[s] -= 1
    assert kraft == (1 << maxbits), kraft
    return lens


def _len_sym(length):
    for i in range(28, -1, -1):
        if length >= _LBASE[i]:
            return i
    raise ValueError(length)


def _dist_sym(dist):
    for i in range(29, -1, -1):
        if dist >= _DBASE[i]:
            return i
    raise ValueError(dist)


def deflate_tokens(tokens, dynamic=True, final=False, sync=True):
    """Encode tokens as one Huffman block (+ the empty stored block of a sync flush)."""
    syms = []                                   # (litlen sym, len extra, dist sym, dist extra)
    lf, df = [0] * 286, [0] * 30
    for t in tokens:
        if isinstance(t, int):
            syms.append((t, 0, -1, 0))
            lf[t] += 1
        else:
            length, dist = t
            ls, ds = _len_sym(length), _dist_sym(dist)
            syms.append((257 + ls, length - _LBASE[ls], ds, dist - _DBASE[ds]))
            lf[257 + ls] += 1
            df[ds] += 1
    lf[256] += 1
    w = _BitWriter()
    w.bits(1 if final else 0, 1)
    if dynamic:
        w.bits(2, 2)
        if not any(df):
            df[0] = 1                           # one distance code of one bit (an unused but complete-enough tree)
        ll, dl = _limited_lengths(lf, 15), _limited_lengths(df, 15)
        if sum(1 for x in dl if x) == 1:        # a lone distance code must be 1 bit (flate/prefix.go:203-208)
            dl = [1 if x else 0 for x in dl]
        hlit = max(257, max(i for i, x in enumerate(ll) if x) + 1)
        hdist = max(1, max(i for i, x in enumerate(dl) if x) + 1)
        seq = ll[:hlit] + dl[:hdist]            # no repeat codes: every length sent as itself
        cf = [0] * 19
        for x in seq:
            cf[x] += 1
        cl = _limited_lengths(cf, 7)
        hclen = max(4, max(i for i, s in enumerate(_CLEN_ORDER) if cl[s]) + 1)
        w.bits(hlit - 257, 5)
        w.bits(hdist - 1, 5)
        w.bits(hclen - 4, 4)
        for i in range(hclen):
            w.bits(cl[_CLEN_ORDER[i]], 3)
        cc = _canonical(cl)
        for x in seq:
            w.code(cc[x], cl[x])
    else:
        w.bits(1, 2)
        ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
        dl = [5] * 30
    lc, dc = _canonical(ll), _canonical(dl)
    for s, le, ds, de in syms:
        w.code(lc[s], ll[s])
        if ds >= 0:
            w.bits(le, _LEXT[s - 257])
            w.code(dc[ds], dl[ds])
            w.bits(de, _DEXT[ds])
    w.code(lc[256], ll[256])
    if sync:
        w.bits(0, 3)                            # empty stored block: BFINAL=0, BTYPE=00
        w.align()
        w.out += b"\x00\x00\xff\xff"
    else:
        w.align()
    return bytes(w.out)


def apply_tokens(tokens):
    """What the tokens decode to (the LZ77 semantics of flate/dict_decoder.go:86-132)."""
    out = bytearray()
    for t in tokens:
        if isinstance(t, int):
            out.append(t)
        else:
            length, dist = t
            start = len(out) - dist
            if dist >= length:
                out += out[start:start + length]
            else:
                out += (bytes(out[start:]) * (-(-length // dist)))[:length]
    return bytes(out)


def match_chunk(size, seed, dists=(1024, 2048, 4096, 8192, 16384, 32768), dynamic=True):
    """BASELINE config C4: 32 KiB of random literals, then only (258, d) matches with d cycling
    through `dists`, cut to `size` bytes.  -> (raw, compressed chunk)"""
    seedlen = min(32768, size)
    toks = list(gen_random(seedlen, seed))
    n, k = seedlen, 0
    while n < size:
        length = min(258, size - n)
        if length < 3:
            toks.extend([0x55] * length)
            n += length
            break
        toks.append((length, dists[k % len(dists)]))
        n += length
        k += 1
    return apply_tokens(toks), deflate_tokens(toks, dynamic)


def run_chunk(size, byte=0, dynamic=True):
    """The degenerate case of C4: one literal, then (258, 1) matches (the shape of zeros.bin)."""
    toks, n = [byte], 1
    while n < size:
        length = min(258, size - n)
        if length < 3:
            toks.extend([byte] * length)
            break
        toks.append((length, 1))
        n += length
    return apply_tokens(toks)[:size], deflate_tokens(toks, dynamic)
