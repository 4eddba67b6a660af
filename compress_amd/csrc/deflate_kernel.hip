// deflate_kernel.hip -- DEFLATE of independent XFLATE chunks on gfx950.
//
// What it stands in for: the per-chunk compressor behind xflate.Writer -- flate.Writer.Write of one
// chunk followed by Flush (xflate/writer.go:138-165,169-204 through xflate/flate.go:97-118), whose
// output is a DEFLATE stream without a final block that ends in the sync marker 00 00 ff ff
// (xflate/common.go:52-57).  Any valid DEFLATE coding of the same bytes is a valid chunk.  Two codings, chosen per
// chunk by XFG_DEFLATE_DYNAMIC in its descriptor, one kernel each (both go over the table, each takes its own):
//   fixed    greedy hash-table LZ77, the fixed Huffman codes (RFC 1951 3.2.6), one pass, one block per chunk:
//            the "level 1" kind;
//   dynamic  the same LZ77 twice: the first pass counts the symbols, then length-limited Huffman codes are built for
//            them (3.2.7; the header's code lengths run-length coded under a code-length code of their own) and the
//            second pass, which takes the same decisions from the same empty hash table, writes the tokens with
//            whichever of the two code sets makes the block smaller.
// A chunk that would not shrink is written as stored blocks.
//
// One wavefront owns one chunk.  Per step the 64 lanes look at 64 consecutive positions:
//   1. every lane hashes the 4 bytes at its position, reads the hash table's candidate (an earlier
//      position: the table is only written after all lanes have read) and enters its own position;
//   2. every lane measures its match against the candidate (<= 258 bytes, <= 32768 back);
//   3. a wave-uniform scan picks tokens greedily, left to right (a match of 4 or more, else a literal);
//   4. the lanes that start a token code it (<= 31 bits with the fixed codes, <= 20 + 28 with built ones), a wave
//      prefix sum gives every token its bit offset, and the bits are OR-ed into an LDS staging row that is flushed
//      to HBM in whole dwords.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/xflate_amd.h"

namespace xfd {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef uint16_t u16;

constexpr int HASH_BITS = 12;
constexpr u32 NO_POS = 0xffffffffu;
constexpr u32 MIN_MATCH = 4, MAX_MATCH = 258, MAX_DIST = 32768;
constexpr int STAGE_WORDS = 104;        // 64 tokens x 48 bits + 31 carried bits < 98 dwords
constexpr u32 NLL = 288, NDIST = 32;    // code spaces: literal/length at [0, 288), distance at [288, 320)
constexpr u32 NSYM = NLL + NDIST;

struct WaveLds {
    u32 table[1 << HASH_BITS];          // hash of 4 bytes -> latest position seen (chunk-relative)
    u32 stage[STAGE_WORDS];
    u32 freq[NSYM];                     // pass 1: how often every symbol is used
    u32 code[NSYM];                     // the codes of the block being written: bits (LSB-first) | length << 16
    // Huffman construction (build_lengths) and the header's run-length coding
    u32 weight[2 * NLL];                // node weights: the used symbols in ascending order, then the internal nodes; later depths
    u16 parent[2 * NLL];
    u16 order[NLL];                     // used symbols by ascending (frequency, symbol)
    u8 lens[NSYM];
    u8 rle_sym[NSYM];                   // code-length symbols 0..18 of the header ...
    u8 rle_extra[NSYM];                 // ... and their extra bits
    u32 clen_freq[19];
    u32 clen_code[19];
    u8 clen_lens[19];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 uni(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 load32(const u8 *p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ u32 rev(u32 code, u32 len) { return len ? __brev(code) >> (32 - len) : 0; }

// Length of the fixed literal/length code of RFC 1951 3.2.6 and the code itself, reversed for the LSB-first stream.
__device__ __forceinline__ u32 fixed_litlen_bits(u32 sym) { return sym < 144 ? 8u : sym < 256 ? 9u : sym < 280 ? 7u : 8u; }
__device__ __forceinline__ u32 fixed_litlen_code(u32 sym) {
    if (sym < 144) return rev(0x30 + sym, 8);
    if (sym < 256) return rev(0x190 + (sym - 144), 9);
    if (sym < 280) return rev(sym - 256, 7);
    return rev(0xc0 + (sym - 280), 8);
}

// A match as symbols: length symbol (257 + s) with e extra bits of value x, distance symbol ds with de extra
// bits of value dx (the inverse of lenRanges / distRanges, flate/prefix.go:23-32).
struct MatchSyms { u32 s, e, x, ds, de, dx; };
__device__ __forceinline__ MatchSyms match_syms(u32 len, u32 dist) {
    MatchSyms m;
    const u32 l = len - 3;
    if (len == 258) { m.s = 28; m.e = 0; m.x = 0; }
    else if (l < 8) { m.s = l; m.e = 0; m.x = 0; }
    else {
        m.e = (31 - __clz(l)) - 2;
        m.s = 4 * (m.e + 1) + ((l >> m.e) & 3);
        m.x = l & ((1u << m.e) - 1);
    }
    const u32 d = dist - 1;
    if (d < 4) { m.ds = d; m.de = 0; m.dx = 0; }
    else {
        m.de = (31 - __clz(d)) - 1;
        m.ds = 2 * (m.de + 1) + ((d >> m.de) & 1);
        m.dx = d & ((1u << m.de) - 1);
    }
    return m;
}

// Wave-uniform bit writer for the bits outside the token steps: words to out32[wpos..].
struct Tail {
    u64 acc;
    u32 nbits;
    u32 wpos;
};
__device__ __forceinline__ void tail_put(Tail &t, u32 *out32, u32 v, u32 n, u32 lane) {
    t.acc |= (u64)v << t.nbits;
    t.nbits += n;
    if (t.nbits >= 32) {
        if (lane == 0) out32[t.wpos] = (u32)t.acc;
        t.wpos++;
        t.acc >>= 32;
        t.nbits -= 32;
    }
}

// Stored coding of the chunk (readRawData's input, flate/reader.go:137-159): 00 LEN NLEN data per
// 65535 bytes, then the empty stored block of the sync flush.  Returns the bytes written.
__device__ u32 write_stored(const u8 *src, u32 n, u8 *dst, u32 lane) {
    u32 o = 0;
    for (u32 p = 0; p < n; p += 65535) {
        const u32 k = n - p < 65535 ? n - p : 65535;
        if (lane == 0) {
            dst[o] = 0; dst[o + 1] = (u8)k; dst[o + 2] = (u8)(k >> 8); dst[o + 3] = (u8)~k; dst[o + 4] = (u8)(~k >> 8);
        }
        for (u32 i = lane; i < k; i += 64) dst[o + 5 + i] = src[p + i];
        o += 5 + k;
    }
    if (lane == 0) { dst[o] = 0; dst[o + 1] = 0; dst[o + 2] = 0; dst[o + 3] = 0xff; dst[o + 4] = 0xff; }
    return o + 5;
}
__device__ __forceinline__ u32 stored_size(u32 n) { return n + 5 * ((n + 65534) / 65535) + 5; }

// ---------------------------------------------------------------------------
// Length-limited Huffman code lengths for the symbols [base, base + n) of L.freq into L.lens (the job of
// prefix.GenerateLengths' callers on the write side; what the reader checks is prefix.go:326-384: a complete
// code, or the lone 1-bit code of handleDegenerateCodes, flate/prefix.go:196-208).  The used symbols are ranked
// by the whole wave; the tree (two queues over the sorted weights), the depth count, the repair of depths beyond
// maxbits (leaves move down a level until the Kraft sum fits, as zlib's gen_bitlen does) and the assignment (the rarest
// symbols get the longest codes) run on lane 0: a few hundred dependent steps per block, against milliseconds
// for the block's data.  Wave-uniform control flow; ends with a wave_sync.
// ---------------------------------------------------------------------------
__device__ void build_lengths(WaveLds &L, u32 base, u32 n, u32 maxbits, u32 lane) {
    // rank of every used symbol by (frequency, symbol)
    for (u32 i = lane; i < n; i += 64) L.lens[base + i] = 0;
    u32 used = 0;
    for (u32 b0 = 0; b0 < n; b0 += 64) {
        const u32 i = b0 + lane;
        const u32 f = i < n ? L.freq[base + i] : 0;
        u32 rank = 0;
        if (f) {
            for (u32 j = 0; j < n; j++) {
                const u32 g = L.freq[base + j];
                rank += (g != 0 && (g < f || (g == f && j < i))) ? 1u : 0u;
            }
            L.order[rank] = (u16)i;
            L.weight[rank] = f;
        }
        used += (u32)__popcll(__ballot(f != 0));
    }
    wave_sync();
    const u32 m = used;                                         // (uniform)
    if (lane == 0) {
        if (m == 0) {
            L.lens[base] = 1;                                   // nothing of this alphabet is used: one code nobody sends
        } else if (m == 1) {
            L.lens[base + L.order[0]] = 1;                      // the lone 1-bit code
        } else {
            // the tree: leaves 0 .. m-1 (ascending), internal nodes m .. 2m-2 in the order they are made (ascending too)
            u32 li = 0, ni = m;
            for (u32 k = m; k < 2 * m - 1; k++) {
                u32 sum = 0;
                for (int t = 0; t < 2; t++) {
                    u32 pick;
                    if (li < m && (ni >= k || L.weight[li] <= L.weight[ni])) pick = li++; else pick = ni++;
                    sum += L.weight[pick];
                    L.parent[pick] = (u16)k;
                }
                L.weight[k] = sum;
            }
            // depths, root first (a parent is made after its children: higher index); leaves counted per depth
            u32 bl[33];
            for (u32 b = 0; b <= 32; b++) bl[b] = 0;
            L.weight[2 * m - 2] = 0;
            for (u32 k = 2 * m - 2; k-- > 0;) {
                u32 d = L.weight[L.parent[k]] + 1;
                L.weight[k] = d;
                if (k < m) bl[d < maxbits ? d : maxbits]++;         // a leaf deeper than maxbits is counted at maxbits
            }
            // Leaves pulled up to maxbits leave the code over-subscribed: by `excess`, in units of 2^-maxbits.  Every step
            // below removes one unit: a leaf of the deepest level above maxbits that has one moves a level down and takes
            // a leaf of level maxbits as its sibling (-2^-bits + 2 * 2^-(bits+1) - 2^-maxbits), as zlib's gen_bitlen does.
            u32 kraft = 0;
            for (u32 b = 1; b <= maxbits; b++) kraft += bl[b] << (maxbits - b);
            for (u32 excess = kraft - (1u << maxbits); excess > 0; excess--) {
                u32 bits = maxbits - 1;
                while (bl[bits] == 0) bits--;
                bl[bits]--;
                bl[bits + 1] += 2;
                bl[maxbits]--;
            }
            u32 idx = 0;
            for (u32 bits = maxbits; bits >= 1; bits--)
                for (u32 c = bl[bits]; c > 0; c--) L.lens[base + L.order[idx++]] = (u8)bits;
        }
    }
    wave_sync();
}

// Canonical codes (prefix.go:340-384) of the lengths at L.lens[base .. base + n) into code[0 .. n): bits reversed
// for the LSB-first stream | length << 16.  Lane 0.  The caller synchronises.
__device__ void assign_codes(const u8 *lens, u32 n, u32 *code, u32 lane) {
    if (lane != 0) return;
    u32 cnt[16], next[16];
    for (u32 b = 0; b < 16; b++) cnt[b] = 0;
    for (u32 i = 0; i < n; i++) cnt[lens[i]]++;
    cnt[0] = 0;
    u32 c = 0;
    for (u32 b = 1; b < 16; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
    for (u32 i = 0; i < n; i++) {
        const u32 l = lens[i];
        code[i] = l ? (rev(next[l]++, l) | (l << 16)) : 0;
    }
}

// One pass over the chunk's bytes.  EMIT = false: count the symbols into L.freq and the extra bits into xbits.
// EMIT = true: write the tokens with the codes of L.code after whatever t holds.  Both passes start from an empty
// hash table and take the same decisions, so the second one writes exactly what the first one counted.
// Returns false when the output passed limit_words (the stored coding is smaller).
template <bool EMIT>
__device__ bool run_pass(WaveLds &L, const u8 *src, u32 n, u32 *out32, Tail &t, u32 limit_words, u32 &xbits, u32 lane) {
    for (u32 i = lane; i < (1u << HASH_BITS); i += 64) L.table[i] = NO_POS;
    wave_sync();
    u32 cur = 0;                                                // first position not yet coded
    while (cur < n) {
        const u32 p = cur + lane;
        // ---- 1. candidates
        u32 cand = NO_POS, h = 0;
        const bool can = p + 4 <= n;
        if (can) {
            h = (load32(src + p) * 2654435761u) >> (32 - HASH_BITS);
            cand = L.table[h];
        }
        wave_sync();
        if (can) L.table[h] = p;
        wave_sync();
        // ---- 2. match lengths
        u32 mlen = 0, mdist = 0;
        if (can && cand != NO_POS && cand < p && p - cand <= MAX_DIST) {
            const u32 maxl = n - p < MAX_MATCH ? n - p : MAX_MATCH;
            u32 k = 0;
            while (k + 4 <= maxl) {
                const u32 x = load32(src + cand + k) ^ load32(src + p + k);
                if (x) { k += (u32)(__ffs((int)x) - 1) >> 3; break; }
                k += 4;
            }
            if (k + 4 > maxl) while (k < maxl && src[cand + k] == src[p + k]) k++;
            if (k >= MIN_MATCH) { mlen = k; mdist = p - cand; }
        }
        // ---- 3. greedy selection, left to right (wave-uniform)
        const u64 has = __ballot(mlen != 0);
        u64 sel = 0;
        u32 q = 0;                                              // offset in this step's 64 positions
        const u32 span = n - cur < 64 ? n - cur : 64;
        while (q < span) {
            sel |= 1ull << q;
            q += ((has >> q) & 1) ? (u32)__builtin_amdgcn_readlane((int)mlen, (int)q) : 1u;
        }
        const bool mine = (sel >> lane) & 1;
        if (!EMIT) {
            // ---- 4a. count
            u32 xb = 0;
            if (mine) {
                if (mlen) {
                    const MatchSyms ms = match_syms(mlen, mdist);
                    atomicAdd(&L.freq[257 + ms.s], 1u);
                    atomicAdd(&L.freq[NLL + ms.ds], 1u);
                    xb = ms.e + ms.de;
                } else {
                    atomicAdd(&L.freq[src[p]], 1u);
                }
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) xb += __shfl_xor(xb, d);
            xbits += uni(xb);
        } else {
            // ---- 4b. code the tokens: literal/length code and its extra bits, distance code and its extra bits
            u32 bits_a = 0, na = 0, bits_b = 0, nbb = 0;
            bool uncoded = false;                               // a symbol the counting pass did not see: its code has no bits
            if (mine) {
                if (mlen) {
                    const MatchSyms ms = match_syms(mlen, mdist);
                    const u32 ca = L.code[257 + ms.s], cb = L.code[NLL + ms.ds];
                    uncoded = (ca >> 16) == 0 || (cb >> 16) == 0;
                    na = ca >> 16; bits_a = (ca & 0xffff) | (ms.x << na); na += ms.e;          // <= 15 + 5
                    nbb = cb >> 16; bits_b = (cb & 0xffff) | (ms.dx << nbb); nbb += ms.de;     // <= 15 + 13
                } else {
                    const u32 ca = L.code[src[p]];
                    uncoded = (ca >> 16) == 0;
                    na = ca >> 16; bits_a = ca & 0xffff;
                }
            }
            // This pass replays the counting pass's LZ77 decisions, which hang on which lane wins a same-address write
            // to the hash table -- not an ordered thing.  Should the two passes ever part, a token's symbol may have no
            // code; writing it would drop the token silently (status OK, wrong bytes).  Give the chunk up instead: the
            // caller writes it as stored blocks.
            if (__ballot(uncoded)) return false;
            const u32 nb = na + nbb;
            // exclusive prefix sum of nb over the wave
            u32 incl = nb;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 v = __shfl_up(incl, d);
                if (lane >= (u32)d) incl += v;
            }
            const u32 total = uni(__shfl(incl, 63)) + t.nbits;
            const u32 off = t.nbits + incl - nb;
            for (u32 i = lane; i < STAGE_WORDS; i += 64) L.stage[i] = 0;
            wave_sync();
            if (lane == 0) L.stage[0] = (u32)t.acc;
            wave_sync();
            if (na) {
                const u64 v = (u64)bits_a << (off & 31);        // <= 20 + 31 bits
                atomicOr(&L.stage[off >> 5], (u32)v);
                if ((u32)(v >> 32)) atomicOr(&L.stage[(off >> 5) + 1], (u32)(v >> 32));
            }
            if (nbb) {
                const u32 ob = off + na;
                const u64 v = (u64)bits_b << (ob & 31);         // <= 28 + 31 bits
                atomicOr(&L.stage[ob >> 5], (u32)v);
                if ((u32)(v >> 32)) atomicOr(&L.stage[(ob >> 5) + 1], (u32)(v >> 32));
            }
            wave_sync();
            const u32 full = total >> 5;
            if (t.wpos + full + 4 > limit_words) return false;
            for (u32 i = lane; i < full; i += 64) out32[t.wpos + i] = L.stage[i];
            t.acc = uni(L.stage[full]);
            t.nbits = total & 31;
            t.wpos += full;
            wave_sync();
        }
        cur += q;                                               // a match may reach beyond the 64 positions
    }
    return true;
}

// The header of a dynamic block (what ReadPrefixCodes reads, flate/prefix.go:83-194) for the lengths in L.lens:
// HLIT, HDIST, HCLEN, the code-length code, the run-length coded lengths.  Lane 0 turns the lengths into code-length
// symbols (zeros in runs of 17 / 18, repeats of a length as 16 -- never straight after a zero run, where the Go
// standard library and the in-repo reader disagree about what 16 repeats, flate/prefix.go:127-142); the wave builds
// the code-length code; the bits go out through t.  put = false only adds up the header's size.  Returns its bits.
__device__ u32 dynamic_header(WaveLds &L, u32 *out32, Tail &t, bool put, u32 &nrle_io, u32 &hlit_io, u32 &hdist_io, u32 lane) {
    const u32 order19[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (!put) {
        // how many of each alphabet are sent
        u32 hlit = 257, hdist = 1;
        for (u32 i = 257; i < 286; i++) if (uni(L.lens[i])) hlit = i + 1;
        for (u32 i = 1; i < 30; i++) if (uni(L.lens[NLL + i])) hdist = i + 1;
        hlit_io = hlit; hdist_io = hdist;
        if (lane < 19) L.clen_freq[lane] = 0;
        wave_sync();
        if (lane == 0) {
            const u32 total = hlit + hdist;
            u32 k = 0, i = 0;
            while (i < total) {
                const u32 v = i < hlit ? L.lens[i] : L.lens[NLL + i - hlit];
                u32 run = 1;
                while (i + run < total && (i + run < hlit ? L.lens[i + run] : L.lens[NLL + i + run - hlit]) == v) run++;
                u32 left = run;
                if (v == 0) {
                    while (left >= 11) { const u32 r = left < 138 ? left : 138; L.rle_sym[k] = 18; L.rle_extra[k++] = (u8)(r - 11); left -= r; }
                    if (left >= 3) { L.rle_sym[k] = 17; L.rle_extra[k++] = (u8)(left - 3); left = 0; }
                    while (left > 0) { L.rle_sym[k] = 0; L.rle_extra[k++] = 0; left--; }
                } else {
                    L.rle_sym[k] = (u8)v; L.rle_extra[k++] = 0; left--;          // the length itself, then repeats of it
                    while (left >= 3) { const u32 r = left < 6 ? left : 6; L.rle_sym[k] = 16; L.rle_extra[k++] = (u8)(r - 3); left -= r; }
                    while (left > 0) { L.rle_sym[k] = (u8)v; L.rle_extra[k++] = 0; left--; }
                }
                i += run;
            }
            for (u32 j = 0; j < k; j++) L.clen_freq[L.rle_sym[j]]++;
            L.stage[0] = k;
        }
        wave_sync();
        nrle_io = uni(L.stage[0]);
        // the code-length code: at most 7 bits (prefix.go:92-98 reads 3 bits per length)
        // (the 19 frequencies go through the distance part of L.freq, which pass 1 is done with; the lengths come back
        // through L.lens' distance part, saved and restored around it)
        u32 save_f = 0, save_l = 0;
        if (lane < NDIST) { save_f = L.freq[NLL + lane]; save_l = L.lens[NLL + lane]; }
        wave_sync();
        if (lane < NDIST) L.freq[NLL + lane] = lane < 19 ? L.clen_freq[lane] : 0;
        wave_sync();
        build_lengths(L, NLL, 19, 7, lane);
        if (lane < 19) L.clen_lens[lane] = L.lens[NLL + lane];
        wave_sync();
        if (lane < NDIST) { L.freq[NLL + lane] = save_f; L.lens[NLL + lane] = (u8)save_l; }
        wave_sync();
        assign_codes(L.clen_lens, 19, L.clen_code, lane);
        wave_sync();
    }
    const u32 nrle = nrle_io;
    u32 hclen = 4;
    for (u32 i = 4; i < 19; i++) if (uni(L.clen_lens[order19[i]])) hclen = i + 1;
    u32 bits = 5 + 5 + 4 + 3 * hclen;
    if (put) {
        tail_put(t, out32, hlit_io - 257, 5, lane);
        tail_put(t, out32, hdist_io - 1, 5, lane);
        tail_put(t, out32, hclen - 4, 4, lane);
        for (u32 i = 0; i < hclen; i++) tail_put(t, out32, uni(L.clen_lens[order19[i]]), 3, lane);
    }
    for (u32 j = 0; j < nrle; j++) {
        const u32 sy = uni(L.rle_sym[j]);
        const u32 c = uni(L.clen_code[sy]);
        const u32 xn = sy == 16 ? 2u : sy == 17 ? 3u : sy == 18 ? 7u : 0u;
        bits += (c >> 16) + xn;
        if (put) {
            tail_put(t, out32, c & 0xffff, c >> 16, lane);
            if (xn) tail_put(t, out32, uni(L.rle_extra[j]), xn, lane);
        }
    }
    return bits;
}

__global__ void __launch_bounds__(64)
deflate_dynamic_kernel(const u8 *__restrict__ raw, u64 raw_len, const xfg_chunk *__restrict__ chunks, u32 nchunks,
                      u8 *__restrict__ comp, u64 comp_len, xfg_result *__restrict__ res)
{
    __shared__ WaveLds L;
    const u32 lane = threadIdx.x;
    for (u32 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const xfg_chunk c = chunks[ci];
        const u8 *src = raw + c.raw_off;
        const u32 n = uni((u32)c.raw_size);
        u8 *dst = comp + c.comp_off;                            // 4-byte aligned (checked by the host)
        u32 *out32 = reinterpret_cast<u32 *>(dst);
        const u32 cap = uni((u32)c.comp_size);                  // bytes this chunk may write
        if (!(c.flags & XFG_DEFLATE_DYNAMIC)) continue;          // the one-pass kernel's
        const bool dynamic = true;
        int status = XFG_ST_OK;
        u32 produced = 0;
        // an untrusted table: sizes beyond the limits, ranges outside the buffers or a misaligned slot are refused
        const bool bad_desc = c.raw_size > (u64)XFG_MAX_CHUNK_RAW || c.comp_size > (u64)XFG_MAX_CHUNK_COMP ||
                              c.raw_off > raw_len || c.raw_size > raw_len - c.raw_off ||
                              c.comp_off > comp_len || c.comp_size > comp_len - c.comp_off || (c.comp_off & 3) != 0;
        if (bad_desc) {
            status = XFG_ST_INTERNAL;
        } else if (cap < stored_size(n) + 8) {
            status = XFG_ST_OUT_FULL;                           // the slot must hold the stored coding at least
        } else {
            Tail t;
            t.acc = 2; t.nbits = 3; t.wpos = 0;                 // BFINAL = 0, BTYPE = 01
            const u32 limit_words = (stored_size(n) + 3) / 4;   // beyond this the stored coding wins
            bool give_up = false;
            u32 xbits = 0;
            bool use_dynamic = false;
            if (dynamic && n > 0) {
                // ---- pass 1: what the block will hold; then the better of the fixed and a dynamic code for it
                for (u32 i = lane; i < NSYM; i += 64) L.freq[i] = 0;
                wave_sync();
                run_pass<false>(L, src, n, out32, t, limit_words, xbits, lane);
                if (lane == 0) L.freq[256] = 1;                 // the end of the block
                wave_sync();
                build_lengths(L, 0, 286, 15, lane);
                build_lengths(L, NLL, 30, 15, lane);
                u32 nrle = 0, hlit = 0, hdist = 0;
                const u32 hdr = dynamic_header(L, out32, t, false, nrle, hlit, hdist, lane);
                u32 dyn = 0, fix = 0;
                for (u32 i = lane; i < NSYM; i += 64) {
                    const u32 f = L.freq[i];
                    dyn += f * L.lens[i];
                    fix += f * (i < NLL ? fixed_litlen_bits(i) : 5u);
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { dyn += __shfl_xor(dyn, d); fix += __shfl_xor(fix, d); }
                // (the header is written without a look at the slot's end: only a block that will fit is begun)
                const u32 dyn_bytes = (3 + hdr + uni(dyn) + xbits + 3 + 7 + 32 + 7) / 8;
                use_dynamic = uni(dyn) + hdr < uni(fix) && dyn_bytes + 16 <= stored_size(n);
                if (use_dynamic) {
                    assign_codes(L.lens, 286, L.code, lane);
                    assign_codes(L.lens + NLL, 30, L.code + NLL, lane);
                    wave_sync();
                    t.acc = 4;                                  // BFINAL = 0, BTYPE = 10
                    dynamic_header(L, out32, t, true, nrle, hlit, hdist, lane);
                }
            }
            if (!use_dynamic) {
                for (u32 i = lane; i < NSYM; i += 64)
                    L.code[i] = i < NLL ? (fixed_litlen_code(i) | (fixed_litlen_bits(i) << 16)) : (rev(i - NLL, 5) | (5u << 16));
            }
            wave_sync();
            give_up = t.wpos + 4 > limit_words || !run_pass<true>(L, src, n, out32, t, limit_words, xbits, lane);
            if (!give_up) {
                const u32 eob = uni(L.code[256]);
                tail_put(t, out32, eob & 0xffff, eob >> 16, lane);   // end of block
                tail_put(t, out32, 0, 3, lane);                 // BFINAL = 0, BTYPE = 00: the sync flush
                if (t.nbits & 7) tail_put(t, out32, 0, 8 - (t.nbits & 7), lane);
                tail_put(t, out32, 0x0000, 16, lane);           // LEN
                tail_put(t, out32, 0xffff, 16, lane);           // NLEN
                produced = t.wpos * 4 + t.nbits / 8;
                if (lane == 0) for (u32 b = 0; b < t.nbits / 8; b++) dst[t.wpos * 4 + b] = (u8)(t.acc >> (8 * b));
                if (produced >= stored_size(n)) give_up = true;
            }
            if (give_up) {
                wave_sync();
                produced = write_stored(src, n, dst, lane);
            }
        }
        if (lane == 0) {
            xfg_result r;
            r.status = status; r.reserved = 0;
            r.in_consumed = (status == XFG_ST_OK) ? (int64_t)n : 0;
            r.out_produced = produced;
            res[ci] = r;
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// The one-pass kernel: fixed Huffman codes worked out arithmetically, a token is at most 31 bits, 16.7 KB of LDS.
// It takes the chunks without XFG_DEFLATE_DYNAMIC and skips the others.
// ---------------------------------------------------------------------------
constexpr int FIXED_STAGE_WORDS = 72;   // 64 tokens x 31 bits + 31 carried bits < 64 dwords
struct FixedLds {
    u32 table[1 << HASH_BITS];
    u32 stage[FIXED_STAGE_WORDS];
};

// Fixed literal/length code of RFC 1951 3.2.6, reversed for the LSB-first bit stream.
__device__ __forceinline__ void fixed_litlen_old(u32 sym, u32 &bits, u32 &n) {
    if (sym < 144) { bits = rev(0x30 + sym, 8); n = 8; }
    else if (sym < 256) { bits = rev(0x190 + (sym - 144), 9); n = 9; }
    else if (sym < 280) { bits = rev(sym - 256, 7); n = 7; }
    else { bits = rev(0xc0 + (sym - 280), 8); n = 8; }
}

// One match as bits: length symbol + extra, distance symbol + extra (the inverse of lenRanges /
// distRanges, flate/prefix.go:23-32).  At most 8 + 5 + 5 + 13 = 31 bits.
__device__ __forceinline__ void code_match_fixed(u32 len, u32 dist, u64 &bits, u32 &n) {
    u32 s, e, x;
    const u32 l = len - 3;
    if (len == 258) { s = 28; e = 0; x = 0; }
    else if (l < 8) { s = l; e = 0; x = 0; }
    else {
        e = (31 - __clz(l)) - 2;
        s = 4 * (e + 1) + ((l >> e) & 3);
        x = l & ((1u << e) - 1);
    }
    u32 b, nb;
    fixed_litlen_old(257 + s, b, nb);
    bits = b;
    n = nb;
    bits |= (u64)x << n;
    n += e;
    const u32 d = dist - 1;
    u32 ds, de, dx;
    if (d < 4) { ds = d; de = 0; dx = 0; }
    else {
        de = (31 - __clz(d)) - 1;
        ds = 2 * (de + 1) + ((d >> de) & 1);
        dx = d & ((1u << de) - 1);
    }
    bits |= (u64)rev(ds, 5) << n;
    n += 5;
    bits |= (u64)dx << n;
    n += de;
}

__global__ void __launch_bounds__(64)
deflate_fixed_kernel(const u8 *__restrict__ raw, u64 raw_len, const xfg_chunk *__restrict__ chunks, u32 nchunks,
                      u8 *__restrict__ comp, u64 comp_len, xfg_result *__restrict__ res)
{
    __shared__ FixedLds L;
    const u32 lane = threadIdx.x;
    for (u32 ci = blockIdx.x; ci < nchunks; ci += gridDim.x) {
        const xfg_chunk c = chunks[ci];
        if (c.flags & XFG_DEFLATE_DYNAMIC) continue;             // the other kernel's
        const u8 *src = raw + c.raw_off;
        const u32 n = uni((u32)c.raw_size);
        u8 *dst = comp + c.comp_off;                            // 4-byte aligned (checked by the host)
        u32 *out32 = reinterpret_cast<u32 *>(dst);
        const u32 cap = uni((u32)c.comp_size);                  // bytes this chunk may write
        int status = XFG_ST_OK;
        u32 produced = 0;
        // an untrusted table: sizes beyond the limits, ranges outside the buffers or a misaligned slot are refused
        const bool bad_desc = c.raw_size > (u64)XFG_MAX_CHUNK_RAW || c.comp_size > (u64)XFG_MAX_CHUNK_COMP ||
                              c.raw_off > raw_len || c.raw_size > raw_len - c.raw_off ||
                              c.comp_off > comp_len || c.comp_size > comp_len - c.comp_off || (c.comp_off & 3) != 0;
        if (bad_desc) {
            status = XFG_ST_INTERNAL;
        } else if (cap < stored_size(n) + 8) {
            status = XFG_ST_OUT_FULL;                           // the slot must hold the stored coding at least
        } else {
            for (u32 i = lane; i < (1u << HASH_BITS); i += 64) L.table[i] = NO_POS;
            wave_sync();
            Tail t;
            t.acc = 2; t.nbits = 3; t.wpos = 0;                 // BFINAL = 0, BTYPE = 01
            const u32 limit_words = (stored_size(n) + 3) / 4;   // beyond this the stored coding wins
            bool give_up = false;
            u32 cur = 0;                                        // first position not yet coded
            while (cur < n) {
                const u32 p = cur + lane;
                // ---- 1. candidates
                u32 cand = NO_POS, h = 0;
                const bool can = p + 4 <= n;
                if (can) {
                    h = (load32(src + p) * 2654435761u) >> (32 - HASH_BITS);
                    cand = L.table[h];
                }
                wave_sync();
                if (can) L.table[h] = p;
                wave_sync();
                // ---- 2. match lengths
                u32 mlen = 0, mdist = 0;
                if (can && cand != NO_POS && cand < p && p - cand <= MAX_DIST) {
                    const u32 maxl = n - p < MAX_MATCH ? n - p : MAX_MATCH;
                    u32 k = 0;
                    while (k + 4 <= maxl) {
                        const u32 x = load32(src + cand + k) ^ load32(src + p + k);
                        if (x) { k += (u32)(__ffs((int)x) - 1) >> 3; break; }
                        k += 4;
                    }
                    if (k + 4 > maxl) while (k < maxl && src[cand + k] == src[p + k]) k++;
                    if (k >= MIN_MATCH) { mlen = k; mdist = p - cand; }
                }
                // ---- 3. greedy selection, left to right (wave-uniform)
                const u64 has = __ballot(mlen != 0);
                u64 sel = 0;
                u32 q = 0;                                      // offset in this step's 64 positions
                const u32 span = n - cur < 64 ? n - cur : 64;
                while (q < span) {
                    sel |= 1ull << q;
                    q += ((has >> q) & 1) ? (u32)__builtin_amdgcn_readlane((int)mlen, (int)q) : 1u;
                }
                // ---- 4. code the tokens
                const bool mine = (sel >> lane) & 1;
                u64 bits = 0;
                u32 nb = 0;
                if (mine) {
                    if (mlen) code_match_fixed(mlen, mdist, bits, nb);
                    else { u32 b; fixed_litlen_old(src[p], b, nb); bits = b; }
                }
                // exclusive prefix sum of nb over the wave
                u32 incl = nb;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const u32 v = __shfl_up(incl, d);
                    if (lane >= (u32)d) incl += v;
                }
                const u32 total = uni(__shfl(incl, 63)) + t.nbits;
                const u32 off = t.nbits + incl - nb;
                for (u32 i = lane; i < FIXED_STAGE_WORDS; i += 64) L.stage[i] = 0;
                wave_sync();
                if (lane == 0) L.stage[0] = (u32)t.acc;
                wave_sync();
                if (nb) {
                    const u64 v = bits << (off & 31);           // <= 31 + 31 bits
                    atomicOr(&L.stage[off >> 5], (u32)v);
                    if ((u32)(v >> 32)) atomicOr(&L.stage[(off >> 5) + 1], (u32)(v >> 32));
                }
                wave_sync();
                const u32 full = total >> 5;
                if (t.wpos + full + 4 > limit_words) { give_up = true; break; }
                if (lane < full) out32[t.wpos + lane] = L.stage[lane];
                t.acc = uni(L.stage[full]);
                t.nbits = total & 31;
                t.wpos += full;
                wave_sync();
                cur += q;                                       // a match may reach beyond the 64 positions
            }
            if (!give_up) {
                tail_put(t, out32, 0, 7, lane);                 // end of block: symbol 256 is 0000000
                tail_put(t, out32, 0, 3, lane);                 // BFINAL = 0, BTYPE = 00: the sync flush
                if (t.nbits & 7) tail_put(t, out32, 0, 8 - (t.nbits & 7), lane);
                tail_put(t, out32, 0x0000, 16, lane);           // LEN
                tail_put(t, out32, 0xffff, 16, lane);           // NLEN
                produced = t.wpos * 4 + t.nbits / 8;
                if (lane == 0) for (u32 b = 0; b < t.nbits / 8; b++) dst[t.wpos * 4 + b] = (u8)(t.acc >> (8 * b));
                if (produced >= stored_size(n)) give_up = true;
            }
            if (give_up) {
                wave_sync();
                produced = write_stored(src, n, dst, lane);
            }
        }
        if (lane == 0) {
            xfg_result r;
            r.status = status; r.reserved = 0;
            r.in_consumed = (status == XFG_ST_OK) ? (int64_t)n : 0;
            r.out_produced = produced;
            res[ci] = r;
        }
        wave_sync();
    }
}

} // namespace xfd

extern "C" hipError_t xfg_launch_deflate(const void *d_raw, size_t raw_len, const void *d_chunks, uint32_t n,
                                         void *d_comp, size_t comp_len, void *d_res, int num_cu, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    // Which chunks ask for dynamic codes is written in the table, on the device: both kernels go over it and each
    // takes its own (a chunk the other one owns costs a descriptor read).
    uint32_t cap = (uint32_t)num_cu * 8u;                       // 16.7 KB of LDS per wave: 8 and some per CU
    uint32_t blocks = n < cap ? n : cap;
    hipLaunchKernelGGL(xfd::deflate_fixed_kernel, dim3(blocks), dim3(64), 0, stream,
                       (const uint8_t *)d_raw, (uint64_t)raw_len, (const xfg_chunk *)d_chunks, n, (uint8_t *)d_comp, (uint64_t)comp_len, (xfg_result *)d_res);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    cap = (uint32_t)num_cu * 6u;                                // 24.6 KB of LDS per wave: 6 per CU
    blocks = n < cap ? n : cap;
    hipLaunchKernelGGL(xfd::deflate_dynamic_kernel, dim3(blocks), dim3(64), 0, stream,
                       (const uint8_t *)d_raw, (uint64_t)raw_len, (const xfg_chunk *)d_chunks, n, (uint8_t *)d_comp, (uint64_t)comp_len, (xfg_result *)d_res);
    return hipGetLastError();
}
