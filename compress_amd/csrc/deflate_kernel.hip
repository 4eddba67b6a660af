// deflate_kernel.hip -- DEFLATE of independent XFLATE chunks on gfx950.
//
// What it stands in for: the per-chunk compressor behind xflate.Writer -- flate.Writer.Write of one
// chunk followed by Flush (xflate/writer.go:138-165,169-204 through xflate/flate.go:97-118), whose
// output is a DEFLATE stream without a final block that ends in the sync marker 00 00 ff ff
// (xflate/common.go:52-57).  Any valid DEFLATE coding of the same bytes is a valid chunk; this one is
// of the "level 1" kind: greedy hash-table LZ77, fixed Huffman codes (RFC 1951 3.2.6), one block per
// chunk.  A chunk that would not shrink is written as stored blocks.
//
// One wavefront owns one chunk.  Per step the 64 lanes look at 64 consecutive positions:
//   1. every lane hashes the 4 bytes at its position, reads the hash table's candidate (an earlier
//      position: the table is only written after all lanes have read) and enters its own position;
//   2. every lane measures its match against the candidate (<= 258 bytes, <= 32768 back);
//   3. a wave-uniform scan picks tokens greedily, left to right (a match of 4 or more, else a literal);
//   4. the lanes that start a token code it (<= 31 bits), a wave prefix sum gives every token its bit
//      offset, and the bits are OR-ed into an LDS staging row that is flushed to HBM in whole dwords.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/xflate_amd.h"

namespace xfd {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int HASH_BITS = 12;
constexpr u32 NO_POS = 0xffffffffu;
constexpr u32 MIN_MATCH = 4, MAX_MATCH = 258, MAX_DIST = 32768;
constexpr int STAGE_WORDS = 72;         // 64 tokens x 31 bits + 31 carried bits < 64 dwords

struct WaveLds {
    u32 table[1 << HASH_BITS];          // hash of 4 bytes -> latest position seen (chunk-relative)
    u32 stage[STAGE_WORDS];
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u32 uni(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 load32(const u8 *p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ u32 rev(u32 code, u32 len) { return __brev(code) >> (32 - len); }

// Fixed literal/length code of RFC 1951 3.2.6, reversed for the LSB-first bit stream.
__device__ __forceinline__ void fixed_litlen(u32 sym, u32 &bits, u32 &n) {
    if (sym < 144) { bits = rev(0x30 + sym, 8); n = 8; }
    else if (sym < 256) { bits = rev(0x190 + (sym - 144), 9); n = 9; }
    else if (sym < 280) { bits = rev(sym - 256, 7); n = 7; }
    else { bits = rev(0xc0 + (sym - 280), 8); n = 8; }
}

// One match as bits: length symbol + extra, distance symbol + extra (the inverse of lenRanges /
// distRanges, flate/prefix.go:23-32).  At most 8 + 5 + 5 + 13 = 31 bits.
__device__ __forceinline__ void code_match(u32 len, u32 dist, u64 &bits, u32 &n) {
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
    fixed_litlen(257 + s, b, nb);
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

// Wave-uniform bit writer for the few bits outside the token steps: words to out32[wpos..].
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

__global__ void __launch_bounds__(64)
deflate_chunks_kernel(const u8 *__restrict__ raw, u64 raw_len, const xfg_chunk *__restrict__ chunks, u32 nchunks,
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
                    if (mlen) code_match(mlen, mdist, bits, nb);
                    else { u32 b; fixed_litlen(src[p], b, nb); bits = b; }
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
                for (u32 i = lane; i < STAGE_WORDS; i += 64) L.stage[i] = 0;
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
    uint32_t cap = (uint32_t)num_cu * 8u;                       // 16.7 KB of LDS per wave: 8 and some per CU
    uint32_t blocks = n < cap ? n : cap;
    hipLaunchKernelGGL(xfd::deflate_chunks_kernel, dim3(blocks), dim3(64), 0, stream,
                       (const uint8_t *)d_raw, (uint64_t)raw_len, (const xfg_chunk *)d_chunks, n, (uint8_t *)d_comp, (uint64_t)comp_len, (xfg_result *)d_res);
    return hipGetLastError();
}
