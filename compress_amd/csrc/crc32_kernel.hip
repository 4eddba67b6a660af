// crc32_kernel.hip -- CRC-32 (IEEE 802.3) of every chunk's decoded bytes, on gfx950.
//
// XFLATE streams embedded in gzip (RFC 1952) carry a CRC-32 of
// the uncompressed data in the trailer (xflate/example_test.go:171-182 computes it with
// hash/crc32.NewIEEE on the CPU while writing); a reader that decodes on the GPU checks it here,
// where the bytes already are.  One wave per chunk: the chunk is cut into 64 equal pieces that
// END at the chunk's end (leading lanes may be short or empty), each lane runs a table-driven
// CRC (slicing-by-4, tables in LDS) over its piece, and the 64 partial remainders are folded
// with a tree of multiplications by x^(8 * piece * 2^k) mod P.  HBM-bound: every output byte is
// read once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/xflate_amd.h"

namespace xfg_crc {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 POLY = 0xedb88320u;        // reflected IEEE polynomial (hash/crc32.IEEE)

// a(x) * b(x) mod P, operands bit-reflected (bit 31 = x^0), as in zlib's multmodp.
__host__ __device__ inline u32 multmodp(u32 a, u32 b) {
    u32 p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1) ? (b >> 1) ^ POLY : b >> 1;
    }
    return p;
}
// x^(8 * n) mod P
__host__ __device__ inline u32 xpow8(u64 n) {
    u32 p = 0x80000000u;                  // x^0
    u32 sq = 0x00800000u;                 // x^8
    while (n) {
        if (n & 1) p = multmodp(sq, p);
        sq = multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}

__device__ __forceinline__ u32 uni(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ void __launch_bounds__(64)
crc32_chunks_kernel(const u8 *__restrict__ raw, u64 raw_len, const xfg_chunk *__restrict__ chunks, const xfg_result *__restrict__ res,
                    u32 n, u32 *__restrict__ crc_out)
{
    __shared__ u32 T[4][256];
    const u32 lane = threadIdx.x;
    // slicing-by-4 tables: T[0] is the byte table, T[k][i] = T[k-1][i] advanced by one more zero byte
    for (u32 i = lane; i < 256; i += 64) {
        u32 c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ POLY : c >> 1;
        T[0][i] = c;
    }
    __syncthreads();
    for (int k = 1; k < 4; k++) {
        for (u32 i = lane; i < 256; i += 64) {
            const u32 c = T[k - 1][i];
            T[k][i] = (c >> 8) ^ T[0][c & 0xff];
        }
        __syncthreads();
    }

    u32 g_piece = 0xffffffffu, g[6] = {0, 0, 0, 0, 0, 0};       // x^(8 * piece * 2^k) for the piece size seen last
    for (u32 ci = blockIdx.x; ci < n; ci += gridDim.x) {
        const u64 off = chunks[ci].raw_off;
        u64 len64 = res ? (u64)res[ci].out_produced : chunks[ci].raw_size;
        if (len64 > chunks[ci].raw_size) len64 = chunks[ci].raw_size;
        // an untrusted table: a chunk beyond the limit or outside the buffer hashes nothing
        if (len64 > (u64)XFG_MAX_CHUNK_RAW || off > raw_len || len64 > raw_len - off) len64 = 0;
        const u32 len = uni((u32)len64);                        // <= XFG_MAX_CHUNK_RAW < 2^31
        const u8 *p = raw + off;
        // pieces of `piece` bytes (a multiple of 16) that end at the chunk's end: lane l owns
        // [len - (64 - l) * piece, len - (63 - l) * piece) clipped to [0, len)
        // (never a multiple of 2 KiB: 64 lanes striding by a large power of two queue on one memory channel; measured 1.1 -> 1.7 TB/s at 256 KiB chunks)
        u32 piece = ((len + 63) / 64 + 15) & ~15u;
        if ((piece & 2047) == 0) piece += 16;
        const long long lo0 = (long long)len - (long long)(64 - lane) * piece;
        const long long hi0 = lo0 + piece;
        const u32 lo = lo0 < 0 ? 0u : (u32)lo0;
        const u32 hi = hi0 < 0 ? 0u : (u32)hi0;
        // the lane that owns byte 0 starts from the all-ones register; the others from zero (linearity)
        u32 c = (hi > 0 && lo == 0) ? 0xffffffffu : 0u;
        u32 i = lo;
        // head: up to the first 16-byte aligned address
        while (i < hi && (((uintptr_t)(p + i)) & 15)) { c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xff]; i++; }
        for (; i + 16 <= hi; i += 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + i);
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c ^= w[k];
                c = T[3][c & 0xff] ^ T[2][(c >> 8) & 0xff] ^ T[1][(c >> 16) & 0xff] ^ T[0][c >> 24];
            }
        }
        for (; i < hi; i++) c = (c >> 8) ^ T[0][(c ^ p[i]) & 0xff];

        // fold: at level s (1, 2, 4, ...) lane l (a multiple of 2s) absorbs lane l + s, whose span of
        // s pieces follows its own: R = R_left * x^(8 * piece * s) + R_right
        if (piece != g_piece) {                                 // wave-uniform; chunks of one stream mostly share it
            g_piece = piece;
            g[0] = xpow8(piece);
#pragma unroll
            for (int k = 1; k < 6; k++) g[k] = multmodp(g[k - 1], g[k - 1]);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const u32 s = 1u << k;
            const u32 right = __shfl_down(c, s);
            if ((lane & (2 * s - 1)) == 0) c = multmodp(g[k], c) ^ right;
        }
        if (lane == 0) crc_out[ci] = len ? ~c : 0u;
    }
}

} // namespace xfg_crc

extern "C" hipError_t xfg_launch_crc32(const void *d_raw, size_t raw_len, const void *d_chunks, const void *d_res, uint32_t n,
                                       void *d_crc, int num_cu, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)num_cu * 16u;
    if (blocks > n) blocks = n;
    hipLaunchKernelGGL(xfg_crc::crc32_chunks_kernel, dim3(blocks), dim3(64), 0, stream,
                       (const uint8_t *)d_raw, (uint64_t)raw_len, (const xfg_chunk *)d_chunks, (const xfg_result *)d_res, n, (uint32_t *)d_crc);
    return hipGetLastError();
}

// crc32(A || B) from crc32(A), crc32(B) and len(B): zlib's crc32_combine, for the whole-stream
// value a gzip trailer holds (xflate/example_test.go:181).
extern "C" uint32_t xfg_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2)
{
    return xfg_crc::multmodp(xfg_crc::xpow8(len2), crc1) ^ crc2;
}
