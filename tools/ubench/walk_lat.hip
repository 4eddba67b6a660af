// Microbenchmark: latency of one Huffman walk step as the inflate kernel does it, piece by piece.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8; typedef int32_t i32;
typedef const __attribute__((address_space(3))) u64 *lds_cu64p;
struct __attribute__((packed)) PackedU64 { u64 v; };
typedef const __attribute__((address_space(3))) PackedU64 *lds_cpu64p;

struct Lds { u64 tab[512]; u64 win[256]; u32 mask[8 * 64]; };

template <int V0>
__global__ void __launch_bounds__(64) k(u32 *out, int iters, unsigned long long *cyc, u32 seed)
{
    __shared__ Lds L;
    constexpr int R = V0 >= 100 ? V0 / 100 : -1, M = (V0 / 10) % 10, X = V0 % 10;
    constexpr int V = V0 >= 100 ? (R == 1 ? 7 : R == 2 ? 2 : 6) : V0;   // legacy selector for the refill code
    const u32 lane = threadIdx.x;
    const u32 tab0 = (u32)(uintptr_t)(lds_cu64p)L.tab;
    const u32 win0 = (u32)(uintptr_t)(const __attribute__((address_space(3))) u8 *)reinterpret_cast<const u8 *>(L.win);
    const u32 st_lit = (511u << 3) | (tab0 << 16);
    for (u32 i = lane; i < 512; i += 64) {
        u32 take = 5 + ((i * 2654435761u) >> 29);                // 5..12 bits
        L.tab[i] = (u64)(take | (1u << 14) | (i << 16)) | ((u64)st_lit << 32);
    }
    for (u32 i = lane; i < 256; i += 64) L.win[i] = (i + seed) * 0x9e3779b97f4a7c15ull;
    for (u32 i = lane; i < 512; i += 64) L.mask[i] = 0;
    __syncthreads();
    u32 acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // one "segment": 32 steps from a lane-specific start
        u32 start = lane * 208 + (it & 7);
        u32 wp = win0 + (start >> 3);
        u64 nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
        u64 buf = (nextw >> (start & 7)) << 3;
        u32 c3 = 51 - (start & 7);
        wp += 6;
        nextw = ((lds_cpu64p)(uintptr_t)wp)->v;
        u32 st = st_lit, rel = 0, e = 0;
        u32 *mrow = L.mask + lane;
        for (int s = 0; s < 16; s++) {
            if (V >= 1) {
                buf |= nextw << c3;
                wp += (63 - c3) >> 3;
                c3 |= 56;
                if (V == 1) nextw = *(lds_cu64p)(uintptr_t)(wp & ~7u);          // aligned refill
                else if (V == 6) {                                               // three aligned dwords, funnel-shifted by bytes
                    const __attribute__((address_space(3))) u32 *q = (const __attribute__((address_space(3))) u32 *)(uintptr_t)(wp & ~3u);
                    const u32 d0 = q[0], d1 = q[1], d2 = q[2];
                    const u32 lo = __builtin_amdgcn_alignbyte(d1, d0, wp & 3);
                    const u32 hi = __builtin_amdgcn_alignbyte(d2, d1, wp & 3);
                    nextw = ((u64)hi << 32) | lo;
                }
                else if (V == 7) nextw = ((lds_cpu64p)(uintptr_t)(wp & ~3u))->v;  // 8 bytes at 4-byte alignment
                else if (V == 8) nextw = ((lds_cpu64p)(uintptr_t)(wp | 1u))->v;   // 8 bytes at odd addresses
                else if (V == 9) {                                               // two 4-byte reads at any alignment
                    struct __attribute__((packed)) P32 { u32 v; };
                    typedef const __attribute__((address_space(3))) P32 *lp32;
                    const u32 a = ((lp32)(uintptr_t)wp)->v; asm volatile("" ::: "memory"); const u32 b2 = ((lp32)(uintptr_t)(wp + 4))->v;
                    nextw = ((u64)b2 << 32) | a;
                }
                else nextw = ((lds_cpu64p)(uintptr_t)wp)->v;                     // unaligned refill
                if (wp > win0 + 1900) wp -= 1800;
            }
#pragma unroll
            for (int step = 0; step < 2; step++) {
                const u32 t = (u32)buf & st;
                const u64 ent = *(lds_cu64p)(uintptr_t)((t & 0xffffu) + (st >> 16));
                e = (u32)ent;
                st = (u32)(ent >> 32);
                const u32 take = e & 63;
                buf >>= take; c3 -= take; rel += take;
                if (V == 0) buf |= (u64)e << 40;                 // keep bits coming without a refill
                if (V0 >= 100) {
                    const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 14u, 1u);
                    if (M == 1) atomicOr(&mrow[((rel >> 5) & 7) << 6], m_tok & (1u << (rel & 31)));
                    if (M == 2) mrow[((rel >> 5) & 7) << 6] = m_tok & (1u << (rel & 31));
                    if (X == 1) { const u32 over = 100000u - 1 - rel; if ((i32)((over & m_tok) | e) < 0) goto out; }
                } else
                if (V >= 3) {
                    const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 14u, 1u);
                    if (V == 3 || V >= 5) atomicOr(&mrow[((rel >> 5) & 7) << 6], m_tok & (1u << (rel & 31)));
                    if (V == 4) mrow[((rel >> 5) & 7) << 6] = m_tok & (1u << (rel & 31));      // plain store instead of the atomic
                }
                if (V0 < 100 && V >= 5) {
                    const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 14u, 1u);
                    const u32 over = 100000u - 1 - rel;
                    if ((i32)((over & m_tok) | e) < 0) goto out;
                }
            }
        }
    out:
        acc += e + rel + (u32)buf;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}


template <int V>
static void run(const char *name)
{
    const int iters = 2000;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%-44s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = cus * 4 * wps;
        u32 *out; unsigned long long *cyc;
        (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc, 1u);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc, 2u);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        printf("  w/SIMD %d: %6.1f ticks/step/wave %6.1f /SIMD", wps, mean / (32.0 * iters), mean / (32.0 * iters) / wps);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
    run<100>("R1 (b64 4-aligned) only");
    run<101>("R1 + exit test");
    run<110>("R1 + ds_or");
    run<120>("R1 + ds_write");
    run<111>("R1 + ds_or + exit");
    run<121>("R1 + ds_write + exit");
    run<300>("R3 (3 dwords alignbyte) only");
    run<321>("R3 + ds_write + exit");
    run<200>("R2 (unaligned b64) only");
    run<0>("V0 lookup chain only");
    run<1>("V1 + aligned refill");
    run<2>("V2 + unaligned refill");
    run<3>("V3 + unaligned refill + ds_or marks");
    run<4>("V4 + unaligned refill + ds_write marks");
    run<5>("V5 full spec step (ds_or, exit test)");
    run<6>("V6 refill: 3 aligned dwords + alignbyte");
    run<7>("V7 refill: b64 at 4-byte alignment");
    run<8>("V8 refill: b64 at odd addresses");
    run<9>("V9 refill: two b32 at any alignment");
    return 0;
}
