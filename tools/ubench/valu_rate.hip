// Microbenchmark: VALU issue rate per SIMD at 1/2/4/8 waves per SIMD, for the instruction
// kinds the inflate kernel's walk loops are made of.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t *out, int iters, unsigned long long *cyc)
{
    __shared__ uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 37u) & 1023u;
    __syncthreads();
    uint32_t a = threadIdx.x, b = a * 3 + 1, c = a ^ 5, d = a + 7, e = a * 9, f = a + 11, g = a ^ 13, h = a + 15;
    uint64_t q = a * 0x9e3779b97f4a7c15ull, r = ~q;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {           // 8 independent v_add_u32
#pragma unroll
            for (int u = 0; u < 8; u++) { a += 1; b += 3; c += 5; d += 7; e += 9; f += 11; g += 13; h += 15;
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)); }
        } else if (KIND == 1) {    // dependent chain v_add_u32
#pragma unroll
            for (int u = 0; u < 64; u++) { a += b; asm volatile("" : "+v"(a)); }
        } else if (KIND == 2) {    // 64-bit shifts, independent pairs
#pragma unroll
            for (int u = 0; u < 16; u++) { q >>= (a & 7); r >>= (b & 7); q |= 0x8000000000000000ull; r |= 0x8000000000000000ull;
                asm volatile("" : "+v"(q), "+v"(r)); }
        } else if (KIND == 3) {    // bfe + bitop3 mix, independent
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a = __builtin_amdgcn_ubfe(a, 3u, 9u) + u; b = __builtin_amdgcn_bitop3_b32(b, c, d, 0xCA); c = (uint32_t)__builtin_amdgcn_sbfe((int)c, 11u, 1u) ^ e; d += e;
                e = __builtin_amdgcn_ubfe(e, 5u, 5u) + f; f = __builtin_amdgcn_bitop3_b32(f, g, h, 0xCA); g += 1; h += 3;
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)); }
        } else if (KIND == 4) {    // dependent LDS read chain (ds_read_b32 latency)
#pragma unroll
            for (int u = 0; u < 16; u++) { a = lds[a & 1023]; }
        } else if (KIND == 5) {    // two independent LDS read chains
#pragma unroll
            for (int u = 0; u < 16; u++) { a = lds[a & 1023]; b = lds[b & 1023]; }
        } else if (KIND == 6) {    // unaligned 64-bit LDS reads, independent
            const uint8_t *p = reinterpret_cast<const uint8_t *>(lds);
#pragma unroll
            for (int u = 0; u < 8; u++) { uint64_t v; __builtin_memcpy(&v, p + ((a + u * 9) & 2047) , 8); q += v; }
        } else if (KIND == 7) {    // aligned 64-bit LDS reads, independent
            const uint64_t *p = reinterpret_cast<const uint64_t *>(lds);
#pragma unroll
            for (int u = 0; u < 8; u++) { q += p[(a + u * 9) & 255]; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t)q + (uint32_t)r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_iter, int iters)
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps;
        uint32_t *out; unsigned long long *cyc;
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        const double ops = (double)per_iter * iters;
        // wall-clock rate per SIMD: ops of all waves on one SIMD / time
        const double ns_per_op_simd = ms * 1e6 / (ops * wps);
        printf("%-28s waves/SIMD %d: %.2f memtime-ticks per op per wave, %.3f ns per op per SIMD (%.2f cycles at 2.4 GHz), %.3f ms\n",
               name, wps, mean / ops, ns_per_op_simd, ns_per_op_simd * 2.4, ms);
        hipFree(out); hipFree(cyc);
    }
}

int main()
{
    run<0>("v_add_u32 x8 indep", 64, 20000);
    run<1>("v_add_u32 dependent", 64, 20000);
    run<2>("v_lshr_b64+or indep (x4)", 64, 20000);
    run<3>("bfe/bitop3/sbfe mix", 64, 20000);
    run<4>("ds_read_b32 dependent", 16, 20000);
    run<5>("ds_read_b32 2 chains", 32, 20000);
    run<6>("ds_read_b64 unaligned", 8, 20000);
    run<7>("ds_read_b64 aligned", 8, 20000);
    return 0;
}
