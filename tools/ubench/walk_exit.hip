// Microbenchmark: forms of the walk loop's exit test (the speculative walk of the inflate kernel: lookup chain,
// aligned top-up, mark, exit).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64; typedef uint8_t u8; typedef int32_t i32;
typedef const __attribute__((address_space(3))) u64 *lds_cu64p;
typedef const __attribute__((address_space(3))) u32 *lds_cu32p;
struct Lds { u64 tab[512]; u64 win[256 + 8]; u32 mask[8 * 64]; };

__device__ __forceinline__ u64 load64_any(u32 a) {
    const lds_cu32p q = (lds_cu32p)(uintptr_t)(a & ~3u);
    const u32 d0 = q[0], d1 = q[1], d2 = q[2];
    const u32 lo = __builtin_amdgcn_alignbyte(d1, d0, a & 3u);
    const u32 hi = __builtin_amdgcn_alignbyte(d2, d1, a & 3u);
    return ((u64)hi << 32) | lo;
}

template <int X>
__global__ void __launch_bounds__(64) k(u32 *out, int iters, unsigned long long *cyc, u32 seed)
{
    __shared__ Lds L;
    const u32 lane = threadIdx.x;
    const u32 tab0 = (u32)(uintptr_t)(lds_cu64p)L.tab;
    const u32 win0 = (u32)(uintptr_t)(const __attribute__((address_space(3))) u8 *)reinterpret_cast<const u8 *>(L.win);
    const u32 mask0 = (u32)(uintptr_t)(lds_cu32p)L.mask;
    const u32 st_lit = (511u << 3) | (tab0 << 16);
    for (u32 i = lane; i < 512; i += 64) {
        u32 take = 5 + ((i * 2654435761u) >> 29);                // 5..12 bits
        L.tab[i] = (u64)(take | (1u << 14) | (i << 16)) | ((u64)st_lit << 32);
    }
    for (u32 i = lane; i < 264; i += 64) L.win[i] = (i + seed) * 0x9e3779b97f4a7c15ull;
    for (u32 i = lane; i < 512; i += 64) L.mask[i] = 0;
    __syncthreads();
    u32 acc = 0, steps = 0;
    const u32 S1 = 208 - 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        u32 start = lane * 208 + (it & 7);
        u32 wp = win0 + (start >> 3);
        u64 nextw = load64_any(wp);
        u64 buf = (nextw >> (start & 7)) << 3;
        u32 c3 = 51 - (start & 7);
        wp += 6;
        nextw = load64_any(wp);
        u32 st = st_lit, rel = 0, e = 0;
        u32 *mrow = L.mask + lane;
        if (X == 1 || X == 2) {
            for (;;) {
                buf |= nextw << c3; wp += (63 - c3) >> 3; c3 |= 56; nextw = load64_any(wp);
#pragma unroll
                for (int step = 0; step < 2; step++) {
                    const u32 t = (u32)buf & st;
                    const u64 ent = *(lds_cu64p)(uintptr_t)((t & 0xffffu) + (st >> 16));
                    e = (u32)ent; st = (u32)(ent >> 32);
                    const u32 take = e & 63;
                    buf >>= take; c3 -= take; rel += take;
                    const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e, 14u, 1u);
                    atomicOr(&mrow[(rel >> 5) << 6], m_tok & (1u << (rel & 31)));
                    if (X == 1 || step == 1) { const u32 over = S1 - rel; if ((i32)((over & m_tok) | e) < 0) goto out; }
                    steps++;
                }
            }
        } else if (X == 3) {
            u32 am = ~0u;                                            // all ones while the lane is live
            for (;;) {
                buf |= nextw << c3; wp += (63 - c3) >> 3; c3 |= 56; nextw = load64_any(wp);
#pragma unroll
                for (int step = 0; step < 2; step++) {
                    const u32 t = (u32)buf & st;
                    const u64 ent = *(lds_cu64p)(uintptr_t)((t & 0xffffu) + (st >> 16));
                    const u32 e1 = (u32)ent; st = (u32)(ent >> 32);
                    const u32 take = e1 & 63 & am;
                    buf >>= take; c3 -= take; rel += take;
                    const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e1, 14u, 1u) & am;
                    atomicOr(&mrow[(rel >> 5) << 6], m_tok & (1u << (rel & 31)));
                    const u32 over = S1 - rel;
                    const u32 fin = (u32)((i32)(((over & m_tok) | e1) & am) >> 31);   // all ones: the lane finishes with this step
                    e = (fin & e1) | (~fin & e);
                    am &= ~fin;
                    steps += am & 1;
                }
                if (__ballot(am != 0) == 0) break;
            }
        } else if (X == 4) {
            // the loop in assembly: lanes that finish drop out of EXEC (v_cmpx), the loop ends when none is left.
            // Fixed registers (the asm cannot name the halves of a 64-bit operand):
            //   v[100:101] buf  v[102:103] nextw  v104 c3  v105 wp  v106 st  v107 rel  v[108:109] entry  v[110:111] t
            //   v[112:113] d0,d1  v114 d2  v115 a  v116 tok  v117 bit  v118 x  v119 steps
            u32 lo = (u32)buf, hi = (u32)(buf >> 32), nlo = (u32)nextw, nhi = (u32)(nextw >> 32), e0 = 0;
            const u32 mrow_a = mask0 + 4 * lane;
            asm volatile(
                "v_mov_b32 v100, %[lo]\n v_mov_b32 v101, %[hi]\n v_mov_b32 v102, %[nlo]\n v_mov_b32 v103, %[nhi]\n"
                "v_mov_b32 v104, %[c3]\n v_mov_b32 v105, %[wp]\n v_mov_b32 v106, %[st]\n v_mov_b32 v107, %[rel]\n v_mov_b32 v119, %[steps]\n"
                "s_mov_b64 s[40:41], exec\n"
                "1:\n"
                "v_lshlrev_b64 v[110:111], v104, v[102:103]\n"
                "v_or_b32 v100, v100, v110\n"
                "v_or_b32 v101, v101, v111\n"
                "v_sub_u32 v110, 63, v104\n"
                "v_lshrrev_b32 v110, 3, v110\n"
                "v_add_u32 v105, v105, v110\n"
                "v_or_b32 v104, 56, v104\n"
                "v_and_b32 v115, -4, v105\n"
                "ds_read2_b32 v[112:113], v115 offset1:1\n"
                "ds_read_b32 v114, v115 offset:8\n"
                // step 1
                "v_and_b32 v110, v100, v106\n"
                "v_add_u32_sdwa v110, v110, v106 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                "ds_read_b64 v[108:109], v110\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_alignbyte_b32 v102, v113, v112, v105\n"
                "v_alignbyte_b32 v103, v114, v113, v105\n"
                "v_and_b32 v110, 63, v108\n"
                "v_lshrrev_b64 v[100:101], v108, v[100:101]\n"
                "v_mov_b32 v106, v109\n"
                "v_sub_u32 v104, v104, v110\n"
                "v_add_u32 v107, v107, v110\n"
                "v_bfe_i32 v116, v108, 14, 1\n"
                "v_lshrrev_b32 v115, 5, v107\n"
                "v_lshl_add_u32 v115, v115, 8, %[mrow]\n"
                "v_lshlrev_b32 v117, v107, 1\n"
                "v_and_b32 v117, v117, v116\n"
                "ds_or_b32 v115, v117\n"
                "v_sub_u32 v118, %[S1], v107\n"
                "v_and_or_b32 v118, v118, v116, v108\n"
                "v_add_u32 v119, 1, v119\n"
                "v_cmpx_lt_i32 -1, v118\n"
                "s_cbranch_execz 2f\n"
                // step 2
                "v_and_b32 v110, v100, v106\n"
                "v_add_u32_sdwa v110, v110, v106 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n"
                "ds_read_b64 v[108:109], v110\n"
                "s_waitcnt lgkmcnt(0)\n"
                "v_and_b32 v110, 63, v108\n"
                "v_lshrrev_b64 v[100:101], v108, v[100:101]\n"
                "v_mov_b32 v106, v109\n"
                "v_sub_u32 v104, v104, v110\n"
                "v_add_u32 v107, v107, v110\n"
                "v_bfe_i32 v116, v108, 14, 1\n"
                "v_lshrrev_b32 v115, 5, v107\n"
                "v_lshl_add_u32 v115, v115, 8, %[mrow]\n"
                "v_lshlrev_b32 v117, v107, 1\n"
                "v_and_b32 v117, v117, v116\n"
                "ds_or_b32 v115, v117\n"
                "v_sub_u32 v118, %[S1], v107\n"
                "v_and_or_b32 v118, v118, v116, v108\n"
                "v_add_u32 v119, 1, v119\n"
                "v_cmpx_lt_i32 -1, v118\n"
                "s_cbranch_execnz 1b\n"
                "2:\n"
                "s_mov_b64 exec, s[40:41]\n"
                "v_mov_b32 %[rel], v107\n v_mov_b32 %[e0], v108\n v_mov_b32 %[steps], v119\n v_mov_b32 %[lo], v100\n"
                : [lo] "+v"(lo), [rel] "+v"(rel), [steps] "+v"(steps), [e0] "+v"(e0)
                : [hi] "v"(hi), [nlo] "v"(nlo), [nhi] "v"(nhi), [c3] "v"(c3), [wp] "v"(wp), [st] "v"(st), [mrow] "v"(mrow_a), [S1] "v"(S1)
                : "memory", "vcc", "s40", "s41", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110",
                  "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
            e = e0; buf = lo;
        }
        else if (X == 5) {
            // two segments per lane, their steps interleaved (uniform loop, masked updates)
            u32 startB = (lane + 64) * 208 + (it & 7);
            if (startB > 8 * 1900) startB -= 8 * 1000;
            u32 wpB = win0 + (startB >> 3);
            u64 nextwB = load64_any(wpB);
            u64 bufB = (nextwB >> (startB & 7)) << 3;
            u32 c3B = 51 - (startB & 7);
            wpB += 6;
            nextwB = load64_any(wpB);
            u32 stB = st_lit, relB = 0, eB = 0;
            u32 am = ~0u, amB = ~0u;
            u32 *mrowB = L.mask + lane;      // (same rows: timing only)
            for (;;) {
                buf |= nextw << c3; wp += (63 - c3) >> 3; c3 |= 56; nextw = load64_any(wp);
                bufB |= nextwB << c3B; wpB += (63 - c3B) >> 3; c3B |= 56; nextwB = load64_any(wpB);
#pragma unroll
                for (int step = 0; step < 2; step++) {
                    const u32 t = (u32)buf & st;
                    const u64 ent = *(lds_cu64p)(uintptr_t)((t & 0xffffu) + (st >> 16));
                    const u32 tB = (u32)bufB & stB;
                    const u64 entB = *(lds_cu64p)(uintptr_t)((tB & 0xffffu) + (stB >> 16));
                    {
                        const u32 e1 = (u32)ent; st = (u32)(ent >> 32);
                        const u32 take = e1 & 63 & am;
                        buf >>= take; c3 -= take; rel += take;
                        const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e1, 14u, 1u) & am;
                        atomicOr(&mrow[(rel >> 5) << 6], m_tok & (1u << (rel & 31)));
                        const u32 over = S1 - rel;
                        const u32 fin = (u32)((i32)(((over & m_tok) | e1) & am) >> 31);
                        e = (fin & e1) | (~fin & e);
                        am &= ~fin;
                        steps += am & 1;
                    }
                    {
                        const u32 e1 = (u32)entB; stB = (u32)(entB >> 32);
                        const u32 take = e1 & 63 & amB;
                        bufB >>= take; c3B -= take; relB += take;
                        const u32 m_tok = (u32)__builtin_amdgcn_sbfe((int)e1, 14u, 1u) & amB;
                        atomicOr(&mrowB[(relB >> 5) << 6], m_tok & (1u << (relB & 31)));
                        const u32 over = S1 - relB;
                        const u32 fin = (u32)((i32)(((over & m_tok) | e1) & amB) >> 31);
                        eB = (fin & e1) | (~fin & eB);
                        amB &= ~fin;
                        steps += amB & 1;
                    }
                }
                if (__ballot((am | amB) != 0) == 0) break;
            }
            acc += eB + relB;
        }
    out:
        acc += e + rel + (u32)buf;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = acc + steps;
    if (lane == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = steps; }
}

template <int X>
static void run(const char *name)
{
    const int iters = 2000;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%-40s", name);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = cus * 4 * wps;
        u32 *out; unsigned long long *cyc;
        (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 16);
        hipLaunchKernelGGL(k<X>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc, 1u);
        hipLaunchKernelGGL(k<X>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc, 2u);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost);
        double mean = 0, st = 0; for (int i = 0; i < blocks; i++) { mean += h[2 * i]; st += h[2 * i + 1]; }
        mean /= blocks; st /= blocks;
        // lane 0's step count stands for the wave's loop length only roughly: report per lane-0 step
        printf("  w/SIMD %d: %6.1f ticks/step/wave %6.1f /SIMD", wps, mean / st, mean / st / wps);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
    run<1>("X1 exit test every step (compiler)");
    run<2>("X2 exit test every other step");
    run<3>("X3 uniform loop, masked updates");
    run<4>("X4 asm loop, v_cmpx");
    run<5>("X5 two chains per lane, masked");
    return 0;
}
