// Microbenchmark: issue cost of single VALU instruction kinds (8 independent chains, 4 waves per SIMD).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define BODY8(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                 : "v"(x), "v"(y), "s"(sx) : "vcc")

#define I_ADD(n)     "v_add_u32_e32 %" #n ", %8, %" #n "\n"
#define I_AND(n)     "v_and_b32_e32 %" #n ", %8, %" #n "\n"
#define I_LSHR(n)    "v_lshrrev_b32_e32 %" #n ", 1, %" #n "\n"
#define I_BFEC(n)    "v_bfe_u32 %" #n ", %" #n ", 3, 29\n"
#define I_BFEV(n)    "v_bfe_u32 %" #n ", %" #n ", 0, %8\n"
#define I_BFEI(n)    "v_bfe_i32 %" #n ", %" #n ", 1, 31\n"
#define I_BITOP3(n)  "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0xca\n"
#define I_BITOP3C(n) "v_bitop3_b32 %" #n ", %" #n ", 8, 9 bitop3:0xca\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_ANDOR(n)   "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD3(n)    "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_CNDMASK(n) "v_cndmask_b32_e32 %" #n ", %8, %" #n ", vcc\n"
#define I_XAD(n)     "v_xad_u32 %" #n ", %" #n ", -1, %8\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 3\n"
#define I_ADDS(n)    "v_add_u32_e32 %" #n ", %10, %" #n "\n"
#define I_MOVDPP(n)  "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CMP(n)     "v_cmp_lt_i32_e32 vcc, -1, %" #n "\n"
#define I_LSHLV(n)   "v_lshlrev_b32_e32 %" #n ", %8, %" #n "\n"
#define I_SUBREV(n)  "v_subrev_u32_e32 %" #n ", %8, %" #n "\n"
#define I_MADU24(n)  "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_PERM(n)    "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_OR3(n)     "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n)  "v_lshl_or_b32 %" #n ", %" #n ", 2, %8\n"
#define I_ADDE64(n)  "v_add_u32_e64 %" #n ", %" #n ", %8\n"
#define I_BFEVV(n)   "v_bfe_u32 %" #n ", %" #n ", %9, %8\n"

template <int KIND>
__global__ void __launch_bounds__(64) k(uint32_t *out, int iters, unsigned long long *cyc)
{
    uint32_t r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * (i + 3) + 12345;
    uint32_t x = threadIdx.x | 1, y = threadIdx.x * 7 + 3;
    uint32_t sx = iters;
    uint64_t q[4];
    for (int i = 0; i < 4; i++) q[i] = (uint64_t)threadIdx.x * 0x9e3779b97f4a7c15ull + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == 0) BODY8(I_ADD);
            if (KIND == 1) BODY8(I_AND);
            if (KIND == 2) BODY8(I_LSHR);
            if (KIND == 3) BODY8(I_BFEC);
            if (KIND == 4) BODY8(I_BFEV);
            if (KIND == 5) BODY8(I_BFEI);
            if (KIND == 6) BODY8(I_BITOP3);
            if (KIND == 7) BODY8(I_BITOP3C);
            if (KIND == 8) BODY8(I_LSHLADD);
            if (KIND == 9) BODY8(I_ANDOR);
            if (KIND == 10) BODY8(I_ADD3);
            if (KIND == 11) BODY8(I_CNDMASK);
            if (KIND == 12) BODY8(I_XAD);
            if (KIND == 13) BODY8(I_ALIGNBIT);
            if (KIND == 14) BODY8(I_ADDS);
            if (KIND == 15) BODY8(I_MOVDPP);
            if (KIND == 16) BODY8(I_CMP);
            if (KIND == 17) BODY8(I_LSHLV);
            if (KIND == 18) BODY8(I_SUBREV);
            if (KIND == 19) BODY8(I_MADU24);
            if (KIND == 20) BODY8(I_PERM);
            if (KIND == 21) BODY8(I_OR3);
            if (KIND == 22) BODY8(I_LSHLOR);
            if (KIND == 23) BODY8(I_ADDE64);
            if (KIND == 24) BODY8(I_BFEVV);
            if (KIND == 25) {
                asm volatile("v_lshrrev_b64 %0, %4, %0\nv_lshrrev_b64 %1, %4, %1\nv_lshrrev_b64 %2, %4, %2\nv_lshrrev_b64 %3, %4, %3\n"
                             "v_lshrrev_b64 %0, %4, %0\nv_lshrrev_b64 %1, %4, %1\nv_lshrrev_b64 %2, %4, %2\nv_lshrrev_b64 %3, %4, %3\n"
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(x));
            }
            if (KIND == 26) {
                asm volatile("v_lshlrev_b64 %0, %4, %0\nv_lshlrev_b64 %1, %4, %1\nv_lshlrev_b64 %2, %4, %2\nv_lshlrev_b64 %3, %4, %3\n"
                             "v_lshlrev_b64 %0, %4, %0\nv_lshlrev_b64 %1, %4, %1\nv_lshlrev_b64 %2, %4, %2\nv_lshlrev_b64 %3, %4, %3\n"
                             : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(x));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += r[i];
    for (int i = 0; i < 4; i++) s += (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name)
{
    const int iters = 20000, per_iter = 32;
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%-26s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * 4 * wps;
        uint32_t *out; unsigned long long *cyc;
        (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= blocks;
        const double ops = (double)per_iter * iters;
        printf("  w/SIMD %d: %5.2f ticks/op/wave %5.2f ticks/op/SIMD", wps, mean / ops, mean / ops / wps);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
    run<0>("v_add_u32 (VOP2)");
    run<1>("v_and_b32 (VOP2)");
    run<2>("v_lshrrev_b32 imm");
    run<17>("v_lshlrev_b32 vgpr");
    run<18>("v_subrev_u32");
    run<23>("v_add_u32_e64");
    run<14>("v_add_u32 sgpr");
    run<3>("v_bfe_u32 v,imm,imm");
    run<4>("v_bfe_u32 v,0,v");
    run<24>("v_bfe_u32 v,v,v");
    run<5>("v_bfe_i32 v,imm,imm");
    run<6>("v_bitop3 v,v,v");
    run<7>("v_bitop3 v,8,9");
    run<8>("v_lshl_add_u32 v,2,v");
    run<22>("v_lshl_or_b32 v,2,v");
    run<9>("v_and_or_b32 v,v,v");
    run<10>("v_add3_u32 v,v,v");
    run<21>("v_or3_b32 v,v,v");
    run<19>("v_mad_u32_u24 v,v,v");
    run<20>("v_perm_b32 v,v,v");
    run<11>("v_cndmask_b32 vcc");
    run<12>("v_xad_u32 v,-1,v");
    run<13>("v_alignbit v,v,3");
    run<15>("v_mov_b32_dpp row_shr");
    run<16>("v_cmp_lt_i32 vcc");
    run<25>("v_lshrrev_b64");
    run<26>("v_lshlrev_b64");
    return 0;
}
