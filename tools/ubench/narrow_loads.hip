// Calibration of rocprofv3's FETCH_SIZE for narrow scattered loads (the match-source reads of resolve_tokens):
// every lane loads 8 bytes at a pseudo-random 8-byte-aligned offset of a buffer much larger than L2 + Infinity Cache.
// usage: narrow_loads <buffer MiB> <loads per lane> [stride mode: 0 random, 1 coalesced 16 B per lane]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

__global__ void __launch_bounds__(64) narrow(const uint64_t *buf, uint64_t nwords, int loads, uint64_t *out)
{
    uint64_t x = (blockIdx.x * 64ull + threadIdx.x) * 0x9e3779b97f4a7c15ull + 12345;
    uint64_t acc = 0;
    for (int i = 0; i < loads; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        acc += buf[x % nwords];
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(64) wide(const uint4 *buf, uint64_t nvec, int loads, uint64_t *out)
{
    uint64_t acc = 0;
    uint64_t i = blockIdx.x * 64ull + threadIdx.x;
    for (int k = 0; k < loads; k++, i += (uint64_t)gridDim.x * 64) { const uint4 v = buf[i % nvec]; acc += v.x + v.w; }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 4096;
    const int loads = argc > 2 ? atoi(argv[2]) : 64;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const size_t bytes = mib << 20;
    void *buf; uint64_t *out;
    const int blocks = 256 * 16;
    (void)hipMalloc(&buf, bytes); (void)hipMemset(buf, 1, bytes);
    (void)hipMalloc(&out, blocks * 64 * 8);
    if (mode == 0) hipLaunchKernelGGL(narrow, dim3(blocks), dim3(64), 0, 0, (const uint64_t *)buf, bytes / 8, loads, out);
    else hipLaunchKernelGGL(wide, dim3(blocks), dim3(64), 0, 0, (const uint4 *)buf, bytes / 16, loads, out);
    (void)hipDeviceSynchronize();
    const double n = (double)blocks * 64 * loads;
    printf("mode %d: %.0f lane-loads of %d bytes = %.0f KB requested; distinct 64-B lines <= %.0f KB, 128-B lines <= %.0f KB\n",
           mode, n, mode ? 16 : 8, n * (mode ? 16 : 8) / 1024, mode ? n * 16 / 1024 : n * 64 / 1024, mode ? n * 16 / 1024 : n * 128 / 1024);
    return 0;
}
