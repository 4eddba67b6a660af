// How pinned uploads and downloads on different streams share the copy engines (host path pipeline):
//  1. H2D alone, D2H alone, both at once
//  2. a D2H enqueued behind a 3 ms kernel on stream A, then an H2D on stream B: does the upload wait for the kernel?
// build: hipcc -O2 --offload-arch=gfx950 -o copy_overlap copy_overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(long long ticks, int *out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) *out = 1;
}
int main() {
    const size_t piece = 16u << 20; const int np = 16;
    void *h_a, *h_b, *d_a, *d_b; int *d_flag;
    CK(hipHostMalloc(&h_a, piece * np, hipHostMallocDefault)); CK(hipHostMalloc(&h_b, piece * np, hipHostMallocDefault));
    CK(hipMalloc(&d_a, piece * np)); CK(hipMalloc(&d_b, piece * np)); CK(hipMalloc(&d_flag, 4));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        for (int i = 0; i < np; i++) CK(hipMemcpyAsync((char *)d_a + i * piece, (char *)h_a + i * piece, piece, hipMemcpyHostToDevice, sa));
        CK(hipStreamSynchronize(sa));
        double t1 = now();
        for (int i = 0; i < np; i++) CK(hipMemcpyAsync((char *)h_b + i * piece, (char *)d_b + i * piece, piece, hipMemcpyDeviceToHost, sb));
        CK(hipStreamSynchronize(sb));
        double t2 = now();
        for (int i = 0; i < np; i++) {
            CK(hipMemcpyAsync((char *)d_a + i * piece, (char *)h_a + i * piece, piece, hipMemcpyHostToDevice, sa));
            CK(hipMemcpyAsync((char *)h_b + i * piece, (char *)d_b + i * piece, piece, hipMemcpyDeviceToHost, sb));
        }
        CK(hipStreamSynchronize(sa)); double t3a = now(); CK(hipStreamSynchronize(sb));
        double t3 = now();
        const double gb = piece * np / 1e6;
        printf("H2D alone %.1f GB/s   D2H alone %.1f GB/s   both at once: %.2f ms (H2D stream done at %.2f) = %.1f GB/s each way\n",
               gb / (t1 - t0), gb / (t2 - t1), t3 - t2, t3a - t2, gb / (t3 - t2));
    }
    // wall_clock64 runs at 100 MHz: 3 ms = 300000 ticks
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        spin<<<1, 64, 0, sa>>>(300000, d_flag);
        CK(hipMemcpyAsync(h_b, d_b, piece, hipMemcpyDeviceToHost, sa));
        CK(hipMemcpyAsync(d_a, h_a, piece, hipMemcpyHostToDevice, sb));
        CK(hipStreamSynchronize(sb)); double t1 = now();
        CK(hipStreamSynchronize(sa)); double t2 = now();
        printf("3 ms kernel + D2H on A, H2D on B: H2D done after %.2f ms, A done after %.2f ms\n", t1 - t0, t2 - t0);
    }
    // the same with the D2H on A held back until the kernel is known to be done
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        spin<<<1, 64, 0, sa>>>(300000, d_flag);
        CK(hipMemcpyAsync(d_a, h_a, piece, hipMemcpyHostToDevice, sb));
        CK(hipStreamSynchronize(sb)); double t1 = now();
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpyAsync(h_b, d_b, piece, hipMemcpyDeviceToHost, sa));
        CK(hipStreamSynchronize(sa)); double t2 = now();
        printf("3 ms kernel on A, H2D on B, D2H on A after the kernel: H2D done after %.2f ms, A done after %.2f ms\n", t1 - t0, t2 - t0);
    }
    return 0;
}
