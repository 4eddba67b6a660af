// xfg_api.cpp -- device context and the batch-decode entry points of include/xflate_amd.h.
//
// Everything here is transport: HIP device, stream, staging buffers, argument
// checks.  The arithmetic lives in inflate_kernel.hip; the reference-shaped
// host logic (index, Reader state machine) lives in host_reader.cpp.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>
#include <string>
#include <vector>

#include "../../include/xflate_amd.h"
#include "xfg_internal.h"

static thread_local std::string g_last_error;

static int hip_fail(hipError_t e, const char *what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    g_last_error = buf;
    return XFG_E_HIP;
}
// XFG_TRACE=1 in the environment prints every HIP call of this file to stderr.
static bool trace_on() {
    static const bool on = [] { const char *e = getenv("XFG_TRACE"); return e && *e && *e != '0'; }();   // initialised once, thread-safe
    return on;
}
// Every entry point makes its context's device current for the call and puts the caller's device back
// on every way out (a process that drives several GPUs must not find its current device changed).
struct DeviceGuard {
    int prev;
    bool restore;
    DeviceGuard() : prev(-1), restore(false) {}
    hipError_t enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) return hipSuccess;
        const hipError_t e = hipSetDevice(device);
        restore = e == hipSuccess && prev >= 0;
        return e;
    }
    ~DeviceGuard() { if (restore) (void)hipSetDevice(prev); }
};
#define HIP_TRY(expr) do { \
        if (trace_on()) fprintf(stderr, "[xfg] %s\n", #expr); \
        hipError_t e_ = (expr); \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

extern "C" const char *xfg_last_error(void) { return g_last_error.c_str(); }
extern "C" void xfg_set_last_error(const char *text) { g_last_error = text ? text : ""; }

extern "C" const char *xfg_strerror(int code) {
    switch (code) {
    case XFG_OK: return "ok";
    case XFG_E_ARG: return "bad argument";
    case XFG_E_HIP: return "HIP runtime error";
    case XFG_E_NODEVICE: return "no usable GPU device";
    case XFG_E_NOMEM: return "out of memory or output buffer too small";
    case XFG_E_LIMIT: return "chunk exceeds the per-chunk size limit";
    default: return "unknown error";
    }
}

extern "C" int xfg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

struct xfg_ctx {
    int device;
    int num_cu;
    hipStream_t stream;
    void *d_scratch;      // the chunk counters of the persistent grid (xfg_scratch_bytes: 2 KB, allocated once)
    // reusable device staging for the host-pointer path
    void *d_comp; size_t comp_cap;
    void *d_raw; size_t raw_cap;
    void *d_chunks; size_t chunks_cap;
    void *d_res; size_t res_cap;
};

static int grow(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return XFG_OK;
    if (*p) { hipError_t e = hipFree(*p); *p = nullptr; *cap = 0; if (e != hipSuccess) return hip_fail(e, "hipFree"); }
    size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) { *p = nullptr; return hip_fail(e, "hipMalloc"); }
    *cap = want;
    return XFG_OK;
}

extern "C" int xfg_create(int device, xfg_ctx **out) {
    if (!out) return XFG_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { hip_fail(e, "hipGetDeviceCount"); return XFG_E_NODEVICE; }
    if (device < 0 || device >= n) return XFG_E_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.enter(device));
    xfg_ctx *c = new (std::nothrow) xfg_ctx();
    if (!c) return XFG_E_NOMEM;
    memset(c, 0, sizeof *c);
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { xfg_destroy(c); return hip_fail(e, "hipGetDeviceProperties"); }
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e = hipMalloc(&c->d_scratch, xfg_scratch_bytes(c->num_cu));
    if (e != hipSuccess) { xfg_destroy(c); return hip_fail(e, "hipMalloc(scratch)"); }
    *out = c;
    return XFG_OK;
}

extern "C" void xfg_destroy(xfg_ctx *c) {
    if (!c) return;
    DeviceGuard guard;
    (void)guard.enter(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_comp) (void)hipFree(c->d_comp);
    if (c->d_raw) (void)hipFree(c->d_raw);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    if (c->d_res) (void)hipFree(c->d_res);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int xfg_sync(xfg_ctx *c) {
    if (!c) return XFG_E_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return XFG_OK;
}

// The host-pointer entry points check every descriptor before any launch and report XFG_E_LIMIT / XFG_E_ARG;
// on the device-resident path the same conditions are checked by the kernels, per chunk (XFG_ST_INTERNAL).
static int check_chunks(const xfg_chunk *chunks, size_t n, size_t comp_len, size_t raw_len) {
    for (size_t i = 0; i < n; i++) {
        const xfg_chunk &c = chunks[i];
        if (c.comp_size > XFG_MAX_CHUNK_COMP || c.raw_size > XFG_MAX_CHUNK_RAW) return XFG_E_LIMIT;
        if (c.comp_off > comp_len || c.comp_size > comp_len - c.comp_off) return XFG_E_ARG;
        if (c.raw_off > raw_len || c.raw_size > raw_len - c.raw_off) return XFG_E_ARG;
        if (c.reserved != 0) return XFG_E_ARG;
    }
    return XFG_OK;
}

extern "C" int xfg_decode_device(xfg_ctx *c, const void *d_comp, size_t comp_len,
                                 const void *d_chunks, size_t n,
                                 void *d_raw, size_t raw_len, void *d_res, void *hip_stream)
{
    if (!c) return XFG_E_ARG;
    if (n == 0) return XFG_OK;
    if (!d_comp || !d_chunks || !d_res || (!d_raw && raw_len)) return XFG_E_ARG;
    if (((uintptr_t)d_comp & 15) != 0) return XFG_E_ARG;
    if (n > (1ull << 30)) return XFG_E_ARG;                    // the kernel counts to 3 n in 32 bits
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // One launch at a time per context: the chunk counters are per context.
    HIP_TRY(xfg_launch_inflate(d_comp, comp_len, d_chunks, (uint32_t)n, d_raw, raw_len, d_res, c->d_scratch, c->num_cu, s));
    return XFG_OK;
}

extern "C" int xfg_decode(xfg_ctx *c, const uint8_t *comp, size_t comp_len,
                          const xfg_chunk *chunks, size_t n,
                          uint8_t *raw, size_t raw_len, xfg_result *res)
{
    if (!c || (!chunks && n) || (!res && n) || (!comp && comp_len) || (!raw && raw_len)) return XFG_E_ARG;
    if (n == 0) return XFG_OK;
    int rc = check_chunks(chunks, n, comp_len, raw_len);
    if (rc != XFG_OK) return rc;
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    if ((rc = grow(&c->d_comp, &c->comp_cap, comp_len + 16)) != XFG_OK) return rc;
    if ((rc = grow(&c->d_raw, &c->raw_cap, raw_len + 16)) != XFG_OK) return rc;
    if ((rc = grow(&c->d_chunks, &c->chunks_cap, n * sizeof(xfg_chunk))) != XFG_OK) return rc;
    if ((rc = grow(&c->d_res, &c->res_cap, n * sizeof(xfg_result))) != XFG_OK) return rc;
    if (comp_len) HIP_TRY(hipMemcpyAsync(c->d_comp, comp, comp_len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_chunks, chunks, n * sizeof(xfg_chunk), hipMemcpyHostToDevice, c->stream));
    rc = xfg_decode_device(c, c->d_comp, comp_len, c->d_chunks, n, c->d_raw, raw_len, c->d_res, nullptr);
    if (rc != XFG_OK) return rc;
    HIP_TRY(hipMemcpyAsync(res, c->d_res, n * sizeof(xfg_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // Copy back only what each chunk produced (chunks own disjoint output ranges); chunks whose
    // outputs follow one another without a gap travel in one copy.
    for (size_t i = 0; i < n;) {
        const size_t lo = (size_t)chunks[i].raw_off;
        size_t hi = lo + (size_t)res[i].out_produced;
        size_t j = i + 1;
        while (j < n && (size_t)chunks[j].raw_off == hi &&
               (size_t)res[j - 1].out_produced == (size_t)chunks[j - 1].raw_size) {
            hi += (size_t)res[j].out_produced;
            j++;
        }
        if (hi > lo) HIP_TRY(hipMemcpyAsync(raw + lo, (const uint8_t *)c->d_raw + lo, hi - lo, hipMemcpyDeviceToHost, c->stream));
        i = j;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return XFG_OK;
}

extern "C" int xfg_crc32_device(xfg_ctx *c, const void *d_raw, size_t raw_len, const void *d_chunks, size_t n,
                                const void *d_res, void *d_crc, void *hip_stream)
{
    if (!c) return XFG_E_ARG;
    if (n == 0) return XFG_OK;
    if (!d_raw || !d_chunks || !d_crc) return XFG_E_ARG;
    if (n > 0xffffffffull) return XFG_E_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HIP_TRY(xfg_launch_crc32(d_raw, raw_len, d_chunks, d_res, (uint32_t)n, d_crc, c->num_cu, s));
    return XFG_OK;
}

extern "C" size_t xfg_deflate_bound(size_t raw_size)
{
    // the stored coding (5 bytes per 65535, 5 for the sync flush) is the worst case, plus the writer's slack
    const size_t stored = raw_size + 5 * ((raw_size + 65534) / 65535) + 5;
    return (stored + 8 + 3) & ~(size_t)3;
}

extern "C" int xfg_deflate_device(xfg_ctx *c, const void *d_raw, size_t raw_len,
                                  const void *d_chunks, size_t n,
                                  void *d_comp, size_t comp_len, void *d_res, void *hip_stream)
{
    if (!c) return XFG_E_ARG;
    if (n == 0) return XFG_OK;
    if (!d_chunks || !d_comp || !d_res || (!d_raw && raw_len)) return XFG_E_ARG;
    if (((uintptr_t)d_comp & 3) != 0 || n > 0xffffffffull) return XFG_E_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    HIP_TRY(xfg_launch_deflate(d_raw, raw_len, d_chunks, (uint32_t)n, d_comp, comp_len, d_res, c->num_cu, s));
    return XFG_OK;
}

extern "C" int xfg_deflate(xfg_ctx *c, const uint8_t *raw, size_t raw_len,
                           const xfg_chunk *chunks, size_t n,
                           uint8_t *comp, size_t comp_len, xfg_result *res)
{
    if (!c || (!chunks && n) || (!res && n) || (!raw && raw_len) || (!comp && comp_len)) return XFG_E_ARG;
    if (n == 0) return XFG_OK;
    int rc = check_chunks(chunks, n, comp_len, raw_len);        // the same range checks, the other way round
    if (rc != XFG_OK) return rc;
    for (size_t i = 0; i < n; i++) if (chunks[i].comp_off & 3) return XFG_E_ARG;
    DeviceGuard guard;
    HIP_TRY(guard.enter(c->device));
    if ((rc = grow(&c->d_comp, &c->comp_cap, comp_len + 16)) != XFG_OK) return rc;
    if ((rc = grow(&c->d_raw, &c->raw_cap, raw_len + 16)) != XFG_OK) return rc;
    if ((rc = grow(&c->d_chunks, &c->chunks_cap, n * sizeof(xfg_chunk))) != XFG_OK) return rc;
    if ((rc = grow(&c->d_res, &c->res_cap, n * sizeof(xfg_result))) != XFG_OK) return rc;
    if (raw_len) HIP_TRY(hipMemcpyAsync(c->d_raw, raw, raw_len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_chunks, chunks, n * sizeof(xfg_chunk), hipMemcpyHostToDevice, c->stream));
    rc = xfg_deflate_device(c, c->d_raw, raw_len, c->d_chunks, n, c->d_comp, comp_len, c->d_res, nullptr);
    if (rc != XFG_OK) return rc;
    HIP_TRY(hipMemcpyAsync(res, c->d_res, n * sizeof(xfg_result), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        const size_t k = (size_t)res[i].out_produced;
        if (res[i].status == XFG_ST_OK && k)
            HIP_TRY(hipMemcpyAsync(comp + chunks[i].comp_off, (const uint8_t *)c->d_comp + chunks[i].comp_off, k,
                                   hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return XFG_OK;
}

extern "C" int xfg_flate_decode(xfg_ctx *c, const uint8_t *in, size_t in_len,
                                uint8_t *out, size_t out_cap,
                                int64_t *in_off, int64_t *out_off, int *err)
{
    if (!c || !in_off || !out_off || !err || (!in && in_len) || (!out && out_cap)) return XFG_E_ARG;
    xfg_chunk ch;
    memset(&ch, 0, sizeof ch);
    ch.comp_off = 0; ch.comp_size = in_len; ch.raw_off = 0; ch.raw_size = out_cap;
    ch.flags = 0;                       // plain stream, in-repo flate package rules
    xfg_result r;
    int rc = xfg_decode(c, in, in_len, &ch, 1, out, out_cap, &r);
    if (rc != XFG_OK) return rc;
    *in_off = r.in_consumed;
    *out_off = r.out_produced;
    switch (r.status) {
    case XFG_ST_OK: *err = XFG_ERR_NONE; break;
    case XFG_ST_CORRUPTED: *err = XFG_ERR_CORRUPTED; break;
    case XFG_ST_UNEXPECTED_EOF: *err = XFG_ERR_UNEXPECTED_EOF; break;
    case XFG_ST_OUT_FULL: *err = XFG_ERR_NONE; return XFG_E_NOMEM;
    default: *err = XFG_ERR_INTERNAL; break;
    }
    return XFG_OK;
}
