// xfg_api.cpp -- device context and the batch-decode entry points of include/xflate_amd.h.
//
// Everything here is transport: HIP device, stream, staging buffers, argument
// checks.  The arithmetic lives in inflate_kernel.hip; the reference-shaped
// host logic (index, Reader state machine) lives in host_reader.cpp.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
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

// A few host threads that copy between the caller's pageable buffers and pinned staging (one thread moves
// about 25 GB/s on the GPU box, four about 70) and issue the blocking pageable uploads.
struct HostPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv, cv_idle;
    std::vector<std::function<void()> > q;
    size_t running;
    bool stop;
    // Throws (std::system_error, std::bad_alloc) if not even one thread can be started; the threads that were
    // started before a failure are kept (make_pool() below is the only caller and catches).
    explicit HostPool(int n) : running(0), stop(false) {
        try {
            threads.reserve((size_t)n);
            for (int i = 0; i < n; i++) threads.emplace_back([this] { work(); });
        } catch (...) {
            if (threads.empty()) throw;                         // (a vector of joinable threads must not be destroyed: none here)
        }
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
    void work() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [this] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.back());
                q.pop_back();
                running++;
            }
            f();
            { std::lock_guard<std::mutex> l(m); running--; }
            cv_idle.notify_all();
        }
    }
    // false: the task could not be queued (out of memory) and was not run
    bool submit(std::function<void()> f) noexcept {
        try {
            { std::lock_guard<std::mutex> l(m); q.push_back(std::move(f)); }
        } catch (...) {
            return false;
        }
        cv.notify_one();
        return true;
    }
};
// No exception may cross the extern "C" entry points: thread creation fails with std::system_error when the
// process is at its thread limit (the GPU box grants 16 CPUs per GPU), allocation with std::bad_alloc.
static HostPool *make_pool(int n) noexcept {
    try {
        return new HostPool(n);
    } catch (...) {
        return nullptr;
    }
}

struct xfg_ctx {
    int device;
    int num_cu;
    hipStream_t stream;
    // the pipelined host path (decode_pipelined): a second stream for the downloads, pinned staging slots, helper threads
    hipStream_t stream_k[2];              // the lanes: [0] is `stream`, the other is the pipeline's own
    hipStream_t stream_up, stream_dl;     // uploads and downloads, one stream each
    void *d_scratch_k;                    // chunk counters for each lane
    void *h_slot[8];                      // pinned download slots (a slice's output)
    void *h_res[8];                       // pinned per-slot copies of the slice's results
    void *h_up[4];                        // pinned upload slots (a slice's compressed bytes)
    HostPool *pool;
    int pipe_setup_failures;              // decode_pipelined: set-ups that could not get their pinned memory, streams or threads
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
    delete c->pool;
    for (int i = 0; i < 8; i++) if (c->h_slot[i]) (void)hipHostFree(c->h_slot[i]);
    for (int i = 0; i < 8; i++) if (c->h_res[i]) (void)hipHostFree(c->h_res[i]);
    for (int i = 0; i < 4; i++) if (c->h_up[i]) (void)hipHostFree(c->h_up[i]);
    if (c->stream_k[1]) (void)hipStreamDestroy(c->stream_k[1]);
    if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
    if (c->stream_dl) (void)hipStreamDestroy(c->stream_dl);
    if (c->d_scratch_k) (void)hipFree(c->d_scratch_k);
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

// ---------------------------------------------------------------------------
// The host-pointer path, pipelined.  A batch whose chunks lie in stream order (ascending, disjoint input and output
// ranges: what an index gives) is cut into slices of at most 16 MiB of output, the transfer units, and runs of slices
// into launch units (32 MiB at first, then up to 80).  A wave needs about 2 ms for a 64 KiB chunk whatever else runs,
// so a launch is latency-bound until it holds thousands of chunks: the units' kernels alternate between kLanes
// streams, each with chunk counters of its own, and two of them run side by side.  All uploads go through pinned
// slots on one stream and all downloads into pinned slots on another (57 GB/s each way, against 31 and 18 GB/s
// with pageable memory), so that each direction keeps a copy engine to itself; a download is enqueued only when its
// kernel is known to be done, because one that waits in an engine's queue holds up what is behind it.  That makes
// four streams, one per hardware queue of the process (GPU_MAX_HW_QUEUES): streams that share a queue run in turn.
// A pool thread (the feeder) copies the compressed bytes into the upload slots and enqueues uploads and kernels;
// the calling thread enqueues the downloads as slots fall free and, with the pool, copies what each chunk produced
// from the slot into the caller's buffer.  Measured on C2 (tools/host_path_rate.py): 33 GB/s on 256 MiB (2.6 ms until
// the first kernel is done, then the downloads at PCIe rate), 42 GB/s on 1 GiB; the plain path below gives 11 GB/s.
// Results are identical to the plain path's (tests/test_gpu_host_pipeline.py).
// ---------------------------------------------------------------------------
namespace {
const size_t kSliceRaw = 16u << 20;       // output (and input) bytes per slice at most: the size of a pinned slot
const size_t kUnitRawMax = 80u << 20;     // output bytes per launch unit, about
const size_t kPipelineMin = 64u << 20;    // smaller batches take the plain path
const int kLanes = 2;                     // streams whose kernels run side by side (see above)
const int kUpSlots = 4;
const int kDownSlots = 8;
const int kCopyThreads = 8;
const size_t kSliceChunks = 65536;        // chunks per slice at most (sizes the pinned result slots)
const int kSetupTries = 3;                // failed set-ups of the pipeline's parts before a context stops trying

struct Slice { size_t a, b; size_t comp_lo, comp_hi, raw_lo, raw_hi; size_t unit; };
struct Unit { size_t s0, s1; };           // slices [s0, s1)

bool stream_ordered(const xfg_chunk *ch, size_t n) {
    for (size_t i = 1; i < n; i++)
        if (ch[i].comp_off < ch[i - 1].comp_off + ch[i - 1].comp_size || ch[i].raw_off < ch[i - 1].raw_off + ch[i - 1].raw_size) return false;
    return true;
}

// copy [lo, hi) of src to dst on the pool's threads and this one
void parallel_copy(HostPool *pool, uint8_t *dst, const uint8_t *src, size_t len) {
    if (len < (1u << 20)) { memcpy(dst, src, len); return; }
    const size_t per = (len / kCopyThreads + 4095) & ~(size_t)4095;
    std::atomic<int> left(0);
    std::mutex m; std::condition_variable cv;
    size_t off = per;                                           // the first piece is this thread's
    for (; off < len; off += per) {
        const size_t k = std::min(per, len - off);
        left++;
        // (notify under the lock: cv lives on the waiter's stack; a piece that cannot be queued is copied here)
        bool queued = false;
        try {
            queued = pool->submit([=, &left, &m, &cv] { memcpy(dst + off, src + off, k); std::lock_guard<std::mutex> l(m); left--; cv.notify_one(); });
        } catch (...) {                                         // (constructing the std::function may allocate)
        }
        if (!queued) { memcpy(dst + off, src + off, k); left--; }
    }
    memcpy(dst, src, std::min(per, len));
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] { return left.load() == 0; });
}
}

// The events of one pipelined call: destroyed together, whichever way the call ends.
struct EventBag {
    std::vector<hipEvent_t> all;
    ~EventBag() { for (hipEvent_t e : all) (void)hipEventDestroy(e); }
    hipError_t make(hipEvent_t *out, unsigned flags) {
        hipError_t e = hipEventCreateWithFlags(out, flags);
        if (e == hipSuccess) all.push_back(*out);
        return e;
    }
};

// A large host-to-host copy on the context's helper threads (the Reader's copy into the caller's buffer).
extern "C" void xfg_host_copy(xfg_ctx *c, void *dst, const void *src, size_t n) {
    if (c && n >= (4u << 20) && !c->pool) c->pool = make_pool(kCopyThreads + 1);
    if (!c || !c->pool || n < (4u << 20)) { memcpy(dst, src, n); return; }
    parallel_copy(c->pool, (uint8_t *)dst, (const uint8_t *)src, n);
}

static int decode_pipelined(xfg_ctx *c, const uint8_t *comp, size_t comp_len, const xfg_chunk *chunks, size_t n,
                            uint8_t *raw, size_t raw_len, xfg_result *res)
{
    // ---- slices and launch units
    std::vector<Slice> sl;
    for (size_t i = 0; i < n;) {
        Slice s; s.a = i; s.comp_lo = (size_t)chunks[i].comp_off; s.raw_lo = (size_t)chunks[i].raw_off; s.unit = 0;
        size_t j = i;
        while (j < n && j - i < kSliceChunks && (size_t)(chunks[j].raw_off + chunks[j].raw_size) - s.raw_lo <= kSliceRaw &&
               (size_t)(chunks[j].comp_off + chunks[j].comp_size) - s.comp_lo <= kSliceRaw) j++;
        if (j == i) return -1;                                  // one chunk larger than a slot: the plain path takes it
        s.b = j; s.comp_hi = (size_t)(chunks[j - 1].comp_off + chunks[j - 1].comp_size); s.raw_hi = (size_t)(chunks[j - 1].raw_off + chunks[j - 1].raw_size);
        sl.push_back(s);
        i = j;
    }
    const size_t ns = sl.size();
    std::vector<Unit> units;
    {
        // short first units, so that the downloads start early
        const size_t total = sl.back().raw_hi - sl.front().raw_lo;
        size_t s0 = 0, in = 0;
        for (size_t i = 0; i < ns; i++) {
            const size_t u = units.size();
            const size_t want = u == 0 ? std::min(kUnitRawMax * 2 / 5, total / 8) : u == 1 ? std::min(kUnitRawMax * 4 / 5, total / 4)
                                                                                             : std::min(kUnitRawMax, total * 5 / 16);
            sl[i].unit = u;
            in += sl[i].raw_hi - sl[i].raw_lo;
            if (in + kSliceRaw / 2 >= want || i + 1 == ns) { units.push_back(Unit{s0, i + 1}); s0 = i + 1; in = 0; }
        }
    }
    const size_t nu = units.size();
    static const bool ptrace = getenv("XFG_PIPE_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double T0 = now();
    std::vector<double> t_up0(ns), t_up1(ns), t_dl(ns), t_f0(ns), t_f1(ns), t_f2(ns);
    // ---- one-time set-up of the context's pipeline parts.  Each part is made and checked on its own (a call that
    // failed half-way leaves the parts it made; the next call makes the rest).  The plain path needs none of them
    // (192 MiB of pinned host memory, three more streams, helper threads): a part that cannot be had sends this
    // call and, once it has failed kSetupTries times, every later one there.
    if (c->pipe_setup_failures >= kSetupTries) return -1;
    const int lanes = kLanes;
    {
        bool ok = true;
        auto need = [&](hipError_t e) { if (e != hipSuccess) { ok = false; (void)hipGetLastError(); } };
        for (int k = 0; k < kDownSlots && ok; k++) {
            if (!c->h_slot[k]) need(hipHostMalloc(&c->h_slot[k], kSliceRaw + 64, hipHostMallocDefault));
            if (ok && !c->h_res[k]) need(hipHostMalloc(&c->h_res[k], kSliceChunks * sizeof(xfg_result), hipHostMallocDefault));
        }
        for (int k = 0; k < kUpSlots && ok; k++)
            if (!c->h_up[k]) need(hipHostMalloc(&c->h_up[k], kSliceRaw + 64, hipHostMallocDefault));
        for (int k = 1; k < lanes && ok; k++)
            if (!c->stream_k[k]) need(hipStreamCreateWithFlags(&c->stream_k[k], hipStreamNonBlocking));
        c->stream_k[0] = c->stream;
        if (ok && !c->stream_up) need(hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking));
        if (ok && !c->stream_dl) need(hipStreamCreateWithFlags(&c->stream_dl, hipStreamNonBlocking));
        if (ok && !c->d_scratch_k) need(hipMalloc(&c->d_scratch_k, kLanes * xfg_scratch_bytes(c->num_cu)));
        if (ok && !c->pool) { c->pool = make_pool(kCopyThreads + 1); ok = c->pool != nullptr; }
        if (!ok) { c->pipe_setup_failures++; return -1; }
    }
    const size_t scratch_each = xfg_scratch_bytes(c->num_cu);
    EventBag bag;
    std::vector<hipEvent_t> ev_d(ns), ev_h(ns), ev_k(nu);
    hipEvent_t ev_c;
    for (size_t u = 0; u < nu; u++) HIP_TRY(bag.make(&ev_k[u], hipEventDisableTiming));
    HIP_TRY(bag.make(&ev_c, hipEventDisableTiming));
    for (size_t i = 0; i < ns; i++) {
        HIP_TRY(bag.make(&ev_d[i], hipEventDisableTiming));
        HIP_TRY(bag.make(&ev_h[i], hipEventDisableTiming));
    }
    std::vector<hipEvent_t> tk0(nu), tk1(nu), tk2(nu);          // XFG_PIPE_TRACE: where each unit was on the GPU's clock
    hipEvent_t tbase = nullptr;
    if (ptrace) {
        for (size_t u = 0; u < nu; u++) { HIP_TRY(bag.make(&tk0[u], hipEventDefault)); HIP_TRY(bag.make(&tk1[u], hipEventDefault)); HIP_TRY(bag.make(&tk2[u], hipEventDefault)); }
        HIP_TRY(bag.make(&tbase, hipEventDefault));
        HIP_TRY(hipEventRecord(tbase, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->d_chunks, chunks, n * sizeof(xfg_chunk), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(ev_c, c->stream));
    for (int k = 1; k < lanes; k++) HIP_TRY(hipStreamWaitEvent(c->stream_k[k], ev_c, 0));
    // ---- the feeder (a pool thread): slice after slice up, and after a unit's last slice its kernel on the next lane
    std::mutex m; std::condition_variable cv;
    size_t launched = 0;                                        // units whose kernel is enqueued
    hipError_t up_err = hipSuccess;
    bool up_done = false, stop = false;
    bool fed_ok = false;
    try {
    fed_ok = c->pool->submit([&] {
        (void)hipSetDevice(c->device);
        hipError_t e = hipSuccess;
        for (size_t u = 0; u < nu && e == hipSuccess; u++) {
            const int lane = (int)(u % lanes);
            hipStream_t ks = c->stream_k[lane];
            { std::lock_guard<std::mutex> l(m); if (stop) break; }
            if (ptrace) (void)hipEventRecord(tk0[u], c->stream_up);
            for (size_t i = units[u].s0; i < units[u].s1 && e == hipSuccess; i++) {
                const Slice &s = sl[i];
                t_up0[i] = now() - T0;
                if (s.comp_hi > s.comp_lo) {
                    if (i >= (size_t)kUpSlots) e = hipEventSynchronize(ev_h[i - kUpSlots]);    // the slot's last upload has landed
                    if (e == hipSuccess) {
                        parallel_copy(c->pool, (uint8_t *)c->h_up[i % kUpSlots], comp + s.comp_lo, s.comp_hi - s.comp_lo);
                        e = hipMemcpyAsync((uint8_t *)c->d_comp + s.comp_lo, c->h_up[i % kUpSlots], s.comp_hi - s.comp_lo, hipMemcpyHostToDevice, c->stream_up);
                    }
                }
                if (e == hipSuccess) e = hipEventRecord(ev_h[i], c->stream_up);
                t_up1[i] = now() - T0;
            }
            if (e == hipSuccess) {
                const size_t a = sl[units[u].s0].a, b = sl[units[u].s1 - 1].b;
                if (ptrace) (void)hipEventRecord(tk1[u], c->stream_up);
                e = hipStreamWaitEvent(ks, ev_h[units[u].s1 - 1], 0);
                if (e == hipSuccess)
                e = xfg_launch_inflate(c->d_comp, comp_len, (const xfg_chunk *)c->d_chunks + a, (uint32_t)(b - a), c->d_raw, raw_len,
                                       (xfg_result *)c->d_res + a, (uint8_t *)c->d_scratch_k + lane * scratch_each, c->num_cu, ks);
            }
            if (ptrace) (void)hipEventRecord(tk2[u], ks);
            if (e == hipSuccess) e = hipEventRecord(ev_k[u], ks);
            if (e == hipSuccess) { { std::lock_guard<std::mutex> l(m); launched = u + 1; } cv.notify_all(); }
        }
        std::lock_guard<std::mutex> l(m);                       // (notify under the lock: cv lives on the caller's stack)
        up_err = e; up_done = true;
        cv.notify_all();
    });
    } catch (...) {                                             // (the std::function's allocation)
    }
    if (!fed_ok) {
        // nothing but the chunk table's upload has been enqueued: let it land, then the plain path takes the call
        (void)hipStreamSynchronize(c->stream);
        return -1;
    }
    // ---- the calling thread: downloads as slots fall free, then slice j's bytes from its slot to the caller
    hipError_t herr = hipSuccess;
    size_t next_dl = 0;                                         // the next slice whose download is to be enqueued
    size_t kdone = 0;                                           // units whose kernel is known to be done
    for (size_t j = 0; j < ns && herr == hipSuccess; j++) {
        for (;;) {
            size_t ready, seen;
            bool fed;
            {
                std::unique_lock<std::mutex> l(m);
                if (next_dl <= j) cv.wait(l, [&] { return launched > sl[next_dl].unit || up_done; });
                seen = launched; fed = up_done;
            }
            // a download is enqueued only when its kernel is done: one that waits in a copy engine's queue holds up the
            // other lanes' uploads behind it
            while (kdone < seen && herr == hipSuccess) {
                hipError_t q = hipEventQuery(ev_k[kdone]);
                if (q == hipErrorNotReady && next_dl <= j && sl[next_dl].unit == kdone) q = hipEventSynchronize(ev_k[kdone]);
                if (q == hipSuccess) kdone++;
                else if (q == hipErrorNotReady) break;
                else herr = q;
            }
            ready = kdone;
            while (next_dl < ns && next_dl < j + kDownSlots && sl[next_dl].unit < ready && herr == hipSuccess) {
                const Slice &s = sl[next_dl];
                hipStream_t ks = c->stream_dl;
                const int slot = (int)(next_dl % kDownSlots);
                if (s.raw_hi > s.raw_lo)
                    herr = hipMemcpyAsync(c->h_slot[slot], (const uint8_t *)c->d_raw + s.raw_lo, s.raw_hi - s.raw_lo, hipMemcpyDeviceToHost, ks);
                if (herr == hipSuccess)
                    herr = hipMemcpyAsync(c->h_res[slot], (const xfg_result *)c->d_res + s.a, (s.b - s.a) * sizeof(xfg_result), hipMemcpyDeviceToHost, ks);
                if (herr == hipSuccess) herr = hipEventRecord(ev_d[next_dl], ks);
                t_dl[next_dl] = now() - T0;
                next_dl++;
            }
            if (next_dl > j || herr != hipSuccess) break;
            if (fed && seen <= sl[next_dl].unit) { herr = hipErrorUnknown; break; }     // the feeder gave up (up_err says why)
        }
        if (herr != hipSuccess) break;
        t_f0[j] = now() - T0;
        herr = hipEventSynchronize(ev_d[j]);
        if (herr != hipSuccess) break;
        t_f1[j] = now() - T0;
        const Slice &s = sl[j];
        const uint8_t *slot = (const uint8_t *)c->h_slot[j % kDownSlots];
        memcpy(res + s.a, c->h_res[j % kDownSlots], (s.b - s.a) * sizeof(xfg_result));
        // only what each chunk produced; chunks whose outputs follow one another without a gap travel in one copy
        for (size_t i = s.a; i < s.b;) {
            const size_t lo = (size_t)chunks[i].raw_off;
            size_t hi = lo + (size_t)res[i].out_produced;
            size_t k = i + 1;
            while (k < s.b && (size_t)chunks[k].raw_off == hi && (size_t)res[k - 1].out_produced == (size_t)chunks[k - 1].raw_size) {
                hi += (size_t)res[k].out_produced;
                k++;
            }
            if (hi > lo) parallel_copy(c->pool, raw + lo, slot + (lo - s.raw_lo), hi - lo);
            i = k;
        }
        t_f2[j] = now() - T0;
    }
    { std::unique_lock<std::mutex> l(m); stop = true; cv.notify_all(); cv.wait(l, [&] { return up_done; }); }
    for (int k = 0; k < lanes; k++) (void)hipStreamSynchronize(c->stream_k[k]);
    (void)hipStreamSynchronize(c->stream_up);
    (void)hipStreamSynchronize(c->stream_dl);
    if (ptrace)
        for (size_t u = 0; u < nu; u++) {
            float a = 0, b = 0, d = 0;
            (void)hipEventElapsedTime(&a, tbase, tk0[u]); (void)hipEventElapsedTime(&b, tbase, tk1[u]); (void)hipEventElapsedTime(&d, tbase, tk2[u]);
            fprintf(stderr, "[pipe] unit %zu lane %zu slices %zu-%zu: gpu time lane free %.2f uploads done %.2f kernel done %.2f ms\n", u, u % lanes, units[u].s0, units[u].s1, a, b, d);
        }
    if (ptrace)
        for (size_t i = 0; i < ns; i++)
            fprintf(stderr, "[pipe] slice %zu (unit %zu): upload %.2f-%.2f  d2h issued %.2f  finish start %.2f event %.2f copied %.2f ms\n",
                    i, sl[i].unit, t_up0[i], t_up1[i], t_dl[i], t_f0[i], t_f1[i], t_f2[i]);
    if (up_err != hipSuccess) return hip_fail(up_err, "pipelined decode (feeder)");
    if (herr != hipSuccess) return hip_fail(herr, "pipelined decode");
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
    // large batches in stream order take the pipelined path (uploads, kernels, downloads and host copies overlap)
    static const bool no_pipeline = [] { const char *e = getenv("XFG_NO_PIPELINE"); return e && *e && *e != '0'; }();   // (tests compare the two paths)
    if (!no_pipeline && raw_len >= kPipelineMin && n <= (1u << 30) && stream_ordered(chunks, n)) {
        // (every allocation of decode_pipelined that can throw -- its vectors, the event bag -- comes before its feeder
        // task is queued: an exception means that nothing has been started, and the plain path takes the call)
        try {
            rc = decode_pipelined(c, comp, comp_len, chunks, n, raw, raw_len, res);
        } catch (...) {
            (void)hipStreamSynchronize(c->stream);
            rc = -1;
        }
        if (rc >= 0) return rc;
    }
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
