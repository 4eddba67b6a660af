// xfg_internal.h -- declarations shared by the translation units of libxflate_amd.so.
#ifndef XFG_INTERNAL_H
#define XFG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" hipError_t xfg_launch_inflate(const void *d_comp, size_t comp_len, const void *d_chunks, uint32_t n,
                                         void *d_raw, size_t raw_len, void *d_res, void *d_scratch,
                                         int num_cu, hipStream_t stream);
extern "C" size_t xfg_scratch_bytes(int num_cu);
// text for xfg_last_error() on the calling thread (xfg_api.cpp)
extern "C" void xfg_set_last_error(const char *text);
struct xfg_ctx;
// memcpy of a large block on the context's helper threads (xfg_api.cpp)
extern "C" void xfg_host_copy(xfg_ctx *c, void *dst, const void *src, size_t n);
extern "C" hipError_t xfg_launch_crc32(const void *d_raw, size_t raw_len, const void *d_chunks, const void *d_res, uint32_t n,
                                       void *d_crc, int num_cu, hipStream_t stream);
extern "C" hipError_t xfg_launch_deflate(const void *d_raw, size_t raw_len, const void *d_chunks, uint32_t n,
                                         void *d_comp, size_t comp_len, void *d_res, int num_cu, hipStream_t stream);

#endif
