// xfg_internal.h -- declarations shared by the translation units of libxflate_amd.so.
#ifndef XFG_INTERNAL_H
#define XFG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" hipError_t xfg_launch_inflate(const void *d_comp, const void *d_chunks, uint32_t n,
                                         void *d_raw, void *d_res, hipStream_t stream);

#endif
