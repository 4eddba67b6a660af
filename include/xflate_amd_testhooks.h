/*
 * xflate_amd_testhooks.h -- test-only entry points of libxflate_amd.so.
 *
 * The product's host-side C++ (compress_amd/csrc/host_reader.cpp: meta::Reader, Index,
 * decode_index) is otherwise reachable only through whole XFLATE streams.  These hooks hand the
 * reference's own unit-test vectors to exactly those pieces, so that the C++ -- not just the Python
 * oracle -- is pinned by them:
 *   xflate/internal/meta/reader_test.go:38-452   TestReader (one meta block per vector)
 *   xflate/internal/meta/reader_test.go:455-610  TestReaderReset (back-to-back meta streams)
 *   xflate/internal/meta/meta_test.go:210-246    TestReverseSearch
 *   xflate/index_test.go:14-184                  TestIndexRoundTrip, TestIndex, TestIndexSearch
 * Host only; nothing here touches a GPU.  Not part of the drop-in boundary: a binding of the
 * reference has no use for them.
 */
#ifndef XFLATE_AMD_TESTHOOKS_H
#define XFLATE_AMD_TESTHOOKS_H

#include "xflate_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* meta.(*Reader).decodeBlock once (meta/reader.go:118-257) on in[0, len): returns the XFG_ERR_* class
 * (XFG_ERR_EOF for an empty input); out (>= 32 bytes) / *out_len = mr.buf, *final = mr.final,
 * *in_off = mr.InputOffset. */
int xfg_test_meta_block(const uint8_t *in, size_t len, uint8_t *out, size_t *out_len, int *final, int64_t *in_off);
/* io.Copy(out, meta.NewReader(in)) then Close (meta/reader.go:66-113): returns the XFG_ERR_* class of Copy
 * (XFG_ERR_NONE at io.EOF); the counters are InputOffset, OutputOffset, NumBlocks, FinalMode. */
int xfg_test_meta_stream(const uint8_t *in, size_t len, uint8_t *out, size_t out_cap, size_t *out_len,
                         int64_t *in_off, int64_t *out_off, int64_t *num_blocks, int *final_mode);
/* meta.ReverseSearch (meta/meta.go:35-44): index of the last magic in data, or -1. */
int xfg_test_meta_reverse_search(const uint8_t *data, size_t len);

/* xflate's index type (xflate/index.go:14-139) as an opaque handle. */
void *xfg_test_index_new(void);
void xfg_test_index_free(void *idx);
int xfg_test_index_append_record(void *idx, int64_t comp_size, int64_t raw_size, int type);   /* 1 = ok */
int xfg_test_index_append_index(void *idx, const void *other);                                /* 1 = ok */
/* index{Records: []record{...}} as a Go literal writes it: one record verbatim, checked by nobody. */
void xfg_test_index_push_raw(void *idx, int64_t comp_offset, int64_t raw_offset, int type);
int64_t xfg_test_index_len(const void *idx);
void xfg_test_index_record(const void *idx, int64_t i, xfg_record *out);
void xfg_test_index_last_record(const void *idx, xfg_record *out);
int64_t xfg_test_index_search(const void *idx, int64_t offset);
void xfg_test_index_get_records(const void *idx, int64_t i, xfg_record *prev, xfg_record *curr);
/* (*Reader).decodeIndex (xflate/reader.go:354-424) on an encoded index enc[0, len) with IndexSize = len:
 * returns the XFG_ERR_* class; fills idx (a fresh handle) and *back_size. */
int xfg_test_decode_index(const uint8_t *enc, size_t len, void *idx, int64_t *back_size);

#ifdef __cplusplus
}
#endif
#endif /* XFLATE_AMD_TESTHOOKS_H */
