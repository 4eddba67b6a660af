/*
 * xflate_amd.h -- C ABI of libxflate_amd.so: MI355X-native XFLATE/DEFLATE decode.
 *
 * This is the drop-in boundary for the read path of dsnet/compress
 * (xflate.NewReader / Reader.Read / Seek / Close, flate.NewReader).  The
 * reference has no FFI on this path (it is pure Go); the shape follows the
 * only native precedent in the reference, the create / stream / destroy cgo
 * shim with caller-owned buffers in internal/cgo/flate/flate.go:18-44,98-133.
 * Every entry point cites the reference interface it replaces.
 *
 * Conventions
 *  - Plain pointers and sizes only; no C++ or torch types.
 *  - int return of the xfg_* transport calls: 0 = XFG_OK, otherwise an XFG_E_*
 *    code (HIP failure, bad argument).  DEFLATE/XFLATE *format* errors are
 *    never transport errors: they are reported per chunk in xfg_result.status
 *    and, in the Reader API, as the reference's error classes (XFG_ERR_*).
 *  - No pointer passed in is retained after the call returns (cgo pointer
 *    rules), except the callbacks / user pointer handed to xfg_reader_new,
 *    which live until xfg_reader_free.
 *  - One xfg_ctx per (thread, device).  Calls on different contexts are
 *    thread-safe; calls on the same context must be serialised by the caller
 *    (the reference Reader is single-goroutine, xflate/reader.go:75-95).
 */
#ifndef XFLATE_AMD_H
#define XFLATE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- transport status -------------------------------------------------- */
#define XFG_OK            0
#define XFG_E_ARG         1   /* bad argument (null pointer, range, alignment, limit) */
#define XFG_E_HIP         2   /* HIP runtime call failed; see xfg_last_error()        */
#define XFG_E_NODEVICE    3   /* no usable MI355X-class device                        */
#define XFG_E_NOMEM       4
#define XFG_E_LIMIT       5   /* a chunk exceeds XFG_MAX_CHUNK_COMP / XFG_MAX_CHUNK_RAW */

/* ---- per-chunk decode status (xfg_result.status) -----------------------
 * 0..2 are the reference's three outcomes of inflating one chunk
 * (flate/reader.go:58-88): io.EOF after a final block, errors.Corrupted,
 * io.ErrUnexpectedEOF.  3..4 are xflate.Reader's own post-chunk checks
 * (xflate/reader.go:190-202).  6 only occurs without XFG_CHUNK_VERIFY. */
#define XFG_ST_OK             0
#define XFG_ST_CORRUPTED      1
#define XFG_ST_UNEXPECTED_EOF 2
#define XFG_ST_SIZE_MISMATCH  3   /* consumed != csize(+5) or produced != rsize (reader.go:199)  */
#define XFG_ST_NO_SYNC        4   /* chunk does not end in 00 00 ff ff        (reader.go:190)  */
#define XFG_ST_INTERNAL       5
#define XFG_ST_OUT_FULL       6   /* raw_size capacity reached on a chunk without VERIFY        */

/* ---- chunk descriptor -------------------------------------------------
 * One independently decodable unit: an XFLATE chunk as located by the index
 * (xflate/index.go:27-31; csize/rsize of xflate/reader.go:265-269), or a whole
 * plain DEFLATE stream for flate.NewReader. */
#define XFG_CHUNK_END_BLOCK  0x01u  /* decode chunk || 01 00 00 ff ff  (chunkReader, xflate/reader.go:32-70) */
#define XFG_CHUNK_SYNC       0x02u  /* deflateType: require trailing 00 00 ff ff (reader.go:190)             */
#define XFG_CHUNK_VERIFY     0x04u  /* require consumed == comp_size(+5) and produced == raw_size (reader.go:199) */
#define XFG_CHUNK_FOOTER     0x08u  /* footerType: the stream ends inside the chunk, no +5 (reader.go:196-198)    */
#define XFG_CHUNK_RFC_REP16  0x10u  /* code-length symbol 16 repeats the previous array entry (RFC 1951 / Go
                                       stdlib, the inflater xflate.Reader calls: xflate/flate.go:43); unset =
                                       the in-repo flate package's rule (flate/prefix.go:127-142)            */
#define XFG_CHUNK_HOST_META  0x20u  /* an index or footer record whose meta blocks the HOST has already decoded, every one
                                       valid and the record consumed to its last byte (decodeIndex / decodeFooter,
                                       xflate/reader.go:354-483; what decodes as meta blocks inflates to nothing,
                                       meta/meta_test.go:28-51).  The device does not inflate it a second time: it reports
                                       status OK, out_produced 0 and the in_consumed that inflating would give (comp_size,
                                       +5 with END_BLOCK unless FOOTER).  xfg_index_parse and the Reader set it; without it
                                       the record is inflated like any chunk (one dynamic-block header per 12-64 bytes:
                                       a 40 KB index record costs one wave about 80 ms). */
/* what xflate.Reader implies for each record type */
#define XFG_CHUNK_XFLATE_DEFLATE (XFG_CHUNK_END_BLOCK | XFG_CHUNK_SYNC | XFG_CHUNK_VERIFY | XFG_CHUNK_RFC_REP16)
#define XFG_CHUNK_XFLATE_INDEX   (XFG_CHUNK_END_BLOCK | XFG_CHUNK_VERIFY | XFG_CHUNK_RFC_REP16)
#define XFG_CHUNK_XFLATE_FOOTER  (XFG_CHUNK_END_BLOCK | XFG_CHUNK_VERIFY | XFG_CHUNK_FOOTER | XFG_CHUNK_RFC_REP16)

/* Per-chunk size limits of the device path (32-bit bit positions inside a chunk).  The reference has
 * none: a chunk (or, for xfg_flate_decode, a whole plain DEFLATE stream) beyond them is reported as
 * XFG_E_LIMIT by the batch calls and as XFG_ERR_LIMIT by the Reader and by xfg_index_parse;
 * no fixture of the reference reaches these sizes, so behaviour there is unpinned. */
#define XFG_MAX_CHUNK_COMP ((1u << 28) - 64)   /* bytes of compressed data per chunk   */
#define XFG_MAX_CHUNK_RAW  ((1u << 31) - 64)   /* bytes of decompressed data per chunk */

typedef struct {
    uint64_t comp_off;    /* byte offset of the chunk in the compressed buffer        */
    uint64_t comp_size;   /* csize                                                    */
    uint64_t raw_off;     /* byte offset of the chunk's output in the raw buffer      */
    uint64_t raw_size;    /* rsize; also the output capacity of this chunk            */
    uint32_t flags;       /* XFG_CHUNK_*                                              */
    uint32_t reserved;    /* must be 0                                                */
} xfg_chunk;

typedef struct {
    int32_t  status;        /* XFG_ST_*                                                            */
    uint32_t reserved;
    int64_t  in_consumed;   /* flateReader.InputOffset at the end of the chunk (xflate/flate.go:31) */
    int64_t  out_produced;  /* flateReader.OutputOffset; bytes [raw_off, raw_off+out_produced) are valid */
} xfg_result;

typedef struct xfg_ctx xfg_ctx;

/* ---- device context ---------------------------------------------------- */
int  xfg_device_count(void);
/* One context = one HIP device + one stream + reusable staging buffers. */
int  xfg_create(int device, xfg_ctx **out);
void xfg_destroy(xfg_ctx *ctx);
const char *xfg_strerror(int code);
/* Text of the last HIP failure on this thread ("" if none). */
const char *xfg_last_error(void);

/*
 * Decode n chunks, host pointers, blocking.  Replaces the per-chunk loop of
 * xflate.Reader.Read (xflate/reader.go:183-212: flateReader.Read over
 * chunkReader, then the sync/size checks) for a whole batch: copies comp to
 * the device, launches the inflate kernels, copies raw and res back.
 *   comp/comp_len   compressed bytes that every chunk's [comp_off, comp_off+comp_size) lies in
 *   raw/raw_len     output; every chunk's [raw_off, raw_off+raw_size) must lie in it and not overlap another's
 *   res[n]          per-chunk outcome
 */
int xfg_decode(xfg_ctx *ctx, const uint8_t *comp, size_t comp_len,
               const xfg_chunk *chunks, size_t n,
               uint8_t *raw, size_t raw_len, xfg_result *res);

/*
 * Same, device-resident and asynchronous: all four pointers are device
 * pointers (d_comp at least 16-byte aligned), the work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the context's stream) and the call
 * returns without synchronising.  Allocates nothing per call.  Launches of one context
 * share the context's chunk counters: enqueue them on one stream (or wait in between), never
 * on two streams that may run them at the same time.  n <= 2^30.  This is what bench.py times.
 *
 * Trust: the chunk table lives on the device and the host never sees it, so it is NOT trusted.  The
 * kernel checks every descriptor against comp_len / raw_len and the per-chunk limits before it touches
 * the chunk: comp_size > XFG_MAX_CHUNK_COMP, raw_size > XFG_MAX_CHUNK_RAW, a range that leaves
 * [0, comp_len) / [0, raw_len), or reserved != 0 gives that chunk status XFG_ST_INTERNAL with
 * in_consumed = out_produced = 0; nothing is read or written for it (what xfg_decode reports as
 * XFG_E_LIMIT / XFG_E_ARG for the whole call).  Overlapping output ranges of two chunks are the
 * caller's business, as in xfg_decode.  xfg_deflate_device and xfg_crc32_device check the same way
 * (a refused chunk compresses / hashes nothing).  The call leaves the calling thread's current HIP
 * device as it found it.
 */
int xfg_decode_device(xfg_ctx *ctx, const void *d_comp, size_t comp_len,
                      const void *d_chunks, size_t n,
                      void *d_raw, size_t raw_len, void *d_res, void *hip_stream);
/*
 * xfg_decode_device for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  No counterpart in the reference (which allocates freely, being Go); it
 * exists for those callers only.  It takes no context: the one scratch buffer of the decode path, the chunk
 * counters of the persistent grid (2048 bytes), is a __device__ array of the library's code object that every
 * GPU the library runs on holds until the process ends.  The call makes no hipMalloc / hipFree and calls no
 * allocating routine; apart from the first call on a device (which looks up the array's device address and the
 * CU count: call once with n = 0 outside any capture to do just that) it enqueues one 2 KB memset and the kernel
 * on hip_stream (NULL = the null stream) and returns without synchronising.  Arguments, per-chunk results and the
 * trust rules are xfg_decode_device's.  Limit: n <= 2^30 chunks per call, more is refused with XFG_E_LIMIT.
 * All calls on one GPU share that GPU's counters: two calls that may run at the same time on one GPU race --
 * keep them on one stream, or order them.  The calling thread's current device is left as it was.
 */
int xfg_decode_device_noalloc(int device, const void *d_comp, size_t comp_len,
                              const void *d_chunks, size_t n,
                              void *d_raw, size_t raw_len, void *d_res, void *hip_stream);
/*
 * How many wavefronts decode one chunk.  No counterpart in the reference (whose Reader decodes one chunk at a
 * time on one goroutine, xflate/reader.go:162-215); a tuning and test knob.  0 (the default, or what the
 * environment variable XFG_DECODE_WAVES says): chosen per launch from the chunk count -- a table with more than
 * 1.25 chunks per wave slot of the GPU (4096 slots on MI355X) gives every chunk one wavefront, a smaller one
 * gives every chunk a workgroup of 2, 3, 4 or 12 wavefronts that decode consecutive batches of the chunk's blocks
 * together (a Reader's window, flate.NewReader's single stream, a few huge chunks).  1: always one wavefront.
 * 2..16: always a workgroup of that many.  Results are bit-identical whatever the setting.  Process-wide; returns
 * XFG_E_ARG for any other value.  xfg_decode_waves_for() says what a launch of n chunks would use.
 */
int xfg_set_decode_waves(int waves);
int xfg_decode_waves_for(uint32_t n, int num_cu);
/*
 * CRC-32 (IEEE 802.3: Go's hash/crc32.IEEE, the checksum of gzip trailers and of the XFLATE
 * index, xflate/reader.go:386) of every chunk's decoded bytes, computed where they are: the
 * reference's gzip embedding hashes on the CPU while writing (xflate/example_test.go:171-182),
 * a GPU reader checks the trailer with this.  Device pointers, asynchronous like
 * xfg_decode_device.  d_crc[i] = crc32(raw[raw_off_i .. raw_off_i + out_produced_i)); d_res may
 * be NULL (then every chunk counts raw_size bytes).  raw_len = bytes at d_raw: a chunk whose range
 * leaves it hashes nothing (d_crc[i] = 0).
 */
int xfg_crc32_device(xfg_ctx *ctx, const void *d_raw, size_t raw_len, const void *d_chunks, size_t n,
                     const void *d_res, void *d_crc /* uint32_t[n] */, void *hip_stream);
/* crc32(A || B) from crc32(A), crc32(B), len(B): folds the per-chunk values into the stream's. Host only. */
uint32_t xfg_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);
/* Wait for everything enqueued on the context's stream. */
int xfg_sync(xfg_ctx *ctx);

/*
 * Compress n chunks (the write side of the same chunk format).  Replaces
 * what xflate.Writer does to one chunk: flate.Writer.Write of the chunk's bytes and the sync flush
 * that ends it (xflate/writer.go:138-165 and 169-204 through xflate/flate.go:97-118), i.e. a DEFLATE
 * stream without a final block that ends in 00 00 ff ff.  The coding is this library's own (greedy
 * LZ77, fixed Huffman codes, stored blocks when that is smaller): as with any DEFLATE compressor only
 * the decoded bytes are pinned, not the compressed ones.  Index and footer stay on the host.
 *   chunk i reads  raw[raw_off, raw_off + raw_size)
 *   and writes     comp[comp_off, comp_off + out_produced), out_produced <= comp_size;
 *   comp_off must be a multiple of 4 and comp_size >= xfg_deflate_bound(raw_size), else the chunk's
 *   status is XFG_ST_OUT_FULL and nothing is written for it.  reserved must be 0.  flags: XFG_DEFLATE_DYNAMIC
 *   or 0; the XFG_CHUNK_* bits of the decode side are ignored.
 *   res[i].in_consumed = raw_size, res[i].out_produced = compressed bytes.
 * xfg_deflate takes host pointers and blocks; xfg_deflate_device takes device pointers (d_comp at
 * least 4-byte aligned), enqueues on hip_stream (NULL = the context's stream) and allocates nothing.
 */
#define XFG_DEFLATE_DYNAMIC 0x100u   /* two passes over the chunk: count the symbols, build length-limited Huffman codes
                                        for them (RFC 1951 3.2.7) and use those when they beat the fixed ones, header
                                        included.  About half the speed, markedly smaller output on data whose bytes
                                        are not evenly distributed. */
size_t xfg_deflate_bound(size_t raw_size);
int xfg_deflate(xfg_ctx *ctx, const uint8_t *raw, size_t raw_len,
                const xfg_chunk *chunks, size_t n,
                uint8_t *comp, size_t comp_len, xfg_result *res);
int xfg_deflate_device(xfg_ctx *ctx, const void *d_raw, size_t raw_len,
                       const void *d_chunks, size_t n,
                       void *d_comp, size_t comp_len, void *d_res, void *hip_stream);

/* ======================================================================== */
/* Host-side mirror of the reference reader                                  */
/* ======================================================================== */

/* Error classes of the reference (internal/errors/errors.go:36-56), plus the
 * three io sentinels its tests compare by identity. */
#define XFG_ERR_NONE            0
#define XFG_ERR_INTERNAL        1    /* errors.Internal  */
#define XFG_ERR_INVALID         2    /* errors.Invalid   */
#define XFG_ERR_DEPRECATED      3    /* errors.Deprecated*/
#define XFG_ERR_CORRUPTED       4    /* errors.Corrupted */
#define XFG_ERR_CLOSED          5    /* errors.Closed    */
#define XFG_ERR_EOF             16   /* io.EOF           */
#define XFG_ERR_UNEXPECTED_EOF  17   /* io.ErrUnexpectedEOF */
#define XFG_ERR_IO              18   /* the caller's read/seek callback failed */
#define XFG_ERR_DEVICE          19   /* transport failure (see xfg_last_error) */
#define XFG_ERR_LIMIT           20   /* a chunk exceeds XFG_MAX_CHUNK_COMP / XFG_MAX_CHUNK_RAW: a limit of this
                                        library's device path that the reference does not have (it reads such
                                        a stream); not a property of the input */

/* io.ReadSeeker supplied by the caller (xflate.NewReader's rs argument).
 * read: up to n bytes into buf, returns bytes read, 0 at EOF, <0 on error.
 * seek: whence 0/1/2 = io.SeekStart/Current/End, returns new offset or <0. */
typedef int64_t (*xfg_read_fn)(void *user, uint8_t *buf, int64_t n);
typedef int64_t (*xfg_seek_fn)(void *user, int64_t offset, int whence);

typedef struct xfg_reader xfg_reader;

/* record types of xflate/index.go:7-12 */
#define XFG_REC_UNKNOWN 0
#define XFG_REC_DEFLATE 1
#define XFG_REC_INDEX   2
#define XFG_REC_FOOTER  3
typedef struct {
    int64_t comp_offset;   /* ENDING offsets, as in xflate/index.go:27-31 */
    int64_t raw_offset;
    int32_t type;
    int32_t reserved;
} xfg_record;

/*
 * xflate.NewReader(rs, nil)  (xflate/reader.go:113-117).  Seeks rs to its end,
 * decodes footer and indexes on the host (decodeFooter :430, decodeIndexes
 * :296), leaves the reader at raw offset 0.  Like the reference it returns a
 * non-NULL reader together with a non-zero error class when the index is bad;
 * that reader's Read/Seek/Close then report the sticky error.
 * Returns an XFG_ERR_* class.
 */
int xfg_reader_new(xfg_ctx *ctx, xfg_read_fn rd, xfg_seek_fn sk, void *user, xfg_reader **out);
/* xflate.NewReader(bytes.NewReader(stream), nil): the same over a stream in memory, which the caller keeps alive and
 * unchanged until xfg_reader_free (a mapped file, say).  No callbacks: the library reads the bytes itself. */
int xfg_reader_new_mem(xfg_ctx *ctx, const uint8_t *stream, size_t len, xfg_reader **out);
/* Reset(bytes.NewReader(stream)) on any reader (it keeps its decode window's buffers). */
int xfg_reader_reset_mem(xfg_reader *r, const uint8_t *stream, size_t len);
/* xflate.(*Reader).Reset (reader.go:124). */
int xfg_reader_reset(xfg_reader *r, xfg_read_fn rd, xfg_seek_fn sk, void *user);
/* xflate.(*Reader).Read (reader.go:162): returns the byte count (> 0) or 0 with *err set.
 * How many bytes one call returns is the one thing that differs from the reference, which never returns more than
 * the rest of the current chunk: a request of 1 MiB or more returns a run of whole chunks that decoded cleanly (one
 * copy out of the decoded window), and a request of 32 MiB or more that needs a new window has the chunks it
 * covers decoded straight into buf, which is then used as scratch beyond the count returned (as io.Reader allows).
 * Bytes, offsets, and the error class after a given number of bytes are the reference's either way. */
int64_t xfg_reader_read(xfg_reader *r, uint8_t *buf, int64_t n, int *err);
/* xflate.(*Reader).Seek (reader.go:221): returns the new offset, or 0 with *err set. */
int64_t xfg_reader_seek(xfg_reader *r, int64_t offset, int whence, int *err);
/* xflate.(*Reader).Close (reader.go:282): idempotent; returns the sticky error class. */
int xfg_reader_close(xfg_reader *r);
void xfg_reader_free(xfg_reader *r);
/* The merged index (what reader.go:76 says it wants to export): n records, ending offsets. */
int64_t xfg_reader_num_records(const xfg_reader *r);
int xfg_reader_get_records(const xfg_reader *r, xfg_record *out, int64_t cap);
/* Bytes pulled through the read callback so far (countReadSeeker, xflate/reader_test.go:19-28). */
int64_t xfg_reader_bytes_read(const xfg_reader *r);
/* How many chunks one device launch may cover (read-ahead window), default 256 MiB raw. */
int xfg_reader_set_window(xfg_reader *r, int64_t max_raw_bytes);

/*
 * Parse footer + indexes of a whole in-memory XFLATE stream and emit the
 * device chunk table for every record (deflate, index and footer records, in
 * stream order), ready for xfg_decode / xfg_decode_device.  Host only.
 * Returns an XFG_ERR_* class; *n_out = number of descriptors written (<= cap),
 * or the number required when chunks == NULL.
 */
int xfg_index_parse(const uint8_t *stream, size_t len, xfg_chunk *chunks, size_t cap, size_t *n_out);

/*
 * Index-range sharding over the GPUs of a node.  The reference reads chunks one
 * after the other (xflate/reader.go:183-212) although they are independent
 * (xflate/index.go:27-31 gives every chunk's input and output range up front);
 * this cuts the chunk table into n_shards contiguous record ranges of about
 * equal comp_size + raw_size.  first[] has n_shards + 1 entries; shard k owns
 * records [first[k], first[k+1]).  Host only, deterministic: every process of
 * a one-process-per-GPU job computes the same plan and takes the range of its
 * rank; no data-path collective is involved.  Returns an XFG_E_* code.
 */
int xfg_shard_plan(const xfg_chunk *chunks, size_t n, int n_shards, size_t *first);

/*
 * flate.NewReader(r, nil) + ReadAll + Close in one call (flate/reader.go:39-97)
 * for an in-memory stream: inflates one plain DEFLATE stream on the device
 * with the in-repo flate package's semantics.  *err = XFG_ERR_NONE (clean
 * io.EOF), XFG_ERR_CORRUPTED or XFG_ERR_UNEXPECTED_EOF; *in_off / *out_off are
 * Reader.InputOffset / OutputOffset (flate/reader.go:15-16).  out_cap too
 * small returns XFG_E_NOMEM with *out_off = bytes that fit.
 */
int xfg_flate_decode(xfg_ctx *ctx, const uint8_t *in, size_t in_len,
                     uint8_t *out, size_t out_cap,
                     int64_t *in_off, int64_t *out_off, int *err);

#ifdef __cplusplus
}
#endif
#endif /* XFLATE_AMD_H */
