// NOT COMPILED: written without a Go toolchain (see ../../README.md).
package amd

/*
#include <stdint.h>
#include "xflate_amd.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/dsnet/compress/internal/errors"
)

// ChunkResult is the outcome of one record of a batch decode: what flateReader and the post-chunk
// checks of xflate.Reader.Read report for it (xflate/reader.go:183-212).
type ChunkResult struct {
	Status      int   // XFG_ST_*: 0 ok, 1 corrupted, 2 unexpected EOF, 3 size mismatch, 4 no sync marker
	InConsumed  int64 // flateReader.InputOffset
	OutProduced int64 // flateReader.OutputOffset
}

// DecodeAll decodes a whole in-memory XFLATE stream with one batch call, the records of shard
// `shard` of `nShards` only (index-range sharding: xflate/index.go:27-31 gives every chunk's input
// and output range up front; one process per GPU takes the range of its rank, no collective).
// It returns the decoded bytes of the shard, the raw offset they start at, and per-record results.
func DecodeAll(device int, stream []byte, shard, nShards int) (raw []byte, rawStart int64, res []ChunkResult, err error) {
	if len(stream) == 0 || nShards < 1 || shard < 0 || shard >= nShards {
		return nil, 0, nil, errorf(errors.Invalid, "bad shard arguments")
	}
	sp := (*C.uint8_t)(unsafe.Pointer(&stream[0]))
	var n C.size_t
	if cls := C.xfg_index_parse(sp, C.size_t(len(stream)), nil, 0, &n); cls != C.XFG_ERR_NONE {
		return nil, 0, nil, classErr(cls)
	}
	if n == 0 {
		return nil, 0, nil, nil
	}
	chunks := make([]C.xfg_chunk, int(n))
	if cls := C.xfg_index_parse(sp, C.size_t(len(stream)), &chunks[0], n, &n); cls != C.XFG_ERR_NONE {
		return nil, 0, nil, classErr(cls)
	}
	first := make([]C.size_t, nShards+1)
	if rc := C.xfg_shard_plan(&chunks[0], n, C.int(nShards), &first[0]); rc != C.XFG_OK {
		return nil, 0, nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	a, b := int(first[shard]), int(first[shard+1])
	if a == b {
		return nil, 0, nil, nil
	}
	mine := chunks[a:b]
	rawStart = int64(mine[0].raw_off)
	rawLen := uint64(mine[len(mine)-1].raw_off+mine[len(mine)-1].raw_size) - uint64(mine[0].raw_off)
	for i := range mine { // the shard's output starts at 0 in its own buffer
		mine[i].raw_off -= C.uint64_t(rawStart)
	}
	raw = make([]byte, rawLen+1)
	cres := make([]C.xfg_result, len(mine))

	var ctx *C.xfg_ctx
	if rc := C.xfg_create(C.int(device), &ctx); rc != C.XFG_OK {
		return nil, 0, nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	defer C.xfg_destroy(ctx)
	// (the error text is per OS thread: stay on this one until it has been fetched)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := C.xfg_decode(ctx, sp, C.size_t(len(stream)), &mine[0], C.size_t(len(mine)),
		(*C.uint8_t)(unsafe.Pointer(&raw[0])), C.size_t(rawLen), &cres[0])
	if rc != C.XFG_OK {
		return nil, 0, nil, errorf(errors.Internal, C.GoString(C.xfg_last_error()))
	}
	res = make([]ChunkResult, len(cres))
	for i, r := range cres {
		res[i] = ChunkResult{int(r.status), int64(r.in_consumed), int64(r.out_produced)}
	}
	return raw[:rawLen], rawStart, res, nil
}
