// NOT COMPILED: written without a Go toolchain (see ../../README.md).
package amd

/*
#include <stdint.h>
#include "xflate_amd.h"
*/
import "C"

import (
	"encoding/binary"
	"hash/crc32"
	"io"
	"runtime"
	"unsafe"

	"github.com/dsnet/compress/internal/errors"
	"github.com/dsnet/compress/xflate/internal/meta"
)

// FlushMode and its values mirror xflate/common.go:76-112.
type FlushMode int

const (
	FlushSync FlushMode = iota
	FlushFull
	FlushIndex
)

const (
	DefaultChunkSize = 1 << 18 // xflate/writer.go:49
	DefaultIndexSize = 1 << 12 // xflate/writer.go:65
)

var magic = [3]byte{'X', 'F', 0x00} // xflate/common.go

// WriterConfig mirrors xflate.WriterConfig (writer.go:40-71).  Level is ignored: the device
// compressor has one (greedy LZ77, fixed Huffman codes or stored blocks).
type WriterConfig struct {
	Level     int
	ChunkSize int64
	IndexSize int64
	Device    int
	// BatchChunks closed chunks are compressed in one launch (default 1024): one wave per chunk, so
	// a launch per chunk would leave the GPU empty.
	BatchChunks int
	_           struct{}
}

type record struct{ comp, raw int64 } // ending offsets, xflate/index.go:27-31

type queued struct {
	raw  []byte // bytes of the chunk not yet compressed
	zout int64  // compressed bytes of the chunk already written (sync flushes inside it)
	zin  int64  // the chunk's raw size
}

// Writer mirrors xflate.Writer (writer.go:19-36).  Closed chunks wait in a queue and are compressed a
// batch at a time, so OutputOffset and the index lag behind InputOffset until the queue is drained
// (by FlushSync, FlushIndex, Close or a full queue).
type Writer struct {
	InputOffset  int64
	OutputOffset int64

	wr   io.Writer
	mw   meta.Writer
	ctx  *C.xfg_ctx
	nchk int64
	nidx int64
	nbat int

	pend      []byte // current chunk, not yet compressed
	zin, zout int64  // current chunk: raw bytes taken, compressed bytes written
	queue     []queued
	recs      []record
	backSize  int64
	err       error
	scratch   [64]byte
}

// NewWriter mirrors xflate.NewWriter (writer.go:78-104).
func NewWriter(wr io.Writer, conf *WriterConfig) (*Writer, error) {
	var nchk, nidx int64
	dev, nbat := 0, 1024
	if conf != nil {
		switch {
		case conf.ChunkSize < 0:
			return nil, errorf(errors.Invalid, "invalid chunk size")
		case conf.ChunkSize > 0:
			nchk = conf.ChunkSize
		}
		switch {
		case conf.IndexSize < 0:
			nidx = -1
		case conf.IndexSize > 0:
			nidx = conf.IndexSize
		}
		dev = conf.Device
		if conf.BatchChunks > 0 {
			nbat = conf.BatchChunks
		}
	}
	if nchk == 0 {
		nchk = DefaultChunkSize
	}
	if nidx == 0 {
		nidx = DefaultIndexSize
	}
	xw := &Writer{nchk: nchk, nidx: nidx, nbat: nbat}
	if rc := C.xfg_create(C.int(dev), &xw.ctx); rc != C.XFG_OK {
		return nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	xw.Reset(wr)
	return xw, nil
}

// Reset mirrors (*xflate.Writer).Reset (writer.go:107-131).
func (xw *Writer) Reset(wr io.Writer) error {
	xw.wr, xw.err = wr, nil
	xw.InputOffset, xw.OutputOffset = 0, 0
	xw.pend, xw.zin, xw.zout = xw.pend[:0], 0, 0
	xw.queue, xw.recs, xw.backSize = nil, nil, 0
	return nil
}

func (xw *Writer) emit(b []byte) error {
	n, err := xw.wr.Write(b)
	xw.OutputOffset += int64(n)
	return err
}

var emptyStored = []byte{0x00, 0x00, 0x00, 0xff, 0xff} // what a sync flush with nothing new writes

// deflate compresses the chunks in one launch (xfg_deflate).
func (xw *Writer) deflate(chunks [][]byte) ([][]byte, error) {
	var raw []byte
	desc := make([]C.xfg_chunk, len(chunks))
	var co, ro uint64
	for i, c := range chunks {
		b := uint64(C.xfg_deflate_bound(C.size_t(len(c))))
		b = (b + 3) &^ 3 // comp_off must be a multiple of 4
		desc[i] = C.xfg_chunk{comp_off: C.uint64_t(co), comp_size: C.uint64_t(b), raw_off: C.uint64_t(ro), raw_size: C.uint64_t(len(c))}
		co += b
		ro += uint64(len(c))
		raw = append(raw, c...)
	}
	comp := make([]byte, co+1)
	res := make([]C.xfg_result, len(chunks))
	var rp *C.uint8_t
	if len(raw) > 0 {
		rp = (*C.uint8_t)(unsafe.Pointer(&raw[0]))
	}
	// (the error text is per OS thread: stay on this one until it has been fetched)
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := C.xfg_deflate(xw.ctx, rp, C.size_t(len(raw)), &desc[0], C.size_t(len(desc)),
		(*C.uint8_t)(unsafe.Pointer(&comp[0])), C.size_t(co), &res[0])
	if rc != C.XFG_OK {
		return nil, errorf(errors.Internal, C.GoString(C.xfg_last_error()))
	}
	out := make([][]byte, len(chunks))
	for i := range chunks {
		if res[i].status != C.XFG_ST_OK {
			return nil, errorf(errors.Internal, "device compressor failed on a chunk")
		}
		o := uint64(desc[i].comp_off)
		out[i] = comp[o : o+uint64(res[i].out_produced)]
	}
	return out, nil
}

// drain compresses and writes the queued chunks and appends their records (writer.go:184).
func (xw *Writer) drain() error {
	if len(xw.queue) == 0 {
		return nil
	}
	var todo [][]byte
	for _, q := range xw.queue {
		if len(q.raw) > 0 {
			todo = append(todo, q.raw)
		}
	}
	var comp [][]byte
	if len(todo) > 0 {
		var err error
		if comp, err = xw.deflate(todo); err != nil {
			return err
		}
	}
	k := 0
	for _, q := range xw.queue {
		c := emptyStored
		if len(q.raw) > 0 {
			c = comp[k]
			k++
		}
		if err := xw.emit(c); err != nil {
			return err
		}
		var last record
		if len(xw.recs) > 0 {
			last = xw.recs[len(xw.recs)-1]
		}
		xw.recs = append(xw.recs, record{last.comp + q.zout + int64(len(c)), last.raw + q.zin})
	}
	xw.queue = xw.queue[:0]
	return nil
}

// Write mirrors (*xflate.Writer).Write (writer.go:138-165).
func (xw *Writer) Write(buf []byte) (int, error) {
	if xw.err != nil {
		return 0, xw.err
	}
	var cnt int
	for len(buf) > 0 && xw.err == nil {
		remain := xw.nchk - xw.zin
		if remain <= 0 {
			xw.err = xw.Flush(FlushFull)
			continue
		}
		if remain > int64(len(buf)) {
			remain = int64(len(buf))
		}
		xw.pend = append(xw.pend, buf[:remain]...)
		xw.zin += remain
		buf = buf[remain:]
		cnt += int(remain)
	}
	xw.InputOffset += int64(cnt)
	return cnt, xw.err
}

// Flush mirrors (*xflate.Writer).Flush (writer.go:169-204).
func (xw *Writer) Flush(mode FlushMode) error {
	if xw.err != nil {
		return xw.err
	}
	switch mode {
	case FlushSync:
		if xw.err = xw.drain(); xw.err != nil { // bytes leave in stream order
			return xw.err
		}
		out := emptyStored
		if len(xw.pend) > 0 {
			var comp [][]byte
			if comp, xw.err = xw.deflate([][]byte{xw.pend}); xw.err != nil {
				return xw.err
			}
			out = comp[0]
			xw.pend = xw.pend[:0]
		}
		xw.zout += int64(len(out))
		xw.err = xw.emit(out)
		return xw.err
	case FlushFull:
		xw.queue = append(xw.queue, queued{append([]byte(nil), xw.pend...), xw.zout, xw.zin})
		xw.pend, xw.zin, xw.zout = xw.pend[:0], 0, 0
		if int64(len(xw.recs)+len(xw.queue)) == xw.nidx {
			xw.err = xw.Flush(FlushIndex)
		} else if len(xw.queue) >= xw.nbat {
			xw.err = xw.drain()
		}
		return xw.err
	case FlushIndex:
		if xw.zin+xw.zout > 0 {
			if err := xw.Flush(FlushFull); err != nil {
				return err
			}
		}
		if xw.err = xw.drain(); xw.err != nil {
			return xw.err
		}
		var size int64
		size, xw.err = xw.encodeIndex()
		xw.recs = xw.recs[:0]
		xw.backSize = size
		return xw.err
	default:
		return errorf(errors.Invalid, "invalid flush mode")
	}
}

// Close mirrors (*xflate.Writer).Close (writer.go:209-233).
func (xw *Writer) Close() error {
	if xw.err == errClosed {
		return nil
	}
	if xw.err != nil {
		return xw.err
	}
	if xw.zout+xw.zin > 0 || len(xw.recs) > 0 || len(xw.queue) > 0 {
		if xw.err = xw.Flush(FlushIndex); xw.err != nil {
			return xw.err
		}
	}
	err := xw.encodeFooter(xw.backSize)
	if err != nil {
		xw.err = err
	} else {
		xw.err = errClosed
		C.xfg_destroy(xw.ctx)
		xw.ctx = nil
	}
	return err
}

// encodeIndex writes the index as a meta stream (format of writer.go:238-277): backSize, numRecs,
// totalCompSize, totalRawSize, per-record size pairs, CRC-32 of those varints.
func (xw *Writer) encodeIndex() (int64, error) {
	var crc uint32
	var errVLI error
	writeVLI := func(x int64) {
		b := xw.scratch[:binary.PutUvarint(xw.scratch[:], uint64(x))]
		crc = crc32.Update(crc, crc32.IEEETable, b)
		if _, err := xw.mw.Write(b); err != nil {
			errVLI = err
		}
	}
	xw.mw.Reset(xw.wr)
	defer func() { xw.OutputOffset += xw.mw.OutputOffset }()
	xw.mw.FinalMode = meta.FinalMeta
	var last record
	if len(xw.recs) > 0 {
		last = xw.recs[len(xw.recs)-1]
	}
	writeVLI(xw.backSize)
	writeVLI(int64(len(xw.recs)))
	writeVLI(last.comp)
	writeVLI(last.raw)
	var pre record
	for _, r := range xw.recs {
		writeVLI(r.comp - pre.comp)
		writeVLI(r.raw - pre.raw)
		pre = r
	}
	if errVLI != nil {
		return 0, errVLI
	}
	binary.LittleEndian.PutUint32(xw.scratch[:], crc)
	if _, err := xw.mw.Write(xw.scratch[:4]); err != nil {
		return 0, err
	}
	if err := xw.mw.Close(); err != nil {
		return 0, err
	}
	return xw.mw.OutputOffset, nil
}

// encodeFooter writes the footer (format of writer.go:280-298): magic, backSize, one final meta block.
func (xw *Writer) encodeFooter(backSize int64) error {
	n := copy(xw.scratch[:], magic[:])
	n += binary.PutUvarint(xw.scratch[n:], uint64(backSize))
	xw.mw.Reset(xw.wr)
	defer func() { xw.OutputOffset += xw.mw.OutputOffset }()
	xw.mw.FinalMode = meta.FinalStream
	if _, err := xw.mw.Write(xw.scratch[:n]); err != nil {
		return err
	}
	if err := xw.mw.Close(); err != nil {
		return err
	}
	if xw.mw.NumBlocks != 1 {
		return errorf(errors.Internal, "footer was not a single block")
	}
	return nil
}
