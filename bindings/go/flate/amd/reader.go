// Package amd is flate.NewReader on an AMD MI355X through libxflate_amd.so (xfg_flate_decode): one
// plain DEFLATE stream inflated by one call, with the in-repo flate package's semantics.
// The exported API is the flate package's (flate/reader.go:14-97).
//
// NOT COMPILED: written without a Go toolchain (see ../../README.md).
package amd

/*
#cgo LDFLAGS: -lxflate_amd
#include <stdint.h>
#include "xflate_amd.h"
*/
import "C"

import (
	"io"
	"unsafe"

	"github.com/dsnet/compress"
	"github.com/dsnet/compress/internal/errors"
)

// ReaderConfig mirrors flate.ReaderConfig (flate/reader.go:35-37).
type ReaderConfig struct {
	Device int
	_      struct{}
}

// Reader mirrors flate.Reader (flate/reader.go:14-33).
type Reader struct {
	InputOffset  int64 // bytes of the DEFLATE stream consumed
	OutputOffset int64 // bytes handed out by Read

	ctx  *C.xfg_ctx
	src  io.Reader
	in   []byte // everything read from src so far
	out  []byte // decoded bytes of in
	pos  int    // bytes of out handed out
	done bool   // the stream's outcome is known
	eof  bool   // src is exhausted
	err  error  // the stream's outcome (io.EOF = clean end)
	base int64  // src offset of in[0] when src is an io.Seeker
}

func errorf(c int, msg string) error { return errors.Error{Code: c, Pkg: "flate", Msg: msg} }

var (
	errCorrupted = errorf(errors.Corrupted, "")
	errClosed    = errorf(errors.Closed, "")
)

// NewReader mirrors flate.NewReader (flate/reader.go:39-43).
func NewReader(r io.Reader, conf *ReaderConfig) (*Reader, error) {
	dev := 0
	if conf != nil {
		dev = conf.Device
	}
	zr := new(Reader)
	if rc := C.xfg_create(C.int(dev), &zr.ctx); rc != C.XFG_OK {
		return nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	zr.Reset(r)
	return zr, nil
}

// Reset mirrors (*flate.Reader).Reset (flate/reader.go:45-56).
func (zr *Reader) Reset(r io.Reader) error {
	ctx := zr.ctx
	*zr = Reader{ctx: ctx, src: r, in: zr.in[:0], out: zr.out[:0]}
	if s, ok := r.(io.Seeker); ok {
		zr.base, _ = s.Seek(0, io.SeekCurrent)
	}
	return nil
}

// decode inflates zr.in from its start (the device decodes whole streams) and records the outcome.
func (zr *Reader) decode() error {
	var inOff, outOff C.int64_t
	var cls C.int
	capOut := 4*len(zr.in) + 4096
	for {
		if cap(zr.out) < capOut {
			zr.out = make([]byte, capOut)
		}
		zr.out = zr.out[:capOut]
		var ip *C.uint8_t
		if len(zr.in) > 0 {
			ip = (*C.uint8_t)(unsafe.Pointer(&zr.in[0]))
		}
		rc := C.xfg_flate_decode(zr.ctx, ip, C.size_t(len(zr.in)),
			(*C.uint8_t)(unsafe.Pointer(&zr.out[0])), C.size_t(capOut), &inOff, &outOff, &cls)
		if rc == C.XFG_E_NOMEM { // the output did not fit
			capOut *= 4
			continue
		}
		if rc != C.XFG_OK {
			return errorf(errors.Internal, C.GoString(C.xfg_last_error()))
		}
		break
	}
	zr.out = zr.out[:int(outOff)]
	zr.InputOffset = int64(inOff)
	switch cls {
	case C.XFG_ERR_NONE:
		zr.err, zr.done = io.EOF, true
	case C.XFG_ERR_CORRUPTED:
		zr.err, zr.done = errCorrupted, true
	case C.XFG_ERR_UNEXPECTED_EOF:
		// more input may follow: the outcome stands only once src is exhausted
		zr.err, zr.done = io.ErrUnexpectedEOF, zr.eof
	default:
		zr.err, zr.done = errorf(errors.Internal, "device decode failed"), true
	}
	return nil
}

// pull reads more of src.  "The reader must not read more than necessary" (api.go:29-72): a
// compress.BufferedReader is peeked and only discarded up to InputOffset; an io.Seeker is read
// ahead and seeked back to the stream's end once that is known; any other reader is read a byte
// at a time.
func (zr *Reader) pull() error {
	switch src := zr.src.(type) {
	case compress.BufferedReader:
		want := 2*len(zr.in) + 4096
		b, err := src.Peek(want)
		zr.in = append(zr.in[:0], b...)
		if len(b) < want && err != nil {
			zr.eof = true
		}
	case io.Seeker:
		buf := make([]byte, len(zr.in)+65536)
		n, err := zr.src.Read(buf)
		zr.in = append(zr.in, buf[:n]...)
		if n == 0 && err != nil {
			zr.eof = true
		}
	default:
		var one [1]byte
		n, err := zr.src.Read(one[:])
		zr.in = append(zr.in, one[:n]...)
		if n == 0 && err != nil {
			zr.eof = true
		}
	}
	return nil
}

// settle gives back what the stream did not consume.
func (zr *Reader) settle() {
	switch src := zr.src.(type) {
	case compress.BufferedReader:
		src.Discard(int(zr.InputOffset))
	case io.Seeker:
		src.Seek(zr.base+zr.InputOffset, io.SeekStart)
	}
}

// Read mirrors (*flate.Reader).Read (flate/reader.go:58-88): decoded bytes first, then the outcome.
func (zr *Reader) Read(buf []byte) (int, error) {
	if zr.err == errClosed {
		return 0, errClosed
	}
	for {
		if zr.pos < len(zr.out) { // deliver what is decoded before pulling more input
			n := copy(buf, zr.out[zr.pos:])
			zr.pos += n
			zr.OutputOffset += int64(n)
			return n, nil
		}
		if zr.done {
			return 0, zr.err
		}
		if err := zr.pull(); err != nil {
			return 0, err
		}
		if err := zr.decode(); err != nil {
			zr.err, zr.done = err, true
			return 0, err
		}
		if zr.done {
			zr.settle()
		}
		if len(buf) == 0 {
			return 0, nil
		}
	}
}

// Close mirrors (*flate.Reader).Close (flate/reader.go:90-97).
func (zr *Reader) Close() error {
	if zr.err == io.EOF || zr.err == errClosed {
		zr.err = errClosed
		if zr.ctx != nil {
			C.xfg_destroy(zr.ctx)
			zr.ctx = nil
		}
		return nil
	}
	return zr.err // the sticky error, or nil while the stream is still being read
}
