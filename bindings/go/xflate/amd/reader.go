// Package amd is xflate's read and write path on an AMD MI355X through libxflate_amd.so.
// The exported API is the xflate package's (xflate/reader.go:99-291, xflate/writer.go:40-298).
//
// NOT COMPILED: written without a Go toolchain (see ../../README.md).
package amd

/*
#cgo LDFLAGS: -lxflate_amd
#include <stdint.h>
#include <stdlib.h>
#include "xflate_amd.h"

// the callbacks below, as C function pointers
extern int64_t goXfgRead(void *user, uint8_t *buf, int64_t n);
extern int64_t goXfgSeek(void *user, int64_t off, int whence);
*/
import "C"

import (
	"io"
	"runtime"
	"runtime/cgo"
	"unsafe"

	"github.com/dsnet/compress/internal/errors"
)

// ReaderConfig mirrors xflate.ReaderConfig (reader.go:99-101), plus the device to run on.
type ReaderConfig struct {
	Device int // HIP device ordinal
	_      struct{}
}

// Reader mirrors xflate.Reader (reader.go:75-95): an io.ReadSeeker over an XFLATE stream.
type Reader struct {
	ctx  *C.xfg_ctx
	r    *C.xfg_reader
	h    cgo.Handle     // -> *rsBox
	user unsafe.Pointer // C memory holding h: what the callbacks get
}

type rsBox struct{ rs io.ReadSeeker }

func boxOf(user unsafe.Pointer) *rsBox {
	return cgo.Handle(*(*uintptr)(user)).Value().(*rsBox)
}

//export goXfgRead
func goXfgRead(user unsafe.Pointer, buf *C.uint8_t, n C.int64_t) C.int64_t {
	b := boxOf(user)
	dst := unsafe.Slice((*byte)(unsafe.Pointer(buf)), int(n))
	// io.Reader may return 0, nil; the C side takes 0 as EOF, so insist on progress
	for {
		cnt, err := b.rs.Read(dst)
		if cnt > 0 {
			return C.int64_t(cnt)
		}
		if err == io.EOF {
			return 0
		}
		if err != nil {
			return -1
		}
	}
}

//export goXfgSeek
func goXfgSeek(user unsafe.Pointer, off C.int64_t, whence C.int) C.int64_t {
	pos, err := boxOf(user).rs.Seek(int64(off), int(whence))
	if err != nil {
		return -1
	}
	return C.int64_t(pos)
}

func errorf(c int, msg string) error { return errors.Error{Code: c, Pkg: "xflate", Msg: msg} }

// the values the xflate package compares against (xflate/common.go:118-121); errors.Error is comparable
var (
	errCorrupted = errorf(errors.Corrupted, "")
	errClosed    = errorf(errors.Closed, "")
)

// classErr maps an XFG_ERR_* class to the reference's error values
// (internal/errors/errors.go:36-56, xflate/common.go:114-134).
// xfg_last_error() is per OS thread (a thread_local in the library), and the Go scheduler may move a goroutine to
// another thread between two cgo calls: every method that may end in classErr pins its goroutine to its thread for
// the length of the call (pinned), so that the text fetched is the failing call's.
func pinned() func() {
	runtime.LockOSThread()
	return runtime.UnlockOSThread
}

func classErr(c C.int) error {
	switch c {
	case C.XFG_ERR_NONE:
		return nil
	case C.XFG_ERR_EOF:
		return io.EOF
	case C.XFG_ERR_UNEXPECTED_EOF:
		return io.ErrUnexpectedEOF
	case C.XFG_ERR_CLOSED:
		return errClosed
	case C.XFG_ERR_CORRUPTED:
		return errCorrupted
	case C.XFG_ERR_INVALID:
		return errorf(errors.Invalid, "invalid argument")
	case C.XFG_ERR_DEPRECATED:
		return errorf(errors.Deprecated, "")
	case C.XFG_ERR_IO:
		return errorf(errors.Internal, "the ReadSeeker failed")
	case C.XFG_ERR_LIMIT:
		return errorf(errors.Internal, "a chunk exceeds the device path's per-chunk limit")
	default: // XFG_ERR_INTERNAL, XFG_ERR_DEVICE
		return errorf(errors.Internal, C.GoString(C.xfg_last_error()))
	}
}

// NewReader mirrors xflate.NewReader (reader.go:113-117): like the reference it returns the
// Reader together with the error when the index cannot be parsed.
func NewReader(rs io.ReadSeeker, conf *ReaderConfig) (*Reader, error) {
	dev := 0
	if conf != nil {
		dev = conf.Device
	}
	xr := new(Reader)
	if rc := C.xfg_create(C.int(dev), &xr.ctx); rc != C.XFG_OK {
		return nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	xr.user = C.malloc(C.size_t(unsafe.Sizeof(uintptr(0))))
	runtime.SetFinalizer(xr, (*Reader).free)
	err := xr.Reset(rs)
	return xr, err
}

// NewReaderMem is NewReader over a stream that lies in memory the Go runtime does not move or free -- a mapped file
// (syscall.Mmap), C memory -- and stays valid until the Reader is freed: the library reads it where it lies, no
// callbacks (xfg_reader_new_mem).  Do not pass memory of the Go heap: C keeps the pointer after the call returns.
func NewReaderMem(stream unsafe.Pointer, n int, conf *ReaderConfig) (*Reader, error) {
	defer pinned()()
	dev := 0
	if conf != nil {
		dev = conf.Device
	}
	xr := new(Reader)
	if rc := C.xfg_create(C.int(dev), &xr.ctx); rc != C.XFG_OK {
		return nil, errorf(errors.Internal, C.GoString(C.xfg_strerror(rc)))
	}
	runtime.SetFinalizer(xr, (*Reader).free)
	cls := C.xfg_reader_new_mem(xr.ctx, (*C.uint8_t)(stream), C.size_t(n), &xr.r)
	return xr, classErr(cls)
}

// Reset mirrors (*xflate.Reader).Reset (reader.go:124-157).
func (xr *Reader) Reset(rs io.ReadSeeker) error {
	defer pinned()()
	if xr.h != 0 {
		xr.h.Delete()
	}
	xr.h = cgo.NewHandle(&rsBox{rs})
	if xr.user == nil { // a Reader made by NewReaderMem
		xr.user = C.malloc(C.size_t(unsafe.Sizeof(uintptr(0))))
	}
	*(*uintptr)(xr.user) = uintptr(xr.h)
	rd, sk := C.xfg_read_fn(C.goXfgRead), C.xfg_seek_fn(C.goXfgSeek)
	if xr.r == nil {
		return classErr(C.xfg_reader_new(xr.ctx, rd, sk, xr.user, &xr.r))
	}
	return classErr(C.xfg_reader_reset(xr.r, rd, sk, xr.user))
}

// Read mirrors (*xflate.Reader).Read (reader.go:162-215).
func (xr *Reader) Read(buf []byte) (int, error) {
	defer pinned()()
	if len(buf) == 0 {
		return 0, nil
	}
	var cls C.int
	n := C.xfg_reader_read(xr.r, (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.int64_t(len(buf)), &cls)
	return int(n), classErr(cls)
}

// Seek mirrors (*xflate.Reader).Seek (reader.go:221-279).
func (xr *Reader) Seek(offset int64, whence int) (int64, error) {
	defer pinned()()
	if whence < 0 || whence > 2 {
		whence = 99 // the C side reports errors.Invalid
	}
	var cls C.int
	pos := C.xfg_reader_seek(xr.r, C.int64_t(offset), C.int(whence), &cls)
	return int64(pos), classErr(cls)
}

// Close mirrors (*xflate.Reader).Close (reader.go:282-291): idempotent, returns the sticky error.
func (xr *Reader) Close() error {
	defer pinned()()
	return classErr(C.xfg_reader_close(xr.r))
}

// SetWindow bounds how many decoded bytes one device launch may produce ahead of the reads
// (default 256 MiB).  After a Seek the window restarts at one chunk and doubles while reads stay
// sequential, which keeps the I/O budget of xflate/reader_test.go:392,504.
func (xr *Reader) SetWindow(maxRawBytes int64) { C.xfg_reader_set_window(xr.r, C.int64_t(maxRawBytes)) }

func (xr *Reader) free() {
	if xr.r != nil {
		C.xfg_reader_free(xr.r)
		xr.r = nil
	}
	if xr.ctx != nil {
		C.xfg_destroy(xr.ctx)
		xr.ctx = nil
	}
	if xr.h != 0 {
		xr.h.Delete()
		xr.h = 0
	}
	if xr.user != nil {
		C.free(xr.user)
		xr.user = nil
	}
}
