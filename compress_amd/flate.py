"""flate -- the reference's flate.Reader API (flate/reader.go:14-97) over the GPU inflate kernel.

NewReader(r) takes a bytes-like object or a readable; the whole stream is handed
to the device as one "chunk" (plain DEFLATE has no
index, so there is no inter-chunk parallelism here; the per-stream kernels are
the same).  Semantics follow the in-repo flate package: InputOffset /
OutputOffset, errors.Corrupted vs io.ErrUnexpectedEOF, all output produced
before an error is delivered first.
"""
import ctypes

from . import capi, errors
from .xflate import default_context


class ReaderConfig(object):
    """flate.ReaderConfig (reader.go:35-37): no options."""


_errClosed = errors.Error(errors.Closed, "flate")


class Reader(object):
    def __init__(self):
        self.InputOffset = 0      # flate/reader.go:15
        self.OutputOffset = 0     # flate/reader.go:16
        self._out = b""
        self._pos = 0
        self._err = None
        self._decoded = True
        self._data = b""
        self._ctx = None

    def Reset(self, r, ctx=None):
        """Reset (reader.go:45-56)."""
        self.__init__()
        self._ctx = ctx or default_context()
        self._data = r.read() if hasattr(r, "read") else bytes(r)
        self._decoded = False

    def _decode(self):
        L = capi.lib()
        data = self._data
        cap = max(4096, 4 * len(data))
        while True:
            out = ctypes.create_string_buffer(cap)
            i, o, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
            rc = L.xfg_flate_decode(self._ctx.handle, data, len(data), out, cap,
                                    ctypes.byref(i), ctypes.byref(o), ctypes.byref(e))
            if rc == capi.XFG_E_NOMEM and cap < (len(data) + 8) * 1100:
                cap *= 4
                continue
            capi.check(rc, "xfg_flate_decode")
            break
        self._out = out.raw[:o.value]
        self._final_in = i.value
        if e.value == capi.ERR_NONE:
            self._final_err = errors.EOF()
        elif e.value == capi.ERR_CORRUPTED:
            self._final_err = errors.Error(errors.Corrupted, "flate")
        elif e.value == capi.ERR_UNEXPECTED_EOF:
            self._final_err = errors.UnexpectedEOF()
        else:
            self._final_err = errors.Error(errors.Internal, "flate")
        self._decoded = True

    def Read(self, n=-1):
        """Read (reader.go:58-88).  n < 0: ReadAll -> bytes (clean EOF) or raises after keeping
        the partial output in .partial."""
        if self._err is not None and self._pos >= len(self._out):
            raise self._err
        if not self._decoded:
            self._decode()
        if n is None or n < 0:
            data = self._out[self._pos:]
            self._pos = len(self._out)
            self.OutputOffset += len(data)
            self.InputOffset = self._final_in
            self._err = self._final_err
            if isinstance(self._err, errors.EOF):
                return data
            self.partial = data
            raise self._err
        if self._pos < len(self._out):
            data = self._out[self._pos:self._pos + n]
            self._pos += len(data)
            self.OutputOffset += len(data)
            if self._pos >= len(self._out):
                self.InputOffset = self._final_in
            return data
        self.InputOffset = self._final_in
        self._err = self._final_err
        raise self._err

    read = Read

    def Close(self):
        """Close (reader.go:90-97)."""
        self._out, self._pos = b"", 0
        if self._err is None and not self._decoded:
            return                     # nothing read yet: the reference returns zr.err == nil
        if self._err is None or isinstance(self._err, errors.EOF) or errors.IsClosed(self._err):
            self._err = _errClosed
            return
        raise self._err


def NewReader(r, conf=None, ctx=None):
    """flate.NewReader (reader.go:39-43): -> (Reader, None)."""
    zr = Reader()
    zr.Reset(r, ctx)
    return zr, None
