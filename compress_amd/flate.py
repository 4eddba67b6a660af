"""flate -- the reference's flate.Reader API (flate/reader.go:14-97) over the GPU inflate kernel.

NewReader(r) takes a bytes-like object or a readable; what has been read of the stream
is handed to the device as one "chunk" (plain DEFLATE has no
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
    """flate.Reader (flate/reader.go:14-33).

    The source may be bytes-like or a readable object.  The "never read more than necessary" contract of
    compress.ByteReader / compress.BufferedReader (api.go:29-72; canary test flate/flate_test.go:70-92, sync test
    :124-203) is kept like this, the device being a batch decoder that wants all the input it can get:
      seekable source (seek()/tell(): BytesIO, files -- also when it has peek(), as open(path, "rb") has) -- read
          ahead in blocks, then seek back to the byte after the stream (the un-read);
      peekable source that cannot seek (has peek(), the analogue of BufferedReader: a buffered pipe) -- input is
          looked at with peek() and consumed with read() only up to InputOffset, so a byte after the final block
          stays unread;
      any other readable -- read ahead in blocks as the reference's own bufio wrapper does for a plain io.Reader
          (internal/prefix/reader.go:103-109); no promise about bytes after the stream.
    Input is pulled only when every decoded byte has been handed out, so a stream that ends in a sync flush is
    delivered up to the flush without another read of the source (TestSync).  Every pull decodes the stream
    from its start again (the kernel has no resume state), so a seekable source is read in blocks that double
    (1 MiB, 2 MiB, ...): a stream of n bytes is decoded about log2(n / 1 MiB) times and at most twice its length
    in all.  A source that cannot seek is pulled as it delivers (a pipe's reader must not wait for more than the
    writer has flushed), which is quadratic in the number of pulls: meant for streams of a few flushes.
    """
    BLOCK = 1 << 20

    def __init__(self):
        self.InputOffset = 0      # flate/reader.go:15
        self.OutputOffset = 0     # flate/reader.go:16
        self._out = b""           # decoded bytes of the stream so far
        self._pos = 0             # of which handed out
        self._err = None          # sticky error (reader.go:59-61)
        self._status = None       # outcome of the last decode: None = nothing decoded yet
        self._final_in = 0
        self._data = bytearray()  # input consumed from the source so far
        self._src = None
        self._kind = "bytes"
        self._start = 0
        self._src_eof = True
        self._ctx = None
        self._block = self.BLOCK  # bytes asked of a seekable source next
        self._outbuf = None       # reused output buffer
        self.decodes = 0          # calls of xfg_flate_decode so far (tests)

    def Reset(self, r, ctx=None):
        """Reset (reader.go:45-56)."""
        self.__init__()
        self._ctx = ctx or default_context()
        if hasattr(r, "read"):
            self._src, self._src_eof = r, False
            seekable = hasattr(r, "seek") and hasattr(r, "tell") and (not hasattr(r, "seekable") or r.seekable())
            if seekable:
                self._kind = "seek"
                self._start = r.tell()
            elif hasattr(r, "peek"):
                self._kind = "peek"
            else:
                self._kind = "plain"
        else:
            self._data = bytearray(bytes(r))

    # ---- input
    def _pull(self):
        """More input from the source -> the bytes to decode now (consumed + looked-at), or None at its end."""
        if self._src_eof:
            return None
        if self._kind == "peek":
            look = bytes(self._src.peek(self.BLOCK))
            if not look:
                self._src_eof = True
                return None
            self._look = look
            return bytes(self._data) + look
        if self._kind == "seek":
            # everything it has, up to the block: the blocks double, so the re-decodes add up to twice the stream
            blk = bytearray()
            while len(blk) < self._block:
                b = self._src.read(self._block - len(blk))
                if not b:
                    break
                blk += b
            self._block *= 2
        else:
            rd = getattr(self._src, "read1", None) or self._src.read
            blk = rd(self.BLOCK)
        if not blk:
            self._src_eof = True
            return None
        self._data += blk
        return bytes(self._data)

    def _settle(self, in_off, finished):
        """After a decode of consumed + looked-at bytes: consume what the decoder used (peekable source), or give
        back what it did not (seekable source, once the stream has ended)."""
        if self._kind == "peek":
            use = len(self._look) if not finished else max(0, min(len(self._look), in_off - len(self._data)))
            if use:
                got = self._src.read(use)
                self._data += got
            self._look = b""
        elif self._kind == "seek" and finished:
            self._src.seek(self._start + in_off)

    def _decode(self, data):
        L = capi.lib()
        cap = max(4096, 4 * len(data), 2 * len(self._out))
        while True:
            if self._outbuf is None or len(self._outbuf) < cap:
                self._outbuf = ctypes.create_string_buffer(cap)
            out = self._outbuf
            cap = len(out)
            self.decodes += 1
            i, o, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
            rc = L.xfg_flate_decode(self._ctx.handle, data, len(data), out, cap,
                                    ctypes.byref(i), ctypes.byref(o), ctypes.byref(e))
            if rc == capi.XFG_E_NOMEM and cap < (len(data) + 8) * 1100:
                cap *= 4                                   # DEFLATE expands at most 1032-fold: at most five rounds
                continue
            if rc == capi.XFG_E_LIMIT:
                raise errors.LimitError("a plain DEFLATE stream of more than %d compressed bytes is beyond the device "
                                        "path's single-chunk limit" % capi.MAX_CHUNK_COMP)
            capi.check(rc, "xfg_flate_decode")
            break
        self._out = ctypes.string_at(out, o.value)
        self._final_in = i.value
        self._status = e.value

    def _final_error(self):
        if self._status == capi.ERR_NONE:
            return errors.EOF()
        if self._status == capi.ERR_CORRUPTED:
            return errors.Error(errors.Corrupted, "flate")
        if self._status == capi.ERR_UNEXPECTED_EOF:
            return errors.UnexpectedEOF()
        return errors.Error(errors.Internal, "flate")

    def _advance(self):
        """Decode more.  Returns False when the stream's outcome is final (self._err is set)."""
        while True:
            if self._status is None and self._kind == "bytes":
                self._decode(bytes(self._data))
                data_all = True
            else:
                data = self._pull()
                if data is None:
                    if self._status is None:               # an empty source
                        self._decode(b"")
                    data_all = True
                else:
                    had = len(self._out)
                    self._decode(data)
                    data_all = False
                    finished = self._status != capi.ERR_UNEXPECTED_EOF
                    self._settle(self._final_in, finished)
                    if not finished:
                        if len(self._out) > had:
                            return True                    # new output: hand it out before asking for more input
                        continue
            if self._status != capi.ERR_UNEXPECTED_EOF or data_all:
                self.InputOffset = self._final_in
                self._err = self._final_error()
                return False

    def Read(self, n=-1):
        """Read (reader.go:58-88).  n < 0: ReadAll -> bytes (clean EOF) or raises after keeping
        the partial output in .partial."""
        if n is None or n < 0:
            while self._err is None and self._advance():
                pass
            data = self._out[self._pos:]
            self._pos = len(self._out)
            self.OutputOffset += len(data)
            if isinstance(self._err, errors.EOF):
                return data
            self.partial = data
            raise self._err
        while True:
            if self._pos < len(self._out):
                data = self._out[self._pos:self._pos + n]
                self._pos += len(data)
                self.OutputOffset += len(data)
                if self._pos >= len(self._out) and self._err is not None:
                    self.InputOffset = self._final_in
                return data
            if self._err is not None:
                raise self._err
            self._advance()

    read = Read

    def Close(self):
        """Close (reader.go:90-97): io.EOF and errClosed become errClosed and nil is returned; any other sticky
        error is returned (raised); what was decoded and not yet read is dropped."""
        self._pos = len(self._out)
        if isinstance(self._err, errors.EOF) or errors.IsClosed(self._err):
            self._err = _errClosed
            return
        if self._err is not None:
            raise self._err


def NewReader(r, conf=None, ctx=None):
    """flate.NewReader (reader.go:39-43): -> (Reader, None)."""
    zr = Reader()
    zr.Reset(r, ctx)
    return zr, None
