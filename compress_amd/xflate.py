"""xflate -- the reference's XFLATE API surface over the MI355X decode path.

Reader side (the hot path): NewReader / Reader.Read / Seek / Close / Reset with the
semantics of /root/reference/xflate/reader.go:75-291.  The state machine, index
and meta decoding live in C++ behind the C ABI (csrc/host_reader.cpp); chunk
inflation runs on the GPU (csrc/inflate_kernel.hip).  This file is the thin
binding a Go maintainer would write in cgo (INTEGRATION.md shows that stub).

Writer side (needed early as the stream generator):
NewWriter / Writer.Write / Flush / Close following xflate/writer.go:19-298, with
zlib as the per-chunk DEFLATE compressor (the reference uses Go's stdlib
compressor, xflate/flate.go:83-95; any RFC 1951 encoder yields a valid stream)
and compress_amd.meta for index and footer.
"""
import binascii
import ctypes
import io
import zlib

from . import capi, errors, meta

# xflate/common.go:61-78
NoCompression, BestSpeed, DefaultCompression, BestCompression = -1, 1, 6, 9
DefaultChunkSize = 1 << 18
DefaultIndexSize = 1 << 12
# xflate/common.go:84-112
FlushSync, FlushFull, FlushIndex = 0, 1, 2

SeekStart, SeekCurrent, SeekEnd = 0, 1, 2

_errClosed = errors.Error(errors.Closed, "xflate")


def _errorf(code, msg=""):
    return errors.Error(code, "xflate", msg)


def _raise_class(cls):
    """Map an XFG_ERR_* class from the C ABI to the reference's error values."""
    if cls == capi.ERR_NONE:
        return
    if cls == capi.ERR_EOF:
        raise errors.EOF()
    if cls == capi.ERR_UNEXPECTED_EOF:
        raise errors.UnexpectedEOF()
    if cls == capi.ERR_CLOSED:
        raise _errClosed
    if cls in (capi.ERR_INTERNAL, capi.ERR_INVALID, capi.ERR_DEPRECATED, capi.ERR_CORRUPTED):
        raise _errorf(cls)
    if cls == capi.ERR_IO:
        raise errors.DeviceError("read/seek callback failed")
    if cls == capi.ERR_LIMIT:
        raise errors.LimitError("a chunk exceeds the device path's per-chunk size limit (XFG_MAX_CHUNK_COMP / XFG_MAX_CHUNK_RAW)")
    raise errors.DeviceError("device/transport failure: %s" % capi.lib().xfg_last_error().decode())


class ReaderConfig(object):
    """xflate.ReaderConfig (reader.go:99-101): no options."""


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = capi.Context(device)
    return ctx


class Reader(object):
    """xflate.Reader (reader.go:75-95): an io.ReadSeeker over an XFLATE stream."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        self._rs = None
        self._cbs = None
        self._ctx = None

    # --- callbacks handed to the C ABI (io.ReadSeeker of the caller)
    def _make_callbacks(self, rs):
        def rd(_user, buf, n):
            try:
                data = rs.read(n)
                if data is None:
                    return 0
                k = len(data)
                ctypes.memmove(buf, data, k)
                return k
            except Exception:
                return -1

        def sk(_user, off, whence):
            try:
                return rs.seek(off, whence)
            except Exception:
                return -1
        return capi.READ_FN(rd), capi.SEEK_FN(sk)

    def Reset(self, rs, ctx=None):
        """Reset (reader.go:124): returns None or raises the parse error (the Reader stays usable for Close).
        A bytes-like stream is read where it lies by the library (xfg_reader_new_mem), without callbacks."""
        if not (hasattr(rs, "read") and hasattr(rs, "seek")):
            L = capi.lib()
            data = bytes(rs)                              # kept alive for the reader
            if not self._h:
                self._ctx = ctx or default_context()
                cls = L.xfg_reader_new_mem(self._ctx.handle, data, len(data), ctypes.byref(self._h))
            else:
                cls = L.xfg_reader_reset_mem(self._h, data, len(data))
            self._rs, self._cbs = data, None
            _raise_class(cls)
            return
        self._rs = rs
        self._cbs = self._make_callbacks(rs)
        L = capi.lib()
        if not self._h:
            self._ctx = ctx or default_context()
            cls = L.xfg_reader_new(self._ctx.handle, self._cbs[0], self._cbs[1], None, ctypes.byref(self._h))
        else:
            cls = L.xfg_reader_reset(self._h, self._cbs[0], self._cbs[1], None)
        _raise_class(cls)

    def Read(self, n=-1):
        """Read (reader.go:162): up to n bytes (> 0 bytes), raises errors.EOF at the end.
        n < 0 reads to the end (ioutil.ReadAll) and returns b"" instead of raising EOF."""
        if not self._h:
            raise _errorf(errors.Invalid, "reader not initialised")
        L = capi.lib()
        if n is None or n < 0:
            out = bytearray()
            buf = ctypes.create_string_buffer(1 << 20)
            err = ctypes.c_int()
            while True:
                k = L.xfg_reader_read(self._h, buf, len(buf), ctypes.byref(err))
                if k > 0:
                    out += buf.raw[:k]
                    continue
                if err.value == capi.ERR_EOF:
                    return bytes(out)
                self._partial = bytes(out)
                _raise_class(err.value)
        buf = ctypes.create_string_buffer(max(n, 1))
        err = ctypes.c_int()
        k = L.xfg_reader_read(self._h, buf, n, ctypes.byref(err))
        if k == 0 and err.value != capi.ERR_NONE:
            _raise_class(err.value)
        return buf.raw[:k]

    read = Read

    def Seek(self, offset, whence=SeekStart):
        """Seek (reader.go:221)."""
        if not self._h:
            raise _errorf(errors.Invalid, "reader not initialised")
        err = ctypes.c_int()
        if whence not in (0, 1, 2):
            whence = 99
        offset = max(-(1 << 63), min((1 << 63) - 1, offset))
        pos = capi.lib().xfg_reader_seek(self._h, offset, whence, ctypes.byref(err))
        _raise_class(err.value)
        return pos

    seek = Seek

    def Close(self):
        """Close (reader.go:282): idempotent; raises the sticky error if there is one."""
        if not self._h:
            return
        _raise_class(capi.lib().xfg_reader_close(self._h))

    def records(self):
        """The merged index: [(CompOffset, RawOffset, Type)] ending offsets (xflate/index.go:27-31)."""
        L = capi.lib()
        n = L.xfg_reader_num_records(self._h)
        arr = (capi.Record * max(n, 1))()
        L.xfg_reader_get_records(self._h, arr, n)
        return [(arr[i].comp_offset, arr[i].raw_offset, arr[i].type) for i in range(n)]

    def bytes_read(self):
        return capi.lib().xfg_reader_bytes_read(self._h)

    def set_window(self, max_raw_bytes):
        capi.lib().xfg_reader_set_window(self._h, max_raw_bytes)

    def __del__(self):
        try:
            if self._h:
                capi.lib().xfg_reader_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


def NewReader(rs, conf=None, ctx=None):
    """xflate.NewReader (reader.go:113): -> (Reader, err).  Like the reference it returns the
    Reader together with the error when the index cannot be parsed."""
    xr = Reader()
    try:
        xr.Reset(rs, ctx)
    except (errors.Error, errors.UnexpectedEOF) as e:
        return xr, e
    return xr, None


# ---------------------------------------------------------------------------
# Writer (stream generator)
# ---------------------------------------------------------------------------
class WriterConfig(object):
    """xflate.WriterConfig (writer.go:40-71)."""

    def __init__(self, Level=0, ChunkSize=0, IndexSize=0):
        self.Level, self.ChunkSize, self.IndexSize = Level, ChunkSize, IndexSize


def _uvarint(x):
    out = bytearray()
    while x >= 0x80:
        out.append((x & 0x7f) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


class _Index(object):
    def __init__(self):
        self.Records = []          # (CompOffset, RawOffset) ending offsets
        self.BackSize = 0
        self.IndexSize = 0

    def last(self):
        return self.Records[-1] if self.Records else (0, 0)


class Writer(object):
    """xflate.Writer (writer.go:19-36) writing into a bytearray / file-like `wr`."""

    def __init__(self, wr, conf=None, strategy=zlib.Z_DEFAULT_STRATEGY):
        lvl, nchk, nidx = 0, 0, 0
        if conf is not None:
            lvl = conf.Level
            if conf.ChunkSize < 0:
                raise _errorf(errors.Invalid, "invalid chunk size: %d" % conf.ChunkSize)
            nchk = conf.ChunkSize
            nidx = -1 if conf.IndexSize < 0 else conf.IndexSize
        # newFlateWriter level mapping (xflate/flate.go:83-95): 0 -> default, -1 -> no compression
        if lvl == 0:
            lvl = 6
        elif lvl == -1:
            lvl = 0
        if not (0 <= lvl <= 9):
            raise _errorf(errors.Invalid, "invalid compression level: %d" % lvl)
        self._lvl, self._strategy = lvl, strategy
        self.nchk = nchk or DefaultChunkSize
        self.nidx = nidx or DefaultIndexSize
        self.Reset(wr)

    def Reset(self, wr):
        self.wr = wr
        self.InputOffset = self.OutputOffset = 0
        self.idx = _Index()
        self.err = None
        self._zreset()

    def _zreset(self):
        self._z = zlib.compressobj(self._lvl, zlib.DEFLATED, -15, 8, self._strategy)
        self._zin = self._zout = 0

    def _emit(self, b):
        if isinstance(self.wr, bytearray):
            self.wr += b
        else:
            self.wr.write(b)
        self.OutputOffset += len(b)

    # Write, writer.go:138-165
    def Write(self, buf):
        if self.err is not None:
            raise self.err
        buf = memoryview(bytes(buf))
        cnt = 0
        while len(buf) > 0:
            remain = self.nchk - self._zin
            if remain <= 0:
                self.Flush(FlushFull)
                continue
            remain = min(remain, len(buf))
            out = self._z.compress(buf[:remain])
            self._zin += remain
            self._zout += len(out)
            self._emit(out)
            buf = buf[remain:]
            cnt += remain
        self.InputOffset += cnt
        return cnt

    # Flush, writer.go:169-204
    def Flush(self, mode):
        if self.err is not None:
            raise self.err
        if mode == FlushSync:
            out = self._z.flush(zlib.Z_SYNC_FLUSH)
            if not out:
                # zlib swallows a sync flush with nothing new to flush; the reference's
                # compressor always emits the empty stored block (writer_test.go:32-38).
                out = b"\x00\x00\x00\xff\xff"
            self._zout += len(out)
            self._emit(out)
        elif mode == FlushFull:
            self.Flush(FlushSync)
            lc, lr = self.idx.last()
            self.idx.Records.append((lc + self._zout, lr + self._zin))
            self._zreset()
            if len(self.idx.Records) == self.nidx:
                self.Flush(FlushIndex)
        elif mode == FlushIndex:
            if self._zin + self._zout > 0:
                self.Flush(FlushFull)
            self._encode_index(self.idx)
            back = self.idx.IndexSize
            self.idx = _Index()
            self.idx.BackSize = back
        else:
            raise _errorf(errors.Invalid, "invalid flush mode: %d" % mode)

    # Close, writer.go:209-233
    def Close(self):
        if self.err is not None:
            if errors.IsClosed(self.err):
                return
            raise self.err
        if self._zout + self._zin > 0 or len(self.idx.Records) > 0:
            self.Flush(FlushIndex)
        self._encode_footer(self.idx.BackSize)
        self.err = _errClosed

    # encodeIndex, writer.go:238-277
    def _encode_index(self, index):
        body = bytearray()
        body += _uvarint(index.BackSize)
        body += _uvarint(len(index.Records))
        body += _uvarint(index.last()[0])
        body += _uvarint(index.last()[1])
        pc = pr = 0
        for c, r in index.Records:
            body += _uvarint(c - pc)
            body += _uvarint(r - pr)
            pc, pr = c, r
        crc = binascii.crc32(bytes(body)) & 0xffffffff
        body += crc.to_bytes(4, "little")
        enc = meta.encode(bytes(body), meta.FinalMeta)
        self._emit(enc)
        index.IndexSize = len(enc)

    # encodeFooter, writer.go:280-298
    def _encode_footer(self, back_size):
        mw = meta.Writer()
        mw.FinalMode = meta.FinalStream
        mw.Write(b"XF\x00" + _uvarint(back_size))
        mw.Close()
        if mw.NumBlocks != 1:
            raise _errorf(errors.Internal, "footer was not a single block")
        self._emit(bytes(mw.out))


class GpuWriter(Writer):
    """xflate.Writer whose chunks are compressed on the GPU (xfg_deflate, csrc/deflate_kernel.hip) instead of by
    zlib: the chunker, the index and the footer are the Writer's (writer.go:138-298).  Closed chunks wait in a
    queue and are compressed `batch_chunks` at a time in one launch (one wave per chunk: a launch per chunk would
    leave the GPU empty), so OutputOffset and the records lag behind InputOffset until the queue is drained --
    by a sync flush, an index, Close, or a full queue.  A sync flush inside a chunk compresses what the chunk holds
    so far as a segment of its own (every segment ends in 00 00 ff ff and starts without history: the
    concatenation is what a DEFLATE stream with sync flushes is).  Level 1 (BestSpeed) uses the fixed Huffman codes
    in one pass; every other level has the device build codes for each chunk (XFG_DEFLATE_DYNAMIC, two passes)."""

    def __init__(self, wr, conf=None, ctx=None, batch_chunks=1024):
        self._ctx = ctx
        self._batch = max(1, int(batch_chunks))
        self._dynamic = not (conf is not None and conf.Level == 1)
        Writer.__init__(self, wr, conf)

    def _zreset(self):
        self._pend = bytearray()           # bytes of the current chunk not yet compressed
        self._zin = self._zout = 0

    def Reset(self, wr):
        self._queue = []                   # closed chunks: (raw bytes, compressed bytes already emitted for it, raw size)
        Writer.Reset(self, wr)

    def _drain(self):
        if not self._queue:
            return
        ctx = self._ctx or default_context()
        todo = [q[0] for q in self._queue if len(q[0])]
        comp = iter(ctx.deflate_chunks(todo, self._dynamic))
        for raw, zout, zin in self._queue:
            c = next(comp) if len(raw) else b"\x00\x00\x00\xff\xff"    # the empty stored block of a sync flush
            self._emit(c)
            lc, lr = self.idx.last()
            self.idx.Records.append((lc + zout + len(c), lr + zin))
        self._queue = []

    def Write(self, buf):
        if self.err is not None:
            raise self.err
        buf = memoryview(bytes(buf))
        cnt = 0
        while len(buf) > 0:
            remain = self.nchk - self._zin
            if remain <= 0:
                self.Flush(FlushFull)
                continue
            remain = min(remain, len(buf))
            self._pend += buf[:remain]
            self._zin += remain
            buf = buf[remain:]
            cnt += remain
        self.InputOffset += cnt
        return cnt

    def Flush(self, mode):
        if self.err is not None:
            raise self.err
        if mode == FlushSync:
            self._drain()                  # bytes leave in stream order
            if self._pend:
                ctx = self._ctx or default_context()
                out = ctx.deflate_chunks([bytes(self._pend)], self._dynamic)[0]
                self._pend = bytearray()
            else:
                out = b"\x00\x00\x00\xff\xff"
            self._zout += len(out)
            self._emit(out)
        elif mode == FlushFull:
            self._queue.append((bytes(self._pend), self._zout, self._zin))
            self._zreset()
            if len(self.idx.Records) + len(self._queue) == self.nidx:
                self.Flush(FlushIndex)
            elif len(self._queue) >= self._batch:
                self._drain()
        elif mode == FlushIndex:
            if self._zin + self._zout > 0:
                self.Flush(FlushFull)      # (may itself write a full index; an empty one follows, as in the reference)
            self._drain()
            self._encode_index(self.idx)
            back = self.idx.IndexSize
            self.idx = _Index()
            self.idx.BackSize = back
        else:
            raise _errorf(errors.Invalid, "invalid flush mode: %d" % mode)

    def Close(self):
        if self.err is not None:
            if errors.IsClosed(self.err):
                return
            raise self.err
        if self._zout + self._zin > 0 or len(self.idx.Records) > 0 or self._queue:
            self.Flush(FlushIndex)
        self._encode_footer(self.idx.BackSize)
        self.err = _errClosed


def NewWriter(wr, conf=None, compressor="zlib", ctx=None):
    """xflate.NewWriter (writer.go:78-104).  compressor="gpu": chunks are compressed by xfg_deflate (GpuWriter)."""
    if compressor == "gpu":
        return GpuWriter(wr, conf, ctx=ctx)
    if compressor != "zlib":
        raise _errorf(errors.Invalid, "unknown compressor: %r" % (compressor,))
    return Writer(wr, conf)
