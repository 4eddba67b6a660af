"""flate.Reader over readable sources: the "never read more than necessary" contract (api.go:29-72) and the
sync-flush contract, mirrored from flate/flate_test.go:53-203 (TestRoundTrip's canary byte, TestSync's syncBuffer)
and flate/reader_test.go:956-979 (TestReaderReset).  The compressor is C zlib (the reference tests use Go's
compress/flate writer, which is not here): any DEFLATE writer with a sync flush serves."""
import io
import os
import zlib

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _twain():
    return open(os.path.join(GOLDEN, "twain.txt"), "rb").read()


def _resize(data, n):
    """testutil.ResizeData's shape (internal/testutil/util.go:24-45): repeat the input up to n bytes."""
    out = bytearray()
    k = 0
    while len(out) < n:
        out += bytes(b ^ (k & 0xff) for b in data[:min(len(data), n - len(out))]) if k else data[:n - len(out)]
        k += 1
    return bytes(out[:n])


CORPORA = {
    "nil": b"",
    "twain": None,
    "zeros": bytes(100000),
    "random": os.urandom(70000),
    "digits": (b"0123456789" * 7001)[:65537],
}


class _OnlyRead(object):
    """struct{ io.Reader }: a source with nothing but read."""
    def __init__(self, f):
        self._f = f

    def read(self, n=-1):
        return self._f.read(n)


class _PeekOnly(object):
    """A buffered source that cannot seek (a pipe behind a bufio.Reader): read and peek, nothing else."""
    def __init__(self, f):
        self._f = f

    def read(self, n=-1):
        return self._f.read(n)

    def peek(self, n=0):
        return self._f.peek(n)


@pytest.mark.parametrize("name", sorted(CORPORA))
@pytest.mark.parametrize("kind", ["seek", "peek", "buffered_file"])
def test_round_trip_canary(ctx, name, kind):
    """flate/flate_test.go:53-96: a canary byte after the final block must not be consumed."""
    from compress_amd import flate
    data = CORPORA[name] if CORPORA[name] is not None else _twain()
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = comp.compress(data) + comp.flush() + b"\x7a"
    src = io.BytesIO(stream)
    # (a BufferedReader over a seekable file has peek() AND seek(): it is read ahead and seeked back)
    rd_src = src if kind == "seek" else io.BufferedReader(src, buffer_size=4096)
    if kind == "peek":
        rd_src = _PeekOnly(rd_src)
    zr, err = flate.NewReader(rd_src, ctx=ctx)
    assert err is None
    assert zr._kind == {"seek": "seek", "peek": "peek", "buffered_file": "seek"}[kind]
    got = bytearray()
    from compress_amd import errors
    while True:
        try:
            got += zr.Read(2111)
        except errors.EOF:
            break
    assert bytes(got) == data
    assert zr.OutputOffset == len(data) and zr.InputOffset == len(stream) - 1
    assert zr.Close() is None
    assert rd_src.read(1) == b"\x7a", "Read consumed more data than necessary"


class _SyncBuffer(io.RawIOBase):
    """syncBuffer (flate/flate_test.go:98-120): records whether a read was tried when nothing was there."""
    def __init__(self):
        io.RawIOBase.__init__(self)
        self.buf = bytearray()
        self.blocked = False

    def readable(self):
        return True

    def write_bytes(self, b):
        self.buf += b

    def readinto(self, b):
        n = min(len(b), len(self.buf))
        if n == 0 and len(b) > 0:
            self.blocked = True
            return 0
        b[:n] = self.buf[:n]
        del self.buf[:n]
        return n

    def read(self, n=-1):
        if n is None or n < 0:
            n = len(self.buf)
        out = bytes(self.buf[:n])
        if not out and n > 0:
            self.blocked = True
        del self.buf[:len(out)]
        return out


@pytest.mark.parametrize("kind", ["Reader", "BufferedReader"])
def test_sync(ctx, kind):
    """flate/flate_test.go:122-203 (TestSync): after every flush all data written so far can be read, nothing is
    left unconsumed in the buffer and the reader never asks the empty buffer for more."""
    from compress_amd import flate
    prime = 13
    sizes = list(range(1, 100, 3))
    i = 1
    while i <= 1 << 16:
        sizes += [i, i + prime]
        i *= 4
    i = 1
    while i <= 10000:
        sizes += [i, i + prime]
        i *= 10
    data = _resize(_twain(), sum(sizes))
    buf = _SyncBuffer()
    src = _OnlyRead(buf) if kind == "Reader" else io.BufferedReader(buf, buffer_size=4096)
    wr = zlib.compressobj(6, zlib.DEFLATED, -15)
    zr, err = flate.NewReader(src, ctx=ctx)
    assert err is None
    off = 0
    for n in sizes:
        want = data[off:off + n]
        off += n
        buf.write_bytes(wr.compress(want) + wr.flush(zlib.Z_SYNC_FLUSH))
        got = bytearray()
        while len(got) < n:                                    # io.ReadAtLeast(rd, maxBuf, n)
            got += zr.Read(max(sizes))
        assert bytes(got) == want, "flushSize %d" % n
        assert len(buf.buf) == 0, "flushSize %d: unconsumed buffer data" % n
        assert not buf.blocked, "flushSize %d: attempted over-consumption of buffer" % n
    assert zr.OutputOffset == len(data)


def test_reader_reset(ctx):
    """flate/reader_test.go:956-979 (TestReaderReset), with readable sources."""
    from compress_amd import flate, errors
    rd = flate.Reader()
    assert rd.Close() is None                                   # Close of a zero Reader
    rd.Reset(io.BytesIO(b"garbage"), ctx)
    with pytest.raises(errors.Error) as ei:
        rd.Read()
    assert errors.IsCorrupted(ei.value)
    with pytest.raises(errors.Error) as ei:
        rd.Close()
    assert errors.IsCorrupted(ei.value)
    data = b"\x00\x0c\x00\xf3\xffhello, world\x01\x00\x00\xff\xff"
    rd.Reset(io.BytesIO(data), ctx)
    assert rd.Read() == b"hello, world"
    assert rd.Close() is None
    with pytest.raises(errors.Error) as ei:
        rd.Read(1)
    assert errors.IsClosed(ei.value)


def test_truncated_source(ctx):
    """A source that ends inside the stream: everything decoded so far is delivered, then io.ErrUnexpectedEOF."""
    from compress_amd import flate, errors
    data = _twain()[:50000]
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = comp.compress(data) + comp.flush()
    zr, _ = flate.NewReader(io.BytesIO(stream[:len(stream) // 2]), ctx=ctx)
    got = bytearray()
    with pytest.raises(errors.UnexpectedEOF):
        while True:
            got += zr.Read(4096)
    assert len(got) > 0 and data.startswith(bytes(got))


def test_large_file_is_decoded_a_bounded_number_of_times(ctx, tmp_path):
    """A multi-MiB DEFLATE file opened with open(path, "rb") (an io.BufferedReader: peek() and seek()): the Reader has no
    resume state on the device, so every pull decodes from the stream's start -- the pulls double in size, and the
    number of xfg_flate_decode calls stays logarithmic (it was one per 8 KiB peek).  The canary after the stream stays unread."""
    from compress_amd import flate, errors
    data = _twain() * 8 + os.urandom(4 << 20) + _twain()
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    stream = comp.compress(data) + comp.flush()
    assert len(stream) > (5 << 20)
    path = tmp_path / "big.deflate"
    path.write_bytes(stream + b"\x7a")
    with open(path, "rb") as f:
        zr, err = flate.NewReader(f, ctx=ctx)
        assert err is None and zr._kind == "seek"
        got = bytearray()
        while True:
            try:
                got += zr.Read(1 << 20)
            except errors.EOF:
                break
        assert bytes(got) == data
        assert (zr.InputOffset, zr.OutputOffset) == (len(stream), len(data))
        assert zr.decodes <= 5, zr.decodes                  # 1 + 2 + 4 MiB pulls, and room for an output-buffer retry
        assert f.read(1) == b"\x7a"
