"""xflate_oracle.py -- TEST INFRASTRUCTURE ONLY.

Pure-Python restatement of the reference's host-side XFLATE read path, used to
check the product's C++ host layer (compress_amd/csrc/host_*.cpp) and, with
inflate_oracle.c underneath, the whole xflate.Reader contract.

What it follows (all under /root/reference):
  xflate/internal/meta/meta.go:28-44       magic, ReverseSearch
  xflate/internal/meta/reader.go:69-113    Reader.Read / Close (offsets, FinalMode)
  xflate/internal/meta/reader.go:118-257   decodeBlock
  xflate/index.go:14-139                   index / record / AppendRecord / AppendIndex / Search / GetRecords
  xflate/reader.go:32-70                   chunkReader (chunk || 01 00 00 ff ff, last-4-bytes sync)
  xflate/reader.go:124-157                 Reset
  xflate/reader.go:162-215                 Read
  xflate/reader.go:221-279                 Seek
  xflate/reader.go:282-291                 Close
  xflate/reader.go:296-347                 decodeIndexes
  xflate/reader.go:354-424                 decodeIndex
  xflate/reader.go:430-483                 decodeFooter
  xflate/common.go:52-57,114-134           magic, endBlock, error mapping

The per-chunk inflater of the reference is Go's stdlib compress/flate
(xflate/flate.go:9,43), which is not in /root/reference; here it is
oracle.inflate_chunk (inflate_oracle.c, the restatement of the in-repo flate
package).  On valid DEFLATE every RFC 1951 inflater agrees byte for byte; on
invalid input only the error class is pinned, by xflate/reader_test.go.

Parity pinning: tests/test_oracle_xflate.py runs this against
tests/golden/{meta_reader,xflate_reader}.json (every vector of
meta/reader_test.go TestReader + TestReaderReset and xflate/reader_test.go
TestReader + TestReaderReset).
"""
import binascii
import io

from . import inflate_chunk, OK, CORRUPTED, UNEXPECTED_EOF, OUT_FULL

# error codes (internal/errors/errors.go:36-56)
E_UNKNOWN, E_INTERNAL, E_INVALID, E_DEPRECATED, E_CORRUPTED, E_CLOSED = range(6)


class CompressError(Exception):
    def __init__(self, code, pkg, msg=""):
        Exception.__init__(self, "%s: code %d: %s" % (pkg, code, msg))
        self.code, self.pkg, self.msg = code, pkg, msg


class EOF(Exception):
    """io.EOF"""


class UnexpectedEOF(Exception):
    """io.ErrUnexpectedEOF"""


def is_corrupted(e):
    return isinstance(e, CompressError) and e.code == E_CORRUPTED


def is_invalid(e):
    return isinstance(e, CompressError) and e.code == E_INVALID


def is_closed(e):
    return isinstance(e, CompressError) and e.code == E_CLOSED


# ---------------------------------------------------------------- meta ------
MAGIC_VALS, MAGIC_MASK = 0x05860004, 0xfffe3fc6          # meta.go:28-31
MAX_SYMS = 257
MIN_REP_LAST, MIN_REP_ZERO = 3, 11
MAX_ENC_BYTES = 64
FINAL_NIL, FINAL_META, FINAL_STREAM = 0, 1, 2


def reverse_search(data):
    """meta.ReverseSearch (meta.go:35-44)."""
    magic = 0
    for i in range(len(data) - 1, -1, -1):
        magic = ((magic << 8) | data[i]) & 0xffffffff
        if magic & MAGIC_MASK == MAGIC_VALS:
            return i
    return -1


class _Bits:
    """LSB-first bit reader over bytes with the reference's failure rule:
    a read either gets all its bits or raises UnexpectedEOF consuming none."""

    def __init__(self, data, pos=0):
        self.d = data
        self.bit = pos * 8

    def avail(self):
        return len(self.d) * 8 - self.bit

    def read(self, n):
        if n > self.avail():
            raise UnexpectedEOF()
        v = 0
        for k in range(n):
            b = self.bit + k
            v |= ((self.d[b >> 3] >> (b & 7)) & 1) << k
        self.bit += n
        return v

    def offset(self):                 # prefix.Reader.Flush: ceil(bits/8)
        return (self.bit + 7) // 8


class MetaReader:
    """meta.Reader over an in-memory byte source starting at `pos`."""

    def __init__(self, data, pos=0):
        self.data = bytes(data)
        self.start = pos
        self.br = _Bits(self.data, pos)
        self.input_offset = 0
        self.output_offset = 0
        self.num_blocks = 0
        self.final_mode = FINAL_NIL
        self.final = FINAL_NIL
        self.buf = b""
        self.err = None

    def decode_block(self):
        """meta/reader.go:118-257.  Returns None or an exception instance (EOF included)."""
        off0 = self.br.offset()
        try:
            return self._decode_block()
        except (UnexpectedEOF, CompressError) as e:
            return e
        finally:
            self.input_offset += self.br.offset() - off0

    def _corrupt(self, msg):
        return CompressError(E_CORRUPTED, "meta", msg)

    def _decode_block(self):
        br = self.br
        if br.avail() < 1:
            return EOF()                                  # EOF is okay for first bit
        magic = br.read(32)
        if magic & MAGIC_MASK != MAGIC_VALS:
            raise self._corrupt("invalid meta magic value")
        final_stream = (magic >> 0) & 1 > 0
        pads = (magic >> 3) & 7
        num_hclen = 4 + ((magic >> 13) & 15)
        fail = num_hclen < 6
        for _ in range(5, num_hclen - 1):
            fail = fail or br.read(3) != 0
        fail = fail or br.read(3) != 2
        fail = fail or br.read(1) != 0
        if fail:
            raise self._corrupt("invalid meta header")
        huff_len = 8 - (num_hclen - 4) // 2
        huff_range = 1 << huff_len

        bits = [0]                                        # first symbol is symZero
        bit, ones, fifo, idx = 0, 0, 0xff, 0
        while idx < MAX_SYMS - 1:
            cnt = 1
            # decHuff (meta.go:126-137): 0 -> zero, 10 -> one, 110 -> repLast, 111 -> repZero
            # ReadSymbol looks up zero-extended bits and fails only if the code is longer than what is left.
            a = br.avail()
            b0 = self._peek(0)
            if a < 1:
                raise UnexpectedEOF()
            if b0 == 0:
                br.read(1)
                bit = 0
                fifo = (fifo >> 1) | (0 << 7)
            else:
                b1 = self._peek(1)
                if b1 == 0:
                    if a < 2:
                        raise UnexpectedEOF()
                    br.read(2)
                    bit = 1
                    fifo = (fifo >> 2) | (1 << 6)
                else:
                    if a < 3:
                        raise UnexpectedEOF()
                    b2 = self._peek(2)
                    br.read(3)
                    if b2 == 0:                           # symRepLast
                        val = br.read(2)
                        cnt = val + MIN_REP_LAST
                        fifo = (fifo >> 3) | (3 << 5)
                        fifo = ((fifo >> 2) | (val << 6)) & 0xff
                    else:                                 # symRepZero
                        bit = 0
                        val = br.read(7)
                        cnt = val + MIN_REP_ZERO
                        fifo = (fifo >> 3) | (7 << 5)
                        fifo = ((fifo >> 7) | (val << 1)) & 0xff
            fifo &= 0xff
            if fifo == 0x00:
                raise self._corrupt("invalid sequence of meta symbols")
            bits.extend([bit] * cnt)
            ones += bit * cnt
            idx += cnt
        if len(bits) != MAX_SYMS:
            raise self._corrupt("excessive number of meta symbols")
        bits.extend([0] * (-MAX_SYMS & 7))
        syms = bytearray(33)
        for i, b in enumerate(bits):
            if b:
                syms[i >> 3] |= 1 << (i & 7)
        idx_eob = MAX_SYMS - 1
        if ones != huff_range:
            raise self._corrupt("degenerate meta prefix tree")
        if syms[idx_eob // 8] & (1 << (idx_eob % 8)) == 0:
            raise self._corrupt("missing meta terminator symbol")
        flags = syms[0]
        final_meta = (flags >> 1) & 1 > 0
        invert = (flags >> 2) & 1 > 0
        size = (flags >> 3) & 31
        buf = bytes(syms[1:1 + size])
        if invert:
            buf = bytes(b ^ 0xff for b in buf)
        final = int(final_meta) + int(final_stream)
        if final_stream and not final_meta:
            raise self._corrupt("invalid combination of final bits")
        fail = False
        fail = fail or br.read(pads) > 0
        fail = fail or br.read(1) > 0
        fail = fail or br.read(huff_len) != huff_range - 1
        fail = fail or br.bit % 8 > 0
        if fail:
            raise self._corrupt("invalid meta footer")
        self.buf, self.final = buf, final
        self.num_blocks += 1
        return None

    def _peek(self, k):
        b = self.br.bit + k
        if b >= len(self.data) * 8:
            return 0
        return (self.data[b >> 3] >> (b & 7)) & 1

    def read_all(self):
        """ioutil.ReadAll(mr) / io.Copy(dst, mr): -> (data, err or None)."""
        out = bytearray()
        while True:
            if self.err is not None:
                break
            if self.final != FINAL_NIL:
                self.final_mode = self.final
                self.err = EOF()
                break
            e = self.decode_block()
            if e is not None:
                self.err = e
                break
            out += self.buf
            self.buf = b""
        self.output_offset += len(out)
        err = None if isinstance(self.err, EOF) else self.err
        return bytes(out), err

    def close(self):
        """meta/reader.go:101-113."""
        if is_closed(self.err):
            return None
        if self.err is not None and not isinstance(self.err, EOF):
            return self.err
        self.final_mode = self.final
        self.err = CompressError(E_CLOSED, "meta")
        return None


# --------------------------------------------------------------- index ------
UNKNOWN_TYPE, DEFLATE_TYPE, INDEX_TYPE, FOOTER_TYPE = 0, 1, 2, 3
MAX_INT64 = (1 << 63) - 1


class Index:
    """xflate/index.go:14-139.  Records are (CompOffset, RawOffset, Type) ENDING offsets."""

    def __init__(self):
        self.records = []
        self.back_size = 0
        self.index_size = 0

    def reset(self):
        self.records = []
        self.back_size = 0
        self.index_size = 0

    def last_record(self):
        return self.records[-1] if self.records else (0, 0, 0)

    def append_record(self, comp_size, raw_size, typ):
        if raw_size < 0 or comp_size < 0:
            return False
        lc, lr, _ = self.last_record()
        c, r = lc + comp_size, lr + raw_size
        if c > MAX_INT64 or r > MAX_INT64:        # int64 wrap => rec < lastRec
            return False
        self.records.append((c, r, typ))
        return True

    def append_index(self, other):
        pc, pr = 0, 0
        for i, (c, r, t) in enumerate(other.records):
            if not self.append_record(c - pc, r - pr, t):
                del self.records[len(self.records) - i:]
                return False
            pc, pr = c, r
        return True

    def search(self, offset):
        recs = self.records
        i, imin, imax = -1, 0, len(recs) - 1
        while imax >= imin and i == -1:
            imid = (imin + imax) // 2
            gte_curr = offset >= recs[imid][1]
            lt_next = imid + 1 >= len(recs) or offset < recs[imid + 1][1]
            if gte_curr and lt_next:
                i = imid
            elif gte_curr:
                imin = imid + 1
            else:
                imax = imid - 1
        return i + 1

    def get_records(self, i):
        recs = self.records
        if i > len(recs):
            i = len(recs)
        prev = (0, 0, 0)
        if 0 <= i - 1 < len(recs):
            prev = recs[i - 1]
        if 0 <= i < len(recs):
            curr = recs[i]
        else:
            curr = (prev[0], prev[1], UNKNOWN_TYPE)
        return prev, curr


def uvarint(buf):
    """encoding/binary.Uvarint -> (value, n); n == 0: buffer too small; n < 0: overflow."""
    x, s = 0, 0
    for i, b in enumerate(buf):
        if i == 10:
            return 0, -(i + 1)
        if b < 0x80:
            if i == 9 and b > 1:
                return 0, -(i + 1)
            return x | (b << s), i + 1
        x |= (b & 0x7f) << s
        s += 7
    return 0, 0


# -------------------------------------------------------------- reader ------
END_BLOCK = bytes([0x01, 0x00, 0x00, 0xff, 0xff])
XF_MAGIC = b"XF\x00"
ERR_CORRUPTED = CompressError(E_CORRUPTED, "xflate")
ERR_CLOSED = CompressError(E_CLOSED, "xflate")
SEEK_START, SEEK_CURRENT, SEEK_END = 0, 1, 2


class XflateReader:
    """xflate.Reader over an in-memory stream (the bytes.Reader case of the tests)."""

    def __init__(self, stream):
        self.err = None
        self.reset(stream)

    # -- Reset (reader.go:124-157) -------------------------------------------
    def reset(self, stream):
        self.rd = bytes(stream)
        self.rd_pos = 0
        self.bytes_read = 0             # countReadSeeker.N of reader_test.go:19-28
        self.idx = Index()
        self.ri = 0
        self.offset = 0
        self.discard = 0
        self.chk = (0, 0, 0)            # csize, rsize, typ
        self.cur = None                 # decoded state of the current chunk
        self.err = None
        try:
            back_size, foot_size = self._decode_footer()
            self._decode_indexes(back_size)
            if not self.idx.append_record(foot_size, 0, FOOTER_TYPE):
                raise ERR_CORRUPTED
        except (CompressError, UnexpectedEOF) as e:
            self.err = e
            return e
        _, e = self.seek(0, SEEK_START)
        self.err = e
        return e

    def _rd_read(self, n):
        b = self.rd[self.rd_pos:self.rd_pos + n]
        self.rd_pos += len(b)
        self.bytes_read += len(b)
        return b

    # -- decodeFooter (reader.go:430-483) ------------------------------------
    def _decode_footer(self):
        end = len(self.rd)
        if end > MAX_ENC_BYTES:
            end = MAX_ENC_BYTES
        self.rd_pos = len(self.rd) - end
        br = self._rd_read(end)
        i = reverse_search(br)
        if i < 0:
            raise ERR_CORRUPTED
        mr = MetaReader(br, i)
        data, err = mr.read_all()
        if err is not None:
            raise self._wrap(err)
        if mr.br.offset() != len(br) or mr.num_blocks != 1:
            raise ERR_CORRUPTED
        if mr.final_mode != FINAL_STREAM:
            raise ERR_CORRUPTED
        self.rd_pos -= mr.input_offset
        if len(data) < 3 or data[:3] != XF_MAGIC:
            raise ERR_CORRUPTED
        back_size, cnt = uvarint(data[3:])
        if cnt <= 0:
            raise ERR_CORRUPTED
        if len(data[3 + cnt:]) > 0:
            raise ERR_CORRUPTED
        if back_size > MAX_INT64:
            back_size -= 1 << 64          # int64(backSizeU64) wraps negative
        return back_size, mr.input_offset

    # -- decodeIndexes (reader.go:296-347) -----------------------------------
    def _decode_indexes(self, back_size):
        pos = self.rd_pos
        comp_size = 0
        idxs = []
        while True:
            new_pos = pos - (back_size + comp_size)
            # Go computes this in int64; emulate the wrap before the range test.
            new_pos = (new_pos + (1 << 63)) % (1 << 64) - (1 << 63)
            if new_pos < 0 or new_pos > pos:
                raise ERR_CORRUPTED
            pos = self.rd_pos = new_pos
            if back_size == 0:
                break
            idx = Index()
            idx.index_size = back_size
            self._decode_index(idx)
            idxs.append(idx)
            back_size, comp_size = idx.back_size, idx.last_record()[0]
        if pos != 0:
            raise ERR_CORRUPTED
        for idx in reversed(idxs):
            if not self.idx.append_index(idx):
                raise ERR_CORRUPTED
            if not self.idx.append_record(idx.index_size, 0, INDEX_TYPE):
                raise ERR_CORRUPTED

    # -- decodeIndex (reader.go:354-424) -------------------------------------
    def _decode_index(self, idx):
        br = self._rd_read(idx.index_size)
        self.rd_pos -= len(br)
        mr = MetaReader(br, 0)
        bw, err = mr.read_all()
        if err is not None:
            raise self._wrap(err)
        crc = 0
        if len(bw) > 4:
            crc = binascii.crc32(bw[:-4]) & 0xffffffff
        state = {"buf": bw, "err": False}

        def read_vli():
            x, n = uvarint(state["buf"])
            if n <= 0 or x > MAX_INT64:
                state["err"] = True
                return 0
            state["buf"] = state["buf"][n:]
            return x

        idx.back_size = read_vli()
        num_recs = read_vli()
        total_comp = read_vli()
        total_raw = read_vli()
        if state["err"]:
            raise ERR_CORRUPTED
        chunks = []
        for _ in range(num_recs):
            # a corrupt numRecs can be astronomically large; the reference keeps
            # appending zero chunks after the first VLI error and fails below.
            if state["err"]:
                chunks.append((0, 0))
                break
            c = read_vli()
            r = read_vli()
            chunks.append((c, r))
        rest = state["buf"]
        if len(rest) != 4 or int.from_bytes(rest, "little") != crc:
            raise ERR_CORRUPTED
        if mr.final_mode != FINAL_META:
            raise ERR_CORRUPTED
        if mr.input_offset != idx.index_size:
            raise ERR_CORRUPTED
        for c, r in chunks:
            if c <= 4:
                raise ERR_CORRUPTED
            if not idx.append_record(c, r, DEFLATE_TYPE):
                raise ERR_CORRUPTED
        lc, lr, _ = idx.last_record()
        if lc != total_comp or lr != total_raw:
            raise ERR_CORRUPTED

    @staticmethod
    def _wrap(err):
        """errWrap (common.go:123-134)."""
        if isinstance(err, CompressError):
            return CompressError(err.code, "xflate", err.msg)
        return err

    # -- Seek (reader.go:221-279) --------------------------------------------
    def seek(self, offset, whence):
        if self.err is not None and not isinstance(self.err, EOF):
            return 0, self.err
        end = self.idx.last_record()[1]
        if whence == SEEK_START:
            pos = offset
        elif whence == SEEK_CURRENT:
            pos = self.offset + offset
        elif whence == SEEK_END:
            pos = end + offset
        else:
            return 0, CompressError(E_INVALID, "xflate", "invalid whence: %d" % whence)
        if pos < 0:
            return 0, CompressError(E_INVALID, "xflate", "negative position: %d" % pos)

        discard = pos - self.offset
        remain = self.chk[1] - (self.cur["out_off"] if self.cur else 0)
        if discard > 0 and remain > 0 and discard < remain:
            self.offset, self.discard = pos, discard
            return pos, None

        prev, curr = self.idx.get_records(self.ri)
        if not (prev[1] <= pos <= curr[1]):
            self.ri = self.idx.search(pos)
            prev, curr = self.idx.get_records(self.ri)
        self.ri += 1
        if self.ri > len(self.idx.records):
            self.ri = len(self.idx.records)

        self.chk = (curr[0] - prev[0], curr[1] - prev[1], curr[2])
        self.offset, self.discard = pos, pos - prev[1]
        if pos > end:
            self.discard = end - prev[1]
        self.rd_pos = prev[0]
        # chunkReader.Reset + flateReader.Reset: a fresh lazily-decoded chunk.
        self.cur = {"comp_pos": prev[0], "decoded": False, "out": b"", "rd": 0,
                    "out_off": 0, "in_off": 0, "status": OK}
        self.err = None
        return pos, None

    def _decode_current(self):
        """Inflate the current chunk || endBlock in one go (the reference streams it)."""
        cur = self.cur
        if cur["decoded"]:
            return
        csize, rsize, _ = self.chk
        comp = self.rd[cur["comp_pos"]:cur["comp_pos"] + csize]
        self.bytes_read += len(comp)
        cur["comp"] = comp
        st, out, i, o = inflate_chunk(comp, rsize + 1)
        if st == OUT_FULL:
            st, out, i, o = inflate_chunk(comp, (len(comp) + 5) * 1100 + 64)
        cur.update(decoded=True, out=out, status=st, in_final=i, out_final=o)

    def _zr_read(self, n):
        """flateReader.Read: -> (bytes, err).  Data first, the stream's final error after it."""
        self._decode_current()
        cur = self.cur
        if cur["rd"] < len(cur["out"]):
            b = cur["out"][cur["rd"]:cur["rd"] + n]
            cur["rd"] += len(b)
            cur["out_off"] += len(b)
            return b, None
        cur["in_off"] = cur["in_final"]
        if cur["status"] == OK:
            return b"", EOF()
        if cur["status"] == CORRUPTED:
            return b"", ERR_CORRUPTED
        return b"", UnexpectedEOF()

    # -- Read (reader.go:162-215) --------------------------------------------
    def read(self, n):
        """-> (bytes, err)."""
        if self.err is not None:
            return b"", self.err
        if self.discard > 0:
            got = 0
            while got < self.discard:
                b, e = self._zr_read(self.discard - got)
                got += len(b)
                if e is not None:
                    if isinstance(e, EOF):
                        break
                    self.err = e
                    return b"", self.err
            if got != self.discard:
                self.err = ERR_CORRUPTED
                return b"", self.err
            self.discard = 0

        buf = b""
        while len(buf) == 0 and self.err is None:
            if n == 0:
                # zr.Read(empty) returns (0, nil) for ever in the reference's loop only if
                # data remains; mirror io.Reader convention: return immediately.
                return b"", None
            buf, e = self._zr_read(n)
            self.offset += len(buf)
            if isinstance(e, EOF):
                csize, rsize, typ = self.chk
                if typ == DEFLATE_TYPE and self.cur["comp"][-4:] != b"\x00\x00\xff\xff":
                    self.err = ERR_CORRUPTED
                    break
                if typ != FOOTER_TYPE:
                    csize += len(END_BLOCK)
                    self.chk = (csize, rsize, typ)
                if csize != self.cur["in_off"] or rsize != self.cur["out_off"]:
                    self.err = ERR_CORRUPTED
                    break
                _, e2 = self.seek(self.offset, SEEK_START)
                if e2 is not None:
                    self.err = e2
                    break
                if self.chk[2] == UNKNOWN_TYPE:
                    self.err = EOF()
            elif e is not None:
                self.err = e
        return buf, self.err

    def read_all(self):
        """ioutil.ReadAll(xr): -> (data, err or None)."""
        out = bytearray()
        while True:
            b, e = self.read(1 << 16)
            out += b
            if e is not None:
                return bytes(out), (None if isinstance(e, EOF) else e)

    # -- Close (reader.go:282-291) -------------------------------------------
    def close(self):
        if self.err is ERR_CLOSED:
            return None
        if self.err is not None and not isinstance(self.err, EOF):
            return self.err
        self.err = ERR_CLOSED
        return None
