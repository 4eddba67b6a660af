"""XFLATE meta encoding, writer side (xflate/internal/meta/writer.go:17-275, meta.go:28-165).

Host-side tooling for the stream generator (compress_amd.xflate.Writer): it
turns index / footer bytes into empty dynamic DEFLATE blocks.  The read side
used by the product lives in csrc/host_reader.cpp (C++), the test oracle's in
oracle/xflate_oracle.py; this module is checked byte for byte against
meta/writer_test.go and xflate/writer_test.go (tests/test_writer.py).
"""
from . import errors

MAGIC_VALS = 0x05860004          # meta.go:29
MAX_SYMS = 257                   # meta.go:47
MIN_HUFF_LEN, MAX_HUFF_LEN = 1, 7
MIN_REP_LAST, MAX_REP_LAST = 3, 6
MIN_REP_ZERO, MAX_REP_ZERO = 11, 138
MAX_RAW_BYTES, ENSURE_RAW_BYTES = 31, 22
FinalNil, FinalMeta, FinalStream = 0, 1, 2

_ONES = [bin(i).count("1") for i in range(256)]


def _errorf(code, msg):
    return errors.Error(code, "meta", msg)


class _BitWriter:
    """LSB-first bit packer (the little-endian prefix.Writer)."""

    def __init__(self):
        self.acc = 0
        self.n = 0

    def write(self, v, nb):
        self.acc |= (v & ((1 << nb) - 1)) << self.n
        self.n += nb

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


class Writer:
    """meta.Writer (writer.go:17-36): Write / Close with FinalMode; appends encoded blocks to `out`."""

    def __init__(self, out=None):
        self.Reset(out)

    def Reset(self, out):
        self.out = out if out is not None else bytearray()
        self.InputOffset = self.OutputOffset = self.NumBlocks = 0
        self.FinalMode = FinalNil
        self.buf = bytearray()
        self.buf0s = self.buf1s = 0
        self.err = None

    # computeHuffLen, writer.go:119-130
    @staticmethod
    def _compute_huff_len(zeros, ones):
        inv = ones > zeros
        if inv:
            zeros, ones = ones, zeros
        for huff_len in range(MIN_HUFF_LEN, MAX_HUFF_LEN + 1):
            max_ones = 1 << huff_len
            if MAX_SYMS - max_ones >= zeros + 8 and max_ones >= ones + 8:
                return huff_len, inv
        return 0, False

    # Write, writer.go:62-94
    def Write(self, data):
        if self.err is not None:
            raise self.err
        for b in bytes(data):
            ones = _ONES[b]
            zeros = 8 - ones
            if len(self.buf) >= ENSURE_RAW_BYTES:
                huff_len, _ = self._compute_huff_len(self.buf0s + zeros, self.buf1s + ones)
                if huff_len == 0:
                    self._encode_block(FinalNil)
            self.buf0s += zeros
            self.buf1s += ones
            self.buf.append(b)
            self.InputOffset += 1
        return len(data)

    # Close, writer.go:98-114
    def Close(self):
        if self.err is not None:
            if errors.IsClosed(self.err):
                return
            raise self.err
        try:
            self._encode_block(self.FinalMode)
        except errors.Error as e:
            self.err = e
            raise
        self.err = _errorf(errors.Closed, "")

    # computeCounts, writer.go:139-182: +n = n one bits, -n = n zero bits
    @staticmethod
    def _compute_counts(buf, max_ones, final, invert):
        arr = bytearray(1 + len(buf))
        arr[1:] = buf
        if invert:
            for i in range(1, len(arr)):
                arr[i] ^= 0xff
        arr[0] = (int(final) << 1) | (int(invert) << 2) | (len(buf) << 3)
        cnts, pcnt, zeros, ones = [], 0, 0, 0
        for byte in arr:
            for k in range(8):
                bit = (byte >> k) & 1
                if (bit > 0) != (pcnt > 0):
                    cnts.append(pcnt)
                    pcnt = 0
                pcnt += bit * 2 - 1
            ones += _ONES[byte]
            zeros += 8 - _ONES[byte]
        if pcnt > 0:
            cnts.append(pcnt)
            pcnt = 0
        pcnt += -1 * (MAX_SYMS - max_ones - zeros)
        if pcnt < 0:
            cnts.append(pcnt)
            pcnt = 0
        pcnt += max_ones - ones
        cnts.append(pcnt)
        return cnts

    # encodeBlock, writer.go:188-275
    def _encode_block(self, final):
        bw = _BitWriter()
        huff_len, inv = self._compute_huff_len(self.buf0s, self.buf1s)
        if huff_len == 0:
            raise _errorf(errors.Invalid, "block too large to encode")
        num_hclen = 4 + (8 - huff_len) * 2
        magic = MAGIC_VALS | (1 if final == FinalStream else 0) | ((num_hclen - 4) << 13)
        bw.write(magic, 32)
        for _ in range(5, num_hclen - 1):
            bw.write(0, 3)
        bw.write(2, 3)
        bw.write(0, 1)
        cnts = self._compute_counts(self.buf, 1 << huff_len, final != FinalNil, inv)
        cnts[0] += 1                      # the first zero bit is part of the header
        pre = -1
        # encHuff (meta.go:126-137): zero '0', one '10', repLast '110', repZero '111', LSB-first
        i = 0
        while i < len(cnts):
            if cnts[i] == 0:
                i += 1
                continue
            sym = 1 if cnts[i] > 0 else 0
            cur = sym * 2 - 1
            cnt = cur * cnts[i]
            if cur < 0 and cnt >= MIN_REP_ZERO:
                val = min(MAX_REP_ZERO, cnt)
                bw.write(0b111, 3)
                bw.write(val - MIN_REP_ZERO, 7)
            elif pre == cur and cnt >= MIN_REP_LAST:
                val = min(MAX_REP_LAST, cnt)
                bw.write(0b011, 3)
                bw.write(val - MIN_REP_LAST, 2)
            else:
                val = 1
                if sym:
                    bw.write(0b01, 2)
                else:
                    bw.write(0, 1)
            cnts[i] -= cur * val
            pre = cur
        pads = -(bw.n + 1 + huff_len) & 7
        bw.write(0, pads)
        bw.write(0, 1)
        bw.write((1 << huff_len) - 1, huff_len)
        enc = bytearray(bw.bytes())
        enc[0] |= pads << 3
        self.out += enc
        self.OutputOffset += len(enc)
        self.buf = bytearray()
        self.buf0s = self.buf1s = 0
        self.NumBlocks += 1


def encode(data, final_mode):
    """Encode `data` as one meta substream (>= 1 blocks) ending with final_mode."""
    w = Writer()
    w.FinalMode = final_mode
    w.Write(data)
    w.Close()
    return bytes(w.out)
