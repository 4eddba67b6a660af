"""BitGen decoder -- TEST INFRASTRUCTURE ONLY (fixture generation time).

The reference writes its DEFLATE / meta golden vectors in a small bit-stream
DSL ("BitGen").  This is a from-scratch decoder for that grammar, used only by
tests/golden/gen_fixtures.py to turn the reference's vectors into bytes.

Grammar followed: /root/reference/internal/testutil/bitgen.go:18-115 (the
format description), semantics checked against the worked example at
bitgen.go:93-115 and the vectors of internal/testutil/bitgen_test.go (see
tests/test_bitgen.py).
"""
import re


class BitGenError(ValueError):
    pass


_WS = " \t\r\n\v\f"


def _unquote_go(s):
    """Interpret a Go double-quoted string body (no surrounding quotes) -> bytes."""
    out = bytearray()
    i = 0
    while i < len(s):
        c = s[i]
        if c != "\\":
            out += c.encode("utf-8")
            i += 1
            continue
        i += 1
        e = s[i]
        simple = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11,
                  "\\": 92, "'": 39, '"': 34}
        if e in simple:
            out.append(simple[e]); i += 1
        elif e == "x":
            out.append(int(s[i + 1:i + 3], 16)); i += 3
        elif e in "01234567":
            out.append(int(s[i:i + 3], 8)); i += 3
        elif e == "u":
            out += chr(int(s[i + 1:i + 5], 16)).encode("utf-8"); i += 5
        elif e == "U":
            out += chr(int(s[i + 1:i + 9], 16)).encode("utf-8"); i += 9
        else:
            raise BitGenError("bad escape \\%s" % e)
    return bytes(out)


class _Parser:
    def __init__(self, text):
        self.s = text
        self.i = 0

    def eof(self):
        return self.i >= len(self.s)

    def skip_ignored(self):
        while not self.eof():
            c = self.s[self.i]
            if c in _WS:
                self.i += 1
            elif c == "#":
                j = self.s.find("\n", self.i)
                self.i = len(self.s) if j < 0 else j
            else:
                break

    def check_delim(self):
        if not self.eof() and self.s[self.i] not in _WS + "#)":
            raise BitGenError("unexpected character %r at %d" % (self.s[self.i], self.i))

    def parse_repeat(self):
        if not self.eof() and self.s[self.i] == "*":
            m = re.compile(r"\*(\d+)").match(self.s, self.i)
            if not m:
                raise BitGenError("bad repeat at %d" % self.i)
            self.i = m.end()
            n = int(m.group(1))
            if n > (1 << 63) - 1:       # Go scans the count into an int: overflow is an error
                raise BitGenError("repeat count overflows int")
            return n
        return 1

    def parse_tokens(self):
        toks = []
        while True:
            self.skip_ignored()
            order = None
            if not self.eof() and self.s[self.i] in "<>":
                order = self.s[self.i]
                self.i += 1
            c = self.s[self.i] if not self.eof() else ""
            tok = None
            if c in ("0", "1"):
                m = re.compile(r"[01]+").match(self.s, self.i)
                bits = m.group(0)
                self.i = m.end()
                tok = ["bits", order, int(bits, 2), len(bits)]
            elif c == "D":
                m = re.compile(r"D(\d+):(\d+)").match(self.s, self.i)
                if not m:
                    raise BitGenError("bad D token at %d" % self.i)
                self.i = m.end()
                tok = ["bits", order, int(m.group(2)), int(m.group(1))]
            elif c == "H":
                m = re.compile(r"H(\d+):([0-9a-fA-F]+)").match(self.s, self.i)
                if not m:
                    raise BitGenError("bad H token at %d" % self.i)
                self.i = m.end()
                tok = ["bits", order, int(m.group(2), 16), int(m.group(1))]
            elif c == "X":
                m = re.compile(r"X:([0-9a-fA-F]*)").match(self.s, self.i)
                if not m or len(m.group(1)) % 2:
                    raise BitGenError("bad X token at %d" % self.i)
                self.i = m.end()
                if order:
                    raise BitGenError("order cannot be set on bytes value")
                tok = ["bytes", None, bytes.fromhex(m.group(1))]
            elif c == '"':
                j = self.i + 1
                while self.s[j] != '"':
                    j += 2 if self.s[j] == "\\" else 1
                body = self.s[self.i + 1:j]
                self.i = j + 1
                if order:
                    raise BitGenError("order cannot be set on bytes value")
                tok = ["bytes", None, _unquote_go(body)]
            elif c == "(":
                self.i += 1
                sub = self.parse_tokens()
                if self.eof() or self.s[self.i] != ")":
                    raise BitGenError("unterminated group")
                self.i += 1
                tok = ["group", order, sub]
            if tok is not None:
                tok.append(self.parse_repeat())
                toks.append(tok)
            elif order is not None:
                toks.append(["order", order])
            self.check_delim()
            if self.eof() or self.s[self.i] == ")":
                return toks


class _BitBuffer:
    def __init__(self, pack):
        self.pack = pack
        self.parse_mode = None
        self.b = bytearray()
        self.nbits = 0          # bits used in the last byte (0 = aligned)

    def write_bits(self, v, n):
        for k in range(n):
            if self.nbits == 0:
                self.b.append(0)
            if (v >> k) & 1:
                self.b[-1] |= 1 << self.nbits
            self.nbits = (self.nbits + 1) & 7

    def write_bytes(self, data):
        if self.nbits != 0:
            raise BitGenError("unaligned write")
        self.b += data

    def process(self, toks):
        for t in toks:
            kind = t[0]
            if kind == "order":
                self.parse_mode = t[1]
            elif kind == "bits":
                _, order, value, length, rep = t
                if length > 64 or value > (1 << length) - 1:
                    raise BitGenError("invalid bit value: D%d:%d" % (length, value))
                if order == ">" or (order is None and self.parse_mode == ">"):
                    value = int(format(value, "0%db" % length)[::-1], 2) if length else 0
                for _ in range(rep):
                    self.write_bits(value, length)
            elif kind == "bytes":
                _, _, data, rep = t
                if self.pack == ">":
                    data = bytes(_REV[x] for x in data)
                self.write_bytes(data * rep)
            elif kind == "group":
                _, order, sub, rep = t
                save = self.parse_mode
                if order is not None:
                    self.parse_mode = order
                for _ in range(rep):
                    self.process(sub)
                self.parse_mode = save

    def bytes(self):
        if self.pack == ">":
            return bytes(_REV[x] for x in self.b)
        return bytes(self.b)


_REV = [int(format(i, "08b")[::-1], 2) for i in range(256)]


def decode_bitgen(text):
    """Decode a BitGen string into bytes."""
    p = _Parser(text)
    p.skip_ignored()
    if not (p.s.startswith("<<<", p.i) or p.s.startswith(">>>", p.i)):
        raise BitGenError("missing bit-packing order")
    pack = p.s[p.i]
    p.i += 3
    p.check_delim()
    toks = p.parse_tokens()
    if not p.eof():
        raise BitGenError("unmatched group terminator")
    bb = _BitBuffer(pack)
    bb.process(toks)
    return bb.bytes()
