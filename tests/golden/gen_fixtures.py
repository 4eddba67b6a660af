#!/usr/bin/env python3
"""Generate tests/golden/*.json from the reference's own test vectors.

Run in the build container only (needs /root/reference, which does not exist
on the GPU box):

    python tests/golden/gen_fixtures.py

It reads the reference's Go test files AS TEXT, pulls out the vector tables
(inputs, expected outputs, expected offsets / error class), evaluates the few
literal forms they use (hex helpers, BitGen strings, []byte("..."),
make([]byte, n)) and writes DATA ONLY: hex in, hex out, integers, names.
No reference source text is stored.

Sources (all under /root/reference):
  flate/reader_test.go:54-833                 -> flate_reader.json
  xflate/reader_test.go:41-275, 306-316       -> xflate_reader.json
  xflate/internal/meta/reader_test.go:38-431  -> meta_reader.json
  xflate/internal/meta/reader_test.go:480-582 -> meta_reader.json ("streams")
  xflate/writer_test.go:25-112                -> xflate_writer.json
  xflate/index_test.go                        -> index_tables.json (the tables transcribed as data, by hand: three Go struct
                                                 literals; tests/test_index.py and tests/test_host_cpp_vectors.py read them)
  internal/testutil/bitgen_test.go:14-237     -> bitgen.json
  internal/prefix/prefix_test.go (fixed DEFLATE code sets) -> checked in tests by RFC 1951 tables
  testdata/*.{txt,bin}                        -> sha256 + sizes in corpora.json;
                                                 twain.txt & small slices copied as data files
"""
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.bitgen import decode_bitgen, BitGenError, _unquote_go  # noqa: E402

REF = "/root/reference"


# ----------------------------------------------------------------------------
# A tiny evaluator for the Go literal forms used in the vector tables.
# ----------------------------------------------------------------------------
class GoLit:
    def __init__(self, text, consts=None):
        self.s = text
        self.i = 0
        self.consts = consts or {}

    def ws(self):
        while self.i < len(self.s):
            if self.s[self.i] in " \t\r\n":
                self.i += 1
            elif self.s.startswith("//", self.i):
                j = self.s.find("\n", self.i)
                self.i = len(self.s) if j < 0 else j
            else:
                break

    def peek(self):
        self.ws()
        return self.s[self.i] if self.i < len(self.s) else ""

    def expect(self, tok):
        self.ws()
        if not self.s.startswith(tok, self.i):
            raise ValueError("expected %r at %d: %r" % (tok, self.i, self.s[self.i:self.i + 40]))
        self.i += len(tok)

    def string(self):
        """Go string literal (raw or interpreted), with '+' concatenation. -> (bytes, was_raw_text)"""
        parts = []
        while True:
            self.ws()
            c = self.s[self.i]
            if c == "`":
                j = self.s.index("`", self.i + 1)
                parts.append(self.s[self.i + 1:j].encode("utf-8"))
                self.i = j + 1
            elif c == '"':
                j = self.i + 1
                while self.s[j] != '"':
                    j += 2 if self.s[j] == "\\" else 1
                parts.append(_unquote_go(self.s[self.i + 1:j]))
                self.i = j + 1
            else:
                raise ValueError("string expected at %d: %r" % (self.i, self.s[self.i:self.i + 40]))
            self.ws()
            if self.s.startswith("+", self.i):
                self.i += 1
                continue
            return b"".join(parts)

    def ident(self):
        self.ws()
        m = re.compile(r"[A-Za-z_][A-Za-z_0-9.]*").match(self.s, self.i)
        if not m:
            return None
        self.i = m.end()
        return m.group(0)

    def value(self):
        v = self._value()
        # optional slice expression v[a:b] on byte strings
        while isinstance(v, bytes) and self.peek() == "[":
            m = re.compile(r"\[(\d*):(\d*)\]").match(self.s, self.i)
            if not m:
                break
            self.i = m.end()
            a = int(m.group(1)) if m.group(1) else None
            b = int(m.group(2)) if m.group(2) else None
            v = v[a:b]
        return v

    def _value(self):
        c = self.peek()
        if c in "`\"":
            return self.string()
        if c == "-" or c.isdigit():
            m = re.compile(r"-?(0x[0-9a-fA-F]+|\d+)").match(self.s, self.i)
            self.i = m.end()
            return int(m.group(0), 0)
        if self.s.startswith("[]byte", self.i):
            self.i += 6
            c = self.peek()
            if c == "(":
                self.expect("(")
                v = self.string()
                self.expect(")")
                return v
            self.expect("{")
            out = bytearray()
            while self.peek() != "}":
                out.append(self.value())
                if self.peek() == ",":
                    self.i += 1
            self.expect("}")
            return bytes(out)
        name = self.ident()
        if name is None:
            raise ValueError("value expected at %d: %r" % (self.i, self.s[self.i:self.i + 40]))
        if name in ("db", "testutil.MustDecodeBitGen"):
            self.expect("(")
            v = self.string()
            if self.peek() == ",":
                self.i += 1
            self.expect(")")
            return decode_bitgen(v.decode("utf-8"))
        if name in ("dh", "testutil.MustDecodeHex"):
            self.expect("(")
            v = self.string()
            if self.peek() == ",":
                self.i += 1
            self.expect(")")
            return bytes.fromhex(v.decode("ascii"))
        if name == "make":
            self.expect("(")
            self.expect("[]byte")
            self.expect(",")
            n = self.value()
            self.expect(")")
            return bytes(n)
        if name in ("true", "false"):
            return name == "true"
        if name in self.consts:
            return self.consts[name]
        return {"ident": name}


def split_entries(body):
    """Split '{...}, {...}' at depth 0, honouring strings and comments. body excludes the outer braces."""
    entries, depth, i, start = [], 0, 0, None
    n = len(body)
    while i < n:
        c = body[i]
        if c == "`":
            i = body.index("`", i + 1) + 1
            continue
        if c == '"':
            j = i + 1
            while body[j] != '"':
                j += 2 if body[j] == "\\" else 1
            i = j + 1
            continue
        if body.startswith("//", i):
            j = body.find("\n", i)
            i = n if j < 0 else j
            continue
        if c == "{":
            if depth == 0:
                start = i + 1
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                entries.append(body[start:i])
        i += 1
    return entries


def find_table(src, anchor, which=0):
    """Return the text between the outer '{' and '}' of the composite literal that
    follows the which-th occurrence of `anchor` (a '[]struct {' table)."""
    pos = -1
    for _ in range(which + 1):
        pos = src.index(anchor, pos + 1)
    # skip the struct type body
    i = src.index("{", pos)
    depth = 0
    while True:
        if src[i] == "{":
            depth += 1
        elif src[i] == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    # now the literal body
    i = src.index("{", i + 1)
    start = i + 1
    depth = 0
    n = len(src)
    while i < n:
        c = src[i]
        if c == "`":
            i = src.index("`", i + 1) + 1
            continue
        if c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            i = j + 1
            continue
        if src.startswith("//", i):
            i = src.index("\n", i)
            continue
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return src[start:i]
        i += 1
    raise ValueError("unterminated table")


def parse_keyed(entry, consts=None):
    g = GoLit(entry, consts)
    out = {}
    while g.peek():
        key = g.ident()
        g.expect(":")
        out[key] = g.value()
        if g.peek() == ",":
            g.i += 1
    return out


def parse_positional(entry, consts=None):
    g = GoLit(entry, consts)
    out = []
    while g.peek():
        out.append(g.value())
        if g.peek() == ",":
            g.i += 1
    return out


def hx(b):
    return (b or b"").hex()


def read(path):
    with open(os.path.join(REF, path), "r", encoding="utf-8") as f:
        return f.read()


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", path)


# ----------------------------------------------------------------------------
def gen_flate():
    src = read("flate/reader_test.go")
    body = find_table(src, "vectors := []struct")
    vecs = []
    for e in split_entries(body):
        d = parse_keyed(e)
        vecs.append({
            "name": d["name"].decode(),
            "input": hx(d.get("input")),
            "output": hx(d.get("output")),
            "inIdx": d.get("inIdx", 0),
            "outIdx": d.get("outIdx", 0),
            "errf": d.get("errf", b"").decode(),
        })
    dump("flate_reader.json", {
        "source": "flate/reader_test.go:54-833 (TestReader)",
        "vectors": vecs,
    })


def gen_xflate_reader():
    src = read("xflate/reader_test.go")
    body = find_table(src, "vectors := []struct")
    vecs = []
    for e in split_entries(body):
        d = parse_keyed(e)
        vecs.append({
            "desc": d["desc"].decode(),
            "input": hx(d.get("input")),
            "output": hx(d.get("output")),
            "errf": d.get("errf", b"").decode(),
        })
    # TestReaderReset constants (reader_test.go:306-316)
    reset = {}
    for name in ("empty", "badSize", "badData"):
        m = re.search(name + r"\s*=\s*testutil\.MustDecodeHex\(", src)
        g = GoLit(src[m.end():])
        reset[name] = hx(bytes.fromhex(g.string().decode("ascii")))
    dump("xflate_reader.json", {
        "source": "xflate/reader_test.go:41-275 (TestReader), :306-316 (TestReaderReset)",
        "vectors": vecs,
        "reset": reset,
    })


def gen_meta_reader():
    src = read("xflate/internal/meta/reader_test.go")
    consts = {"FinalNil": 0, "FinalMeta": 1, "FinalStream": 2}
    body = find_table(src, "vectors := []struct", 0)
    vecs = []
    for e in split_entries(body):
        d = parse_keyed(e, consts)
        vecs.append({
            "desc": d["desc"].decode(),
            "input": hx(d.get("input")),
            "output": hx(d.get("output")),
            "final": d.get("final", 0),
            "errf": d.get("errf", b"").decode(),
        })
    # TestReaderReset back-to-back streams (reader_test.go:480-582)
    m = re.search(r"data := testutil\.MustDecodeBitGen\(", src)
    g = GoLit(src[m.end():])
    data = decode_bitgen(g.string().decode("utf-8"))
    body = find_table(src, "vectors := []struct", 1)
    streams = []
    for e in split_entries(body):
        v = parse_positional(e, consts)
        streams.append({"data": hx(v[0]), "inOff": v[1], "outOff": v[2], "numBlks": v[3], "final": v[4]})
    dump("meta_reader.json", {
        "source": "xflate/internal/meta/reader_test.go:38-431 (TestReader), :480-582 (TestReaderReset)",
        "vectors": vecs,
        "streams_input": hx(data),
        "streams": streams,
    })


def gen_meta_writer():
    path = "xflate/internal/meta/writer_test.go"
    src = read(path)
    consts = {"FinalNil": 0, "FinalMeta": 1, "FinalStream": 2}
    body = find_table(src, "vectors := []struct", 0)
    vecs = []
    for e in split_entries(body):
        d = parse_keyed(e, consts)
        rec = {}
        for k, v in d.items():
            if isinstance(v, bytes):
                rec[k] = v.decode() if k in ("desc", "errf") else hx(v)
            elif isinstance(v, dict):
                rec[k] = v["ident"]
            else:
                rec[k] = v
        vecs.append(rec)
    dump("meta_writer.json", {"source": path + " (TestWriter)", "vectors": vecs})


def gen_xflate_writer():
    """xflate/writer_test.go:25-112.  Output is split into its labelled segments
    (the trailing // ChunkN, // IndexN, // Footer comments): chunk payload bytes
    depend on Go's stdlib compressor, index/footer bytes only on the meta encoder."""
    src = read("xflate/writer_test.go")
    body = find_table(src, "vectors := []struct")
    vecs = []
    for e in split_entries(body):
        desc = re.search(r'desc:\s*"([^"]*)"', e).group(1)
        conf = {}
        m = re.search(r"conf:\s*&WriterConfig\{([^}]*)\}", e)
        if m:
            for k, v in re.findall(r"(\w+):\s*([^,}]+)", m.group(1)):
                conf[k] = eval(v.strip(), {"__builtins__": {}})      # "1 << 10", "4", "3"
        m = re.search(r"input:\s*\[\]interface\{\}\{(.*?)\},\s*output:", e, re.S)
        toks = []
        g = GoLit(m.group(1))
        while g.peek():
            if g.s.startswith("br(", g.i):
                g.expect("br(")
                b = g.value()
                g.expect(",")
                n = g.value()
                g.expect(")")
                toks.append({"bytes": hx(b * n)})
            else:
                v = g.value()
                if isinstance(v, bytes):
                    toks.append({"bytes": hx(v)})
                else:
                    toks.append({"flush": v["ident"]})
            if g.peek() == ",":
                g.i += 1
        m = re.search(r"output:\s*dh\((.*)\)\s*,?\s*$", e, re.S)
        segs = []
        for hexs, label in re.findall(r'"([0-9a-f]*)"\s*[+,]?\s*//\s*(\w+)', m.group(1)):
            segs.append({"label": label, "hex": hexs})
        vecs.append({"desc": desc, "conf": conf, "input": toks, "segments": segs})
    dump("xflate_writer.json", {"source": "xflate/writer_test.go:25-112 (TestWriter)", "vectors": vecs})


def gen_bitgen():
    src = read("internal/testutil/bitgen_test.go")
    body = find_table(src, "vectors := []struct")
    vecs = []
    for e in split_entries(body):
        d = parse_keyed(e)
        vecs.append({
            "input": d.get("input", b"").decode("utf-8"),
            "output": hx(d.get("output")),
            "valid": bool(d.get("valid", False)),
        })
    dump("bitgen.json", {"source": "internal/testutil/bitgen_test.go:14-237", "vectors": vecs})


def gen_corpora():
    out = {}
    td = os.path.join(REF, "testdata")
    for name in sorted(os.listdir(td)):
        if not (name.endswith(".txt") or name.endswith(".bin")):
            continue
        with open(os.path.join(td, name), "rb") as f:
            data = f.read()
        out[name] = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
    dump("corpora.json", {"source": "testdata/*.{txt,bin}", "files": out})


if __name__ == "__main__":
    gen_bitgen()
    gen_flate()
    gen_xflate_reader()
    gen_meta_reader()
    gen_meta_writer()
    gen_xflate_writer()
    gen_corpora()
