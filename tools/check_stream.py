#!/usr/bin/env python3
"""Debug aid: decode every record of an XFLATE stream on the GPU and compare per chunk with zlib.
usage: tools/check_stream.py [chunk_size] [file]   (default: tests/golden/twain.txt at 1 KiB chunks)"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from compress_amd import capi, xflate  # noqa: E402


def main():
    cs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "twain.txt")
    data = open(path, "rb").read()
    out = bytearray()
    xw = xflate.NewWriter(out, xflate.WriterConfig(ChunkSize=cs))
    xw.Write(data)
    xw.Close()
    stream = bytes(out)
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    ctx = capi.Context(0)
    descs = [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks]
    raw, res = ctx.decode(stream, descs, len(data))
    bad = 0
    for i, (c, r) in enumerate(zip(chunks, res)):
        want = zlib.decompressobj(-15).decompress(stream[c.comp_off:c.comp_off + c.comp_size] + b"\x01\x00\x00\xff\xff")
        got = raw.raw[c.raw_off:c.raw_off + r.out_produced]
        if r.status != 0 or got != want:
            bad += 1
            if bad <= 5:
                k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
                print("chunk %d: status %s in %d/%d out %d/%d first diff at %d" %
                      (i, capi.ST_NAMES[r.status], r.in_consumed, c.comp_size, r.out_produced, c.raw_size, k))
                print("   want", want[max(0, k - 8):k + 24])
                print("   got ", bytes(got[max(0, k - 8):k + 24]))
    print("%d chunks, %d bad" % (len(chunks), bad))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
