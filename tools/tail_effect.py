#!/usr/bin/env python3
"""How much of a launch is its tail?  C2's chunk mix at 1x, 2x and 4x the chunks per launch (4, 8, 16 chunks per
wave of the persistent grid): the decompressed rate would be flat if the grid drained instantly.  Arguments name
the descriptor orders to try: "stream" (as the stream holds them, the default), "longest-first" and
"shortest-first" (by compressed size; every output stays where it was)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys
sys.path.insert(0, %r)
import bench
bench.WORKLOADS["c2xN"] = (%d, 65536, ("huffman", "repeats", "digits"), 6, 768)
ORDER = %r
if ORDER != "stream":
    _build = bench.build_shard
    def build_sorted(name, rank):
        from compress_amd import capi
        sh = _build(name, rank)
        idx = sorted(range(sh["n"]), key=lambda i: sh["descs"][i].comp_size, reverse=(ORDER == "longest-first"))
        d = (capi.Chunk * sh["n"])()
        for k, i in enumerate(idx):
            d[k] = sh["descs"][i]
        sh["descs"] = d
        return sh
    bench.build_shard = build_sorted
sys.argv = ["bench.py", "--workload", "c2xN", "--no-cpu", "--no-verify", "--steps", "5", "--warmup", "2"]
bench.main()
'''
ORDERS = sys.argv[1:] or ["stream"]
for order in ORDERS:
    for mult in (1, 2, 4):
        p = subprocess.run([sys.executable, "-c", CODE % (ROOT, 16384 * mult, order)], capture_output=True, text=True,
                           timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("%s x%d failed:" % (order, mult), p.stderr[-500:])
            sys.exit(1)
        j = json.loads(line[-1])
        print("%-14s %6d chunks per launch: %.1f GB/s decompressed, kernel %.3f ms"
              % (order, 16384 * mult, j["value"], j["roofline"]["kernel_ms"]), flush=True)
