#!/usr/bin/env python3
"""How much of a launch is its tail?  C2's chunk mix at 1x, 2x and 4x the chunks per launch (4, 8, 16 chunks per
wave of the persistent grid): the decompressed rate would be flat if the grid drained instantly."""
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
sys.argv = ["bench.py", "--workload", "c2xN", "--no-cpu", "--no-verify", "--steps", "5", "--warmup", "2"]
bench.main()
'''
for mult in (1, 2, 4):
    p = subprocess.run([sys.executable, "-c", CODE % (ROOT, 16384 * mult)], capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("x%d failed:" % mult, p.stderr[-500:])
        sys.exit(1)
    j = json.loads(line[-1])
    print("%6d chunks per launch: %.1f GB/s decompressed, kernel %.3f ms" % (16384 * mult, j["value"], j["roofline"]["kernel_ms"]), flush=True)
