#!/usr/bin/env python3
"""usage: traffic_collect.py <gpurun_out/tag> <build note> [profiles/r3/traffic.json]
Turns the per-workload counter summaries of tools/traffic_pass.sh into the JSON that bench.py's roofline.traffic reads,
and prints the per-config table of DESIGN.md section 4 (GB/s, achieved, frac, traffic / algorithmic)."""
import json
import os
import re
import sys

src, build = sys.argv[1], sys.argv[2]
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r3", "traffic.json")
try:
    out = json.load(open(dst))
except (OSError, ValueError):
    out = {}
out["_comment"] = ("Fabric-side traffic of the inflate kernel per launch from rocprofv3 --pmc (tools/traffic_pass.sh: one run per counter, "
                   "the bench.py command of the workload); KB as rocprofv3 reports them.  Calibration (profiles/r2/calibration_narrow_loads.txt): "
                   "FETCH_SIZE counts a scattered narrow load as one 64-byte request (1 : 1) and a wide coalesced 16-byte-per-lane read at half "
                   "its bytes; WRITE_SIZE is exact.  bench.py's roofline.traffic = FETCH_SIZE + comp_bytes / 2 + WRITE_SIZE.")
rows = []
for w in sorted(os.listdir(src)):
    p = os.path.join(src, w, "pmc_summary.txt")
    if not os.path.exists(p):
        continue
    c = {}
    for line in open(p):
        m = re.match(r"(\w+)\s+\d+\s+per dispatch\s+(\d+)", line)
        if m:
            c[m.group(1)] = int(m.group(2))
    b = json.loads(open(os.path.join(src, w, "bench.json")).read().strip().splitlines()[-1])
    comp = b["config"]["comp_bytes_per_gpu"]
    out[w] = {"build": build, "FETCH_SIZE_KB": c["FETCH_SIZE"], "WRITE_SIZE_KB": c["WRITE_SIZE"], "comp_bytes": comp,
              "source": "profiles/r3/traffic_%s_pmc_summary.txt" % w}
    traffic = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 + comp // 2
    algo = b["roofline"]["algorithmic_bytes_per_launch"]
    rows.append((w, b["value"], b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], b["roofline"]["waves_per_chunk"],
                 traffic / 1e9, algo / 1e9, traffic / algo))
json.dump(out, open(dst, "w"), indent=1)
print("| workload | decompressed GB/s | achieved GB/s | frac | kernel ms | waves/chunk | traffic GB | algorithmic GB | traffic / algorithmic |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %.1f | %.1f | %.4f | %.3f | %d | %.3f | %.3f | %.2f |" % r)
