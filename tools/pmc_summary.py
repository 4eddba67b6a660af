#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per kernel and counter.  usage: pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "inflate"
acc = defaultdict(float)
n = defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if sub not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        n[row["Counter_Name"]] += 1
for k in sorted(acc):
    print("%-28s %18.0f  per dispatch %16.0f  (%d rows)" % (k, acc[k], acc[k] / max(n[k], 1), n[k]))
