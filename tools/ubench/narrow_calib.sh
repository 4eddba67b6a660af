#!/bin/bash
# FETCH_SIZE of the narrow-load and the wide-load microbenchmark, one counter pass each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/narrow_calib
mkdir -p $out
for mode in 0 1; do
  for mib in 64 4096; do
    timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE -d $out/m${mode}_$mib -o p --output-format csv -- $R/tools/ubench/narrow_loads $mib 64 $mode > $out/m${mode}_$mib.log 2>&1 || { echo "pass failed"; tail -3 $out/m${mode}_$mib.log; exit 1; }
    echo "== mode $mode, buffer $mib MiB"; grep "^mode" $out/m${mode}_$mib.log
    python3 - <<PY
import csv,glob
for f in glob.glob("$out/m${mode}_$mib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "narrow" in r["Kernel_Name"] or "wide" in r["Kernel_Name"]:
            print("   FETCH_SIZE", r["Counter_Value"], "KB", r["Kernel_Name"][:20])
PY
  done
done
