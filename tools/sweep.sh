#!/bin/bash
# usage: tools/sweep.sh "base _a _b ..." "workload1 workload2 ..."
# Times bench.py workloads against library variants (base = the shipped libxflate_amd.so,
# _x = compress_amd/lib/libxflate_amd_x.so built with `make variant NAME=x EXTRA=...`).
for lib in $1; do
  suffix=$lib
  [ "$lib" = "base" ] && suffix=""
  for w in $2; do
    XFG_LIB=$PWD/compress_amd/lib/libxflate_amd$suffix.so timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --no-cpu --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$w', d['value'], 'GB/s', d['ms_per_step'], 'ms')" || exit 1
  done
done
