#!/bin/bash
# usage: tools/variant_traffic.sh <workload> "<variants ('base' = the shipped library)>"  -- speed and fabric traffic of tuning variants of the library
w=${1:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $2; do
  lib=$R/compress_amd/lib/libxflate_amd_$v.so
  [ "$v" = base ] && lib=$R/compress_amd/lib/libxflate_amd.so
  export XFG_LIB=$lib
  out=$R/gpurun_out/variant_$v
  mkdir -p $out
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 170 rocprofv3 --pmc $c -d $out/$c -o $c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --workload $w > $out/$c.log 2>&1 || { echo "pass $c failed"; exit 1; }
  done
  echo "== $v"
  python3 $R/tools/pmc_summary.py $out
  for k in 1 2; do timeout -k 10 170 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'GB/s', d['roofline']['kernel_ms'], 'ms')"; done
  find $out -name "*.csv" -delete
done
