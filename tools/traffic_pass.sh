#!/bin/bash
# usage: tools/traffic_pass.sh <tag> "<workloads>"  -- the measurement behind bench.py's roofline.traffic, per workload:
#   two rocprofv3 counter passes of the bench command (FETCH_SIZE, WRITE_SIZE; one run per counter) and one plain
#   bench run for the kernel time.  Results land in gpurun_out/<tag>/<workload>/; tools/traffic_collect.py turns them
#   into profiles/r3/traffic.json.
tag=${1:-traffic}; ws=${2:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in $ws; do
  out=$R/gpurun_out/$tag/$w
  mkdir -p $out
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 170 rocprofv3 --pmc $c -d $out/$c -o $c --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --workload $w > $out/$c.log 2>&1 || { echo "pass $c of $w failed"; tail -5 $out/$c.log; exit 1; }
  done
  timeout -k 10 170 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --workload $w > $out/bench.json 2> $out/bench.err || { echo "bench of $w failed"; tail -5 $out/bench.err; exit 1; }
  python3 $R/tools/pmc_summary.py $out > $out/pmc_summary.txt
  echo "== $w"; cat $out/pmc_summary.txt
  # the raw per-dispatch CSVs are large: the summaries are what is kept
  find $out -name "*.csv" -delete
done
