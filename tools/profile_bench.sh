#!/bin/bash
# usage: tools/profile_bench.sh <tag> [workload]  -- the evidence behind bench.py's roofline line:
#   rocprofv3 --kernel-trace --stats of the bench command, then one counter pass each for
#   FETCH_SIZE, WRITE_SIZE and the SQ counters.  Results land in gpurun_out/<tag>/.
tag=${1:-prof}; w=${2:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $out/trace -o trace --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w > $out/bench_under_trace.log 2>&1 || { echo "trace failed"; tail -5 $out/bench_under_trace.log; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d $out/$name -o $name --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --workload $w > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE || exit 1
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU || exit 1
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS || exit 1
pass grbm GRBM_GUI_ACTIVE || exit 1
python3 $R/tools/pmc_summary.py $out > $out/pmc_summary.txt
cat $out/pmc_summary.txt
find $out/trace -name "*stats*" | head
