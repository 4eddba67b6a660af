#!/bin/bash
# usage: tools/pmc.sh <workload> <tag>   -- rocprofv3 counter passes for the inflate kernel (one run per pass)
w=${1:-huffman}; tag=${2:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
pass() {
  name=$1; shift
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d $out/$name -o $name --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-verify --workload $w > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $out/$name.log; return 1; }
}
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA || exit 1
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU || exit 1
pass p3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE || exit 1
pass p4 FETCH_SIZE || exit 1
pass p5 WRITE_SIZE || exit 1
