# usage: bash tools/wg_bench.sh "workloads" "wave settings"
set -e
for w in ${1:-c2 small c5_dynamic_4m c4 c5_stored_4m text}; do
  for k in ${2:-1 0}; do
    echo -n "$w XFG_DECODE_WAVES=$k: "
    XFG_DECODE_WAVES=$k timeout -k 10 280 python bench.py --workload $w --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'GB/s', d['roofline']['kernel_ms'], 'ms', d['config']['verified_bit_exact'])"
  done
done
