# same-box A/B of the per-proof launch order and stream priorities (G16_* knobs are read once per process)
for rep in 1 2; do
for cfg in "1 1 lhllln" "0 0 lhllln" "1 1 lhlllh" "1 1 llllln"; do
set -- $cfg
G16_QUOTIENT_FIRST=$1 G16_LANES_AFTER_QUOTIENT=$2 G16_STREAM_PRIO=$3 python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('qfirst=$1 after=$2 prio=$3', d['value'], d['value_witness_in_hbm'], d['proof_latency_ms_single_in_flight'])"
done; done
