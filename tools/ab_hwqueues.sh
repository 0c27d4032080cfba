# same-box A/B of the HIP runtime's hardware-queue limit (streams of a process are multiplexed onto this many queues)
for rep in 1 2; do
for q in default 8 16; do
if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q', d['value'], d['value_witness_in_hbm'], d['proof_latency_ms_single_in_flight'])"
done; done
