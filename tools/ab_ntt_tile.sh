# same-box A/B of the two NTT geometries (G16_NTT_TILE is read once per process)
for t in 4096 2048 1024; do echo "== G16_NTT_TILE=$t"; G16_NTT_TILE=$t python -m pytest tests/test_gpu_ntt.py tests/test_gpu_prover.py -x -q 2>&1 | tail -1; G16_NTT_TILE=$t python tools/perf.py --only ntt --log2n 20 2>&1 | grep -E "==|wall"; done
for rep in 1 2; do
for t in 4096 2048 1024; do
G16_NTT_TILE=$t python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ntt_tile=$t', d['value'], d['value_witness_in_hbm'], d['proof_latency_ms_single_in_flight'])"
done; done
