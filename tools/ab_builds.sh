# same-box A/B of two builds of libg16hip.so (G16HIP_LIB selects the library): skew cases, then the bench twice each
#   bash tools/ab_builds.sh build_variants/libA.so build_variants/libB.so
for lib in "$@"; do echo "== $lib"; G16HIP_LIB=$PWD/$lib python tools/perf_skew.py 2>&1 | grep -v amdgpu.ids; done
for rep in 1 2; do
for lib in "$@"; do
G16HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 64 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['value_witness_in_hbm'], d['proof_latency_ms_single_in_flight'])"
done; done
