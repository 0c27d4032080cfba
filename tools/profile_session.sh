#!/bin/bash
# Profiling session on the GPU box (round 5), in PARTS of <= 20 minutes each (gpurun's limit per call):
#   gpurun -- bash tools/profile_session.sh <part>      part = tests | prof | pmc | ab | bench
# Outputs under gpurun_out/{prof_r05,pmc_r05}/ and gpurun_out/r05_*; processed by tools/pmc_traffic.py /
# tools/valu_roofline.py / tools/pmc_abc.py into profiles/r05_*.json (which carry the hash of the library they were
# measured on: bench.py reports counter-derived numbers only next to files of the same device code).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r05 gpurun_out/pmc_r05
V=nim_groth16_amd/csrc/build_variants
case "$1" in
tests)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r05_gpu_tests_final.log 2>&1
  tail -14 gpurun_out/r05_gpu_tests_final.log;;
prof)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/bench -o r05 -- python3 bench.py --no-cpu-baseline --no-poseidon-shape > gpurun_out/r05_rocprofv3_bench_line_2p20.json 2> gpurun_out/prof_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/iso -o r05 -- python3 tools/perf.py --only reg --log2n 20 --reps 5 > gpurun_out/prof_iso.out 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05/poseidon -o r05 -- python3 tools/perf_poseidon.py --only poseidon --no-oracle > gpurun_out/prof_poseidon.out 2>&1
  ./tools/ubench_int > gpurun_out/r05_ubench_int_issue_rates.txt 2>&1
  python tools/perf.py --log2n 20 --reps 5 > gpurun_out/r05_kernel_times_2p20.txt 2>&1
  find gpurun_out/prof_r05 -type f | xargs ls -la | awk '{print $5, $9}';;
pmc)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/fetch -o r05 -- python3 bench.py --steps 2 --warmup 3 --repeats 1 --inflight 1 --no-cpu-baseline --no-poseidon-shape > gpurun_out/pmc_fetch.out 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/write -o r05 -- python3 bench.py --steps 2 --warmup 3 --repeats 1 --inflight 1 --no-cpu-baseline --no-poseidon-shape > gpurun_out/pmc_write.out 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/valu -o r05 -- python3 tools/perf.py --only reg --log2n 20 --reps 2 > gpurun_out/pmc_valu.out 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/valu_abc -o r05 -- python3 tools/perf_poseidon.py --only abc > gpurun_out/pmc_valu_abc.out 2>&1
  ./tools/ubench_int > gpurun_out/r05_ubench_int_issue_rates.txt 2>&1
  SESSION_SKIP= bash tools/pmc_abc_session.sh > gpurun_out/r05_pmc_abc.log 2>&1
  find gpurun_out/pmc_r05 -name "*.csv" | xargs ls -la | awk '{print $5, $9}';;
ab)
  GPU_MAX_HW_QUEUES=8 timeout -k 10 1000 bash tools/ab_rounds.sh 4 gpurun_out/r05_ab_rounds_final.txt r01=$V/libg16hip_r01.so:K r02=$V/libg16hip_r02.so r03=$V/libg16hip_r03.so r04=$V/libg16hip_r04.so r05=nim_groth16_amd/csrc/libg16hip.so > gpurun_out/r05_ab_final.log 2>&1
  tail -7 gpurun_out/r05_ab_rounds_final.txt;;
bench)
  timeout -k 10 700 python bench.py > gpurun_out/r05_bench_2p20_final.json 2> gpurun_out/r05_bench_final.err
  tail -4 gpurun_out/r05_bench_final.err; cut -c1-400 gpurun_out/r05_bench_2p20_final.json
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_2p20_steps20.json 2> gpurun_out/r05_bench_steps20.err
  cut -c1-200 gpurun_out/r05_bench_2p20_steps20.json
  python tools/perf_poseidon.py > gpurun_out/r05_perf_poseidon_2p20_final.txt 2>&1
  grep -E "==|abc_|proofs/s" gpurun_out/r05_perf_poseidon_2p20_final.txt;;
*) echo "usage: profile_session.sh tests|prof|pmc|ab|bench"; exit 2;;
esac
