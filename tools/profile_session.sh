#!/bin/bash
# One profiling session on the GPU box (round 3): the full -m gpu suite, rocprofv3 kernel-trace stats of the bench command
# and of the isolated kernels, the HBM-traffic and VALU counter passes, the instruction issue-rate micro-benchmark.
#   gpurun -- bash tools/profile_session.sh   ->  gpurun_out/{prof_r03,pmc_r03}/..., processed by tools/pmc_traffic.py /
#   tools/valu_roofline.py into profiles/r03_*.json (which carry the hash of the library they were measured on)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r03 gpurun_out/pmc_r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r03_gpu_tests_final.log 2>&1
tail -4 gpurun_out/r03_gpu_tests_final.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/bench -o r03 -- python3 bench.py --no-cpu-baseline > gpurun_out/r03_rocprofv3_bench_line_2p20.json 2> gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03/iso -o r03 -- python3 tools/perf.py --only reg --log2n 20 --reps 5 > gpurun_out/prof_iso.out 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r03/fetch -o r03 -- python3 bench.py --steps 2 --warmup 3 --inflight 1 --no-cpu-baseline > gpurun_out/pmc_fetch.out 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r03/write -o r03 -- python3 bench.py --steps 2 --warmup 3 --inflight 1 --no-cpu-baseline > gpurun_out/pmc_write.out 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r03/valu -o r03 -- python3 tools/perf.py --only reg --log2n 20 --reps 2 > gpurun_out/pmc_valu.out 2>&1
./tools/ubench_int > gpurun_out/r03_ubench_int_issue_rates.txt 2>&1
find gpurun_out/prof_r03 gpurun_out/pmc_r03 -type f | xargs ls -la | awk '{print $5, $9}'
