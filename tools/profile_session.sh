#!/bin/bash
# One profiling session on the GPU box (round 4): the full -m gpu suite, rocprofv3 kernel-trace stats of the bench command
# and of the isolated kernels, the HBM-traffic and VALU counter passes, the instruction issue-rate micro-benchmark, the
# paired A/B of the round-final builds, the bench lines at 2^20 and 2^22.
#   gpurun -- bash tools/profile_session.sh   ->  gpurun_out/{prof_r04,pmc_r04}/..., processed by tools/pmc_traffic.py /
#   tools/valu_roofline.py into profiles/r04_*.json (which carry the hash of the library they were measured on)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r04 gpurun_out/pmc_r04
# SESSION_SKIP="tests 2p22": leave out the parts already run on this build in another call
case " $SESSION_SKIP " in *" tests "*) ;; *)
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r04_gpu_tests_final.log 2>&1
tail -4 gpurun_out/r04_gpu_tests_final.log;; esac
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/bench -o r04 -- python3 bench.py --no-cpu-baseline > gpurun_out/r04_rocprofv3_bench_line_2p20.json 2> gpurun_out/prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04/iso -o r04 -- python3 tools/perf.py --only reg --log2n 20 --reps 5 > gpurun_out/prof_iso.out 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04/fetch -o r04 -- python3 bench.py --steps 2 --warmup 3 --repeats 1 --inflight 1 --no-cpu-baseline > gpurun_out/pmc_fetch.out 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r04/write -o r04 -- python3 bench.py --steps 2 --warmup 3 --repeats 1 --inflight 1 --no-cpu-baseline > gpurun_out/pmc_write.out 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r04/valu -o r04 -- python3 tools/perf.py --only reg --log2n 20 --reps 2 > gpurun_out/pmc_valu.out 2>&1
./tools/ubench_int > gpurun_out/r04_ubench_int_issue_rates.txt 2>&1
V=nim_groth16_amd/csrc/build_variants
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 bash tools/ab_rounds.sh 4 gpurun_out/r04_ab_rounds_final.txt r01=$V/libg16hip_r01.so:K r02=$V/libg16hip_r02.so r03=$V/libg16hip_r03.so r04=nim_groth16_amd/csrc/libg16hip.so > gpurun_out/r04_ab_final.log 2>&1
tail -6 gpurun_out/r04_ab_rounds_final.txt
python tools/perf.py --log2n 20 --reps 5 > gpurun_out/r04_kernel_times_2p20.txt 2>&1
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_2p20_final.json 2> gpurun_out/r04_bench_final.err
tail -3 gpurun_out/r04_bench_final.err; cat gpurun_out/r04_bench_2p20_final.json | cut -c1-400
case " $SESSION_SKIP " in *" 2p22 "*) ;; *)
timeout -k 10 900 python bench.py --log2n 22 --steps 48 --warmup 6 --no-cpu-baseline > gpurun_out/r04_bench_2p22.json 2> gpurun_out/r04_bench_2p22.err
cat gpurun_out/r04_bench_2p22.json | cut -c1-300;; esac
find gpurun_out/prof_r04 gpurun_out/pmc_r04 -type f | xargs ls -la | awk '{print $5, $9}'
