#!/bin/bash
# rocprofv3 counter passes of buildABC on the 2^20 Poseidon shape, with and without the value dictionary, plus the
# calibration of FETCH_SIZE on the kernel's own access pattern:  gpurun -- bash tools/pmc_abc_session.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r05
case " $SESSION_SKIP " in *" abc "*) ;; *)
for d in 1 0; do
export G16_ABC_DICT=$d
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/abc_dict${d}_fetch -o r05 -- python3 tools/perf_poseidon.py --only abc > gpurun_out/pmc_r05/abc_dict${d}_fetch.out 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/abc_dict${d}_write -o r05 -- python3 tools/perf_poseidon.py --only abc > gpurun_out/pmc_r05/abc_dict${d}_write.out 2>&1
done
unset G16_ABC_DICT
G16_ABC_DICT=0 python tools/perf_poseidon.py --only abc > gpurun_out/r05_perf_poseidon_abc_nodict.txt 2>&1
python tools/perf_poseidon.py --only abc > gpurun_out/r05_perf_poseidon_abc_dict.txt 2>&1
;; esac
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/calib_fetch -o r05 -- python3 tools/pmc_abc_calib.py > gpurun_out/pmc_r05/calib_fetch.out 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_r05/calib_write -o r05 -- python3 tools/pmc_abc_calib.py > gpurun_out/pmc_r05/calib_write.out 2>&1
find gpurun_out/pmc_r05 -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
