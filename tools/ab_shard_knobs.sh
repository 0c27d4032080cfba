#!/bin/bash
# per-rank cost of a shard of a 2^20 proof (tools/perf_shard.py) under the knobs that shape its latency chains
cd "${GRAFT_REPO_ROOT:-.}"
export PERF_SHARD_COUNTS=${PERF_SHARD_COUNTS:-4,8}
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 tools/perf_shard.py 20 2>&1 | grep shard_count; }
run X=default
run G16_G2_FIRST=0
run G16_RED_CHUNK=16
run G16_G2_FIRST=0 G16_RED_CHUNK=16
run G16_RED_CHUNK=2
run G16_RED_CHUNK=8
