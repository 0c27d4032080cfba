#!/bin/bash
# Paired same-box A/B of several builds of libg16hip.so: ONE session, ONE key file, ONE binary
# (tools/ab_prove.cpp dlopen()s each build), the builds ALTERNATING for `alts` rounds so that clock / thermal drift of
# the box hits all of them alike.  Prints every batch and, at the end, median / min / max proofs/s per build.
#   bash tools/ab_rounds.sh <alts> <out.txt> name=path[:K][@VAR=val,VAR=val] ...
#   ":K" = one key per context (round 1's rule); "@..." = G16_* knobs for that entry (same library, other settings)
# e.g. bash tools/ab_rounds.sh 5 gpurun_out/ab.txt r01=nim_groth16_amd/csrc/build_variants/libg16hip_r01.so:K \
#        r02=nim_groth16_amd/csrc/build_variants/libg16hip_r02.so r03=nim_groth16_amd/csrc/libg16hip.so
set -e
alts=$1; out=$2; shift 2
dir=${AB_DIR:-/tmp/g16ab}
mkdir -p "$dir" "$(dirname "$out")"
[ -f "$dir/c.zkey" ] || python tools/make_files.py ${AB_LOG2N:-20} "$dir"
g++ -O2 -std=c++17 -Iinclude tools/ab_prove.cpp -ldl -lpthread -o "$dir/ab_prove"
: > "$out"
for rep in $(seq 1 "$alts"); do
  for spec in "$@"; do
    name=${spec%%=*}; path=${spec#*=}; kflag=""; envs=""
    case "$path" in *@*) envs=$(echo "${path#*@}" | tr ',' ' '); path=${path%%@*};; esac
    case "$path" in *:K) path=${path%:K}; kflag="-K";; esac
    env $envs "$dir/ab_prove" -l "$PWD/$path" -z "$dir/c.zkey" -w "$dir/c.wtns" -k ${AB_STEPS:-96} -r 2 $kflag 2>/dev/null \
      | sed "s|^$PWD/$path|$name|" | tee -a "$out"
  done
done
python - "$out" <<'PY' | tee -a "$out"
import sys, statistics as st, collections
v, lat, fp = collections.defaultdict(list), collections.defaultdict(list), set()
for line in open(sys.argv[1]):
    t = line.split()
    if len(t) > 2 and t[1] == "proofs_per_s":
        v[t[0]].append(float(t[2])); fp.add(t[-1])
    elif len(t) > 2 and t[1] == "latency_ms":
        lat[t[0]].append(float(t[2]))
print("== summary (proofs/s over all batches of the session; identical proof bytes from every build: %s)" % (len(fp) == 1))
for k in v:
    print(f"{k}: median {st.median(v[k]):.2f}  min {min(v[k]):.2f}  max {max(v[k]):.2f}  n {len(v[k])}  | single-proof latency median {st.median(lat[k]):.2f} ms")
PY
