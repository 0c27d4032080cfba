#!/bin/bash
# Build a variant of libg16hip.so out of tree (objects of the tree are reused where the flags do not touch them):
#   bash tools/build_variant.sh NAME "make variables" obj1.o obj2.o ...   -> nim_groth16_amd/csrc/build_variants/libg16hip_NAME.so
# e.g. bash tools/build_variant.sh t4 'TAIL_FLAGS="-DG16_TAIL_WAVES_G1=4 -DG16_TAIL_WAVES_G2=2"' msm_g1_reduce1.o msm_g1_reduce2.o
set -e
name=$1; vars=$2; shift 2
src=$(cd "$(dirname "$0")/../nim_groth16_amd/csrc" && pwd)
dst=/tmp/bv_$name/nim_groth16_amd/csrc
rm -rf /tmp/bv_$name && mkdir -p $dst /tmp/bv_$name/include
cp $src/*.cuh $src/*.hip $src/*.inc $src/*.hpp $src/*.o $src/Makefile $dst/
cp $src/../../include/*.h /tmp/bv_$name/include/
(cd $dst && rm -f libg16hip.so "$@" && touch -d '2000-01-01' *.cuh *.hip *.inc *.hpp ../../include/*.h Makefile && eval make -j8 $vars 2>&1 | grep -E "error|Error|warning: .*spill" || true)
mkdir -p $src/build_variants && cp $dst/libg16hip.so $src/build_variants/libg16hip_$name.so
python $(dirname "$0")/kernel_resources.py $(for o in "$@"; do echo $dst/$o; done) | grep -E " (msm_heavy|msm_reduce|msm_fold|msm_accum|ntt_|part_pass)" || true
