#!/bin/bash
# Builds dlrm_amd/libdlrm_hip_<tag>.so = the product library with extra -D flags on the listed sources (A/B of a compile-time constant
# inside ONE GPU visit; load it with DLRM_HIP_LIB).  usage: tools/build_variant_lib.sh <tag> "<-D flags>" <source> [<source> ...]
#   e.g.  tools/build_variant_lib.sh seg1024 "-DDLRM_SEG_TILE=1024" emb_sorted adagrad
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; DEFS=$2; shift 2
OBJ=${TMPDIR:-/tmp}/dlrm_variant_$TAG
mkdir -p "$OBJ"
make -C "$ROOT/dlrm_amd/csrc" -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-result $DEFS -I$ROOT/dlrm_amd/csrc -I$ROOT/include"
objs=""
for s in emb emb_sorted interact gemm loss_opt adagrad metrics datagen gemv smallk multihot calib gemm_bf16 tower; do
  if [[ " $* " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$ROOT/dlrm_amd/csrc/$s.hip" -o "$OBJ/$s.o" &
    objs="$objs $OBJ/$s.o"
  else
    objs="$objs $ROOT/dlrm_amd/csrc/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$ROOT/dlrm_amd/libdlrm_hip_$TAG.so"
echo "built $ROOT/dlrm_amd/libdlrm_hip_$TAG.so"
