#!/bin/bash
# Builds dlrm_amd/libdlrm_hip_tuning.so = the library with -DDLRM_TUNING (the timing-only work-skipping switches DLRM_GEMM_DEBUG /
# DLRM_BF16_DEBUG compiled in; results are WRONG by design when a switch is set).  Objects go to a scratch directory: the product
# build's objects are not touched.  Use:  DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so DLRM_BF16_DEBUG=8 python tools/bf16_gemm_bench.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${TMPDIR:-/tmp}/dlrm_tuning_objs
mkdir -p "$OBJ"
FLAGS="-O3 -std=c++17 -fPIC -munsafe-fp-atomics --offload-arch=gfx950 -Wno-unused-result -DDLRM_TUNING -I$ROOT/dlrm_amd/csrc -I$ROOT/include"
for s in emb emb_sorted interact gemm loss_opt adagrad metrics datagen gemv smallk multihot calib gemm_bf16 tower; do
  if [ ! -f "$OBJ/$s.o" ] || [ "$ROOT/dlrm_amd/csrc/$s.hip" -nt "$OBJ/$s.o" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$ROOT/dlrm_amd/csrc/$s.hip" -o "$OBJ/$s.o" &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OBJ"/*.o -o "$ROOT/dlrm_amd/libdlrm_hip_tuning.so"
echo "built $ROOT/dlrm_amd/libdlrm_hip_tuning.so"
