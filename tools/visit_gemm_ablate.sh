#!/bin/bash
# timing-only ablation of the fp32 GEMM (tuning build; results are WRONG by design): DLRM_GEMM_DEBUG bits 1 = no DMA refill in the k-loop,
# 2 = no wait + barrier, 4 = no epilogue.  usage: tools/visit_gemm_ablate.sh <tag> [ENV=val ...]
OUT=gpurun_out/${1:-ablate}; shift; mkdir -p $OUT
export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for d in 0 4 1 3 7; do
  env "$@" DLRM_GEMM_DEBUG=$d python tools/gemm_forms_bench.py > $OUT/debug_$d.log 2>&1
  echo "== DEBUG=$d $*"; grep -E "top1024->1024|bot512->256|TOTAL" $OUT/debug_$d.log | grep -v "tuning switch"
done
