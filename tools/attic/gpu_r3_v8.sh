#!/bin/bash
# visit 8: occupancy / epilogue probe of the forward GEMM (128-row tiles): 3 vs 2 workgroups per CU, with and without the epilogue
for cfg in "3wg:" "2wg:DLRM_GEMM_LDS_PAD=20000" "3wg_noepi:DLRM_GEMM_DEBUG=4" "2wg_noepi:DLRM_GEMM_DEBUG=4 DLRM_GEMM_LDS_PAD=20000" "tm4:DLRM_GEMM_TM=4" "tm4_noepi:DLRM_GEMM_TM=4 DLRM_GEMM_DEBUG=4"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  echo "== $tag ($envs)"
  env $envs timeout 200 python tools/microbench.py gemm_big 2>&1 | grep "^gemm" | cut -c1-200
done
