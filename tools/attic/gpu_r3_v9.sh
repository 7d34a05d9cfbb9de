#!/bin/bash
# visit 9: the small layers of the towers, isolated: default tiling, forced tilings, without the epilogue
for cfg in "default:" "noepi:DLRM_GEMM_DEBUG=4" "tm2:DLRM_GEMM_TM=2" "tm4:DLRM_GEMM_TM=4"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  echo "== $tag ($envs)"
  env $envs timeout 300 python tools/microbench.py gemm 2>&1 | grep "^gemm" | cut -c1-210
done
