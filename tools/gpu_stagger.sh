#!/bin/bash
for cfg in 0 8 16 0 8 16; do
  echo "== debug=$cfg"
  DLRM_GEMM_DEBUG=$cfg timeout 200 python tools/microbench.py gemm 2>&1 | grep "^gemm" | cut -c1-200
done
