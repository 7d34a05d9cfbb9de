#!/bin/bash
for cfg in 0 192 0; do
  echo "== debug=$cfg"
  DLRM_GEMM_DEBUG=$cfg timeout 200 python tools/microbench.py gemm 2>&1 | grep "^gemm\|Error\|error" | cut -c1-230
done
