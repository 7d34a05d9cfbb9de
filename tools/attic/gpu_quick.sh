#!/bin/bash
# quick GEMM check: a few parity cases + per-layer microbench for one arithmetic
AR=${1:-bf16x6}
echo "== pytest"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "linear and (512-256-64 or 4096-200 or 66000)" 2>&1 | tail -3
echo "== micro gemm $AR"; timeout 600 python tools/microbench.py gemm --arith $AR 2>&1 | grep gemm
