#!/bin/bash
OUT=gpurun_out/r4v4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bwd_weight_bf16 or gemm_bf16_phased" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -15 $OUT/pytest_bf16.log
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py > $OUT/bf16_gemm_bench.txt 2>&1; cat $OUT/bf16_gemm_bench.txt | grep -v amdgpu.ids
