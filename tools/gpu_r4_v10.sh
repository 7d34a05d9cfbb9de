#!/bin/bash
OUT=gpurun_out/r4v10
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bwd_weight_bf16 or gemm_bf16_phased or bf16_storage_tower" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -5 $OUT/pytest_bf16.log
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bf16_gemm_bench.txt; cat $OUT/bf16_gemm_bench.txt
