#!/bin/bash
OUT=gpurun_out/r4v16
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bwd_weight_bf16 or gemm_bf16_phased or bf16_storage_tower" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bf16_gemm_bench.txt; cat $OUT/bf16_gemm_bench.txt
export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for dbg in 0 1 4 6 7; do
  echo "== DLRM_BF16_DEBUG=$dbg (1 no DMA, 2 fragments read once, 4 no MFMA)"
  _BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 DLRM_BF16_DEBUG=$dbg BF16_BENCH_SHAPES=2 timeout 200 python tools/bf16_gemm_bench.py 2>&1 | grep -E "^65536" | cut -c1-23,56-90
done | tee $OUT/ablation.txt
