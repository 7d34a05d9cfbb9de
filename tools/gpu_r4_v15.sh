#!/bin/bash
# ablation of the phased bf16 GEMM main loop (TUNING build, timing-only switches: results are wrong by design)
OUT=gpurun_out/r4v15
mkdir -p $OUT
export TMPDIR=/tmp
export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for dbg in 0 1 2 4 3 5 6 7; do
  echo "== DLRM_BF16_DEBUG=$dbg (1 no DMA, 2 fragments read once, 4 no MFMA)"
  _BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 DLRM_BF16_DEBUG=$dbg BF16_BENCH_SHAPES=2 timeout 200 python tools/bf16_gemm_bench.py 2>&1 | grep -E "^65536" | cut -c1-23,56-90
done | tee $OUT/ablation.txt
