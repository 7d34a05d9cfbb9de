#!/bin/bash
# Round 4, visit 3: phased bf16 GEMM (test + micro-benchmark), HBM copy sweep, tr-read semantics, then the whole GPU suite
OUT=gpurun_out/r4v3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm_bf16_phased or bf16" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -3 $OUT/pytest_bf16.log
timeout 600 python tools/bf16_gemm_bench.py > $OUT/bf16_gemm_bench.txt 2>&1; cat $OUT/bf16_gemm_bench.txt | grep -v amdgpu.ids
hipcc --offload-arch=gfx950 -O3 -w tools/probes/hbm_copy_sweep.hip -o /tmp/hbm_copy_sweep && timeout 120 /tmp/hbm_copy_sweep > $OUT/hbm_copy_sweep.txt 2>&1; sort -k7 -n -r $OUT/hbm_copy_sweep.txt | head -8
hipcc --offload-arch=gfx950 -O3 -w tools/probes/tr_read_probe.hip -o /tmp/tr_read_probe && timeout 60 /tmp/tr_read_probe > $OUT/tr_read_probe.txt 2>&1; head -20 $OUT/tr_read_probe.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -8 $OUT/pytest_gpu.log
