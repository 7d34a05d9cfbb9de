#!/bin/bash
OUT=gpurun_out/r4v12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "bwd_weight_bf16 or gemm_bf16_phased or bf16_storage_tower or mlperf_v2_bench or dcn_v2_cross_network_bf16" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bf16_gemm_bench.txt; cut -c1-23,56-90 $OUT/bf16_gemm_bench.txt
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check"
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16 > $OUT/bench_tb_bf16.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
DLRM_BF16_WIDE_STORE=0 timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16 > $OUT/bench_tb_bf16_narrow.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
python - <<PY
import json
for n in ("bench_tb_bf16","bench_tb_bf16_narrow"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.3f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
    except Exception as e: print(n, "failed", e)
PY
