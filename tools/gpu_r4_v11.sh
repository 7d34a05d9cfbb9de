#!/bin/bash
OUT=gpurun_out/r4v11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "bwd_weight_bf16 or gemm_bf16_phased or bf16_storage_tower or test_linear or relu_sign or training_matches_reference_golden or graphed_step_equals" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest.log
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bf16_gemm_bench.txt; cat $OUT/bf16_gemm_bench.txt
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration"
timeout 200 python bench.py --workload criteo_kaggle --steps 200 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph.json 2>/dev/null
DLRM_GEMM_SMALL=0 timeout 200 python bench.py --workload criteo_kaggle --steps 200 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph_small0.json 2>/dev/null
timeout 200 python bench.py --workload criteo_kaggle --steps 200 --warmup 20 $FLAGS > $OUT/bench_kaggle_eager.json 2>/dev/null
python - <<PY
import json
for n in ("bench_kaggle_graph","bench_kaggle_graph_small0","bench_kaggle_eager"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-26s ms %.4f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
    except Exception as e: print(n, "failed", e)
PY
