#!/bin/bash
OUT=gpurun_out/r4v17
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "test_linear or relu_sign or training_matches_reference_golden or graphed_step_equals or weight_gradient_of_padded or dcn_v2_cross_network_matches or smoke" --deselect "tests/test_gpu_kernels.py::test_linear_fwd_bwd[f32-65536-512-256-1]" --deselect "tests/test_gpu_kernels.py::test_linear_fwd_bwd[bf16x6-65536-512-256-1]" --deselect "tests/test_gpu_kernels.py::test_linear_fwd_bwd[f32-66000-384-272-1]" --deselect "tests/test_gpu_kernels.py::test_linear_fwd_bwd[bf16x6-66000-384-272-1]" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration"
for sm in 2 1; do
DLRM_GEMM_SMALL=$sm timeout 200 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph_small$sm.json 2>/dev/null
done
timeout 200 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS > $OUT/bench_kaggle_eager.json 2>/dev/null
python - <<PY
import json
for n in ("bench_kaggle_graph_small2","bench_kaggle_graph_small1","bench_kaggle_eager"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-28s ms %.4f loss %.6f calls %s" % (n, d["ms_per_step"], d["final_loss"], (d.get("launches") or {}).get("c_abi_calls_per_step")))
    except Exception as e: print(n, "failed", e)
PY
