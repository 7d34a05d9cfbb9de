#!/bin/bash
OUT=gpurun_out/${1:-r2l}; mkdir -p $OUT
export DLRM_BENCH_WATCHDOG=50
run() { name=$1; shift; timeout 100 python bench.py "$@" --no-cpu-baseline --no-alt-arith > $OUT/$name.json 2> $OUT/$name.err; echo "== $name rc=$?"; grep -v amdgpu.ids $OUT/$name.err | tail -8; python - <<PY
import json
try:
    d=json.load(open("$OUT/$name.json")); print("  value %.0f  ms %.3f  loss %.5f" % (d["value"], d["ms_per_step"], d["final_loss"]), "| alt graph:", d.get("alt_hip_graph"))
    print("  kernels:", {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
except Exception as e: print("  no json", e)
PY
}
run tb_eager_altgraph --steps 20 --warmup 5 --alt-graph
run tb_graph --steps 20 --warmup 5 --graph
run kaggle_eager_altgraph --workload criteo_kaggle --steps 200 --warmup 10 --alt-graph --no-kernel-timers
run tb_adagrad --steps 20 --warmup 5 --optimizer rwsadagrad
echo "== pytest graph"; timeout 120 python -m pytest tests/test_gpu_model.py -m gpu -q -x --timeout=100 -p no:cacheprovider -k "graphed or rwsadagrad" 2>&1 | tail -2
