#!/bin/bash
# r2d: datagen + graph (fixed) parity, then graph timing in SEPARATE processes
TAG=${1:-r2d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest datagen+graph+inference"; timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "datagen or graphed or inference" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log
echo "== bench kaggle eager (+alt graph)"; timeout 300 python bench.py --workload criteo_kaggle --steps 200 --warmup 10 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/bench_kaggle.json 2> $OUT/bench_kaggle.err; echo "rc=$?"; tail -2 $OUT/bench_kaggle.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_kaggle.json")); print("kaggle eager", d["value"], "ms", d["ms_per_step"], "| graph", d.get("alt_hip_graph"))
except Exception as e: print("no json", e)
PY
echo "== bench kaggle --graph"; timeout 300 python bench.py --workload criteo_kaggle --graph --steps 200 --warmup 10 --no-cpu-baseline --no-alt-arith > $OUT/bench_kaggle_graph.json 2> $OUT/bench_kaggle_graph.err; echo "rc=$?"; tail -2 $OUT/bench_kaggle_graph.err; cat $OUT/bench_kaggle_graph.json | cut -c1-300
echo "== bench tb eager (+alt graph)"; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith > $OUT/bench_tb.json 2> $OUT/bench_tb.err; echo "rc=$?"; tail -2 $OUT/bench_tb.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_tb.json")); print("tb eager", d["value"], "ms", d["ms_per_step"], "| graph", d.get("alt_hip_graph"))
    for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
except Exception as e: print("no json", e)
PY
