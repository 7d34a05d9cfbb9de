#!/bin/bash
# r2c: whole-step HIP graph (parity + Kaggle/TB timing), on-device inference metrics
TAG=${1:-r2c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest graph+inference"; timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "graphed or inference" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log
echo "== bench kaggle eager+alt graph"; timeout 300 python bench.py --workload criteo_kaggle --steps 200 --warmup 10 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/bench_kaggle.json 2> $OUT/bench_kaggle.err; tail -3 $OUT/bench_kaggle.err; python - <<PY
import json
d=json.load(open("$OUT/bench_kaggle.json"))
print("kaggle eager", d["value"], "ms", d["ms_per_step"], "| graph", d.get("alt_hip_graph"))
PY
echo "== bench tb eager+alt graph"; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith > $OUT/bench_tb.json 2> $OUT/bench_tb.err; tail -3 $OUT/bench_tb.err; python - <<PY
import json
d=json.load(open("$OUT/bench_tb.json"))
print("tb eager", d["value"], "ms", d["ms_per_step"], "| graph", d.get("alt_hip_graph"))
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
PY
