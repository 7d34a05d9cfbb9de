#!/bin/bash
OUT=gpurun_out/${1:-r2y}; mkdir -p $OUT
echo "== pytest bf16"; timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=200 -p no:cacheprovider -k "bf16_arithmetic" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -12 $OUT/pytest.log
echo "== bench bf16"; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith --mlp-arith bf16 > $OUT/bench_bf16.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench_bf16.json")); print("value %.0f ms %.3f loss %.5f dtype %s" % (d["value"], d["ms_per_step"], d["final_loss"], d["dtype"]))
print({k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
