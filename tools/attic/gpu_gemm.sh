#!/bin/bash
# GEMM-focused GPU visit: parity tests of the MLP kernels, per-layer microbench, end-to-end bench (both arithmetics).
TAG=${1:-g01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest (linear + model)"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "linear or training" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
for ar in f32 bf16x6; do
echo "== micro gemm $ar"; timeout 600 python tools/microbench.py gemm --arith $ar > $OUT/micro_gemm_$ar.log 2>&1; cat $OUT/micro_gemm_$ar.log
echo "== bench $ar"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --mlp-arith $ar > $OUT/bench_$ar.json 2> $OUT/bench_$ar.err; python - <<PY
import json
d=json.load(open("$OUT/bench_$ar.json"))
print("value", d["value"], "ms", d["ms_per_step"], "loss", d["final_loss"])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
PY
done
