#!/bin/bash
# GEMM-focused GPU visit: parity tests of the MLP kernels, per-layer microbench, end-to-end bench.
TAG=${1:-g01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest (linear + model)"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "linear or training" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
echo "== micro gemm"; timeout 600 python tools/microbench.py gemm > $OUT/micro_gemm.log 2>&1; cat $OUT/micro_gemm.log
if [ -n "$2" ]; then echo "== micro gemm (fallback kernel)"; DLRM_GEMM_PATH=2 timeout 600 python tools/microbench.py gemm_big > $OUT/micro_gemm_v2.log 2>&1; cat $OUT/micro_gemm_v2.log; fi
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
PY
echo "== bench (no wgrad overlap)"; DLRM_OVERLAP_WGRAD=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_noov.json 2> $OUT/bench_noov.err; python - <<PY
import json
d=json.load(open("$OUT/bench_noov.json"))
print("value", d["value"], "ms", d["ms_per_step"])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
PY
