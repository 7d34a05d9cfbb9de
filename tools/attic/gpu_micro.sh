#!/bin/bash
TAG=${1:-m01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest (emb + model)"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "emb or training or full_batch" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
echo "== micro all"; timeout 600 python tools/microbench.py all > $OUT/micro_all.log 2>&1; cat $OUT/micro_all.log
for f in 1 3 4 7; do echo "== gemm_big DLRM_GEMM_DEBUG=$f"; DLRM_GEMM_DEBUG=$f timeout 300 python tools/microbench.py gemm_big > $OUT/micro_dbg$f.log 2>&1; cat $OUT/micro_dbg$f.log; done
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"])
for k,v in d["kernels"].items(): print(k, round(v["ms_per_step"],3), v.get("achieved"))
PY
