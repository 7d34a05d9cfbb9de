#!/bin/bash
OUT=gpurun_out/${1:-r2u}; mkdir -p $OUT
echo "== pytest"; timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "single_output or small_reduction or (linear_fwd_bwd and 64-1-256) or full_batch" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.log
echo "== micro last layer"; timeout 100 python - <<PY 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0, "tools"); sys.argv=["x","gemm"]
import microbench as mb
mb.gemm([(65536, 512, 16), (65536, 1, 256)])
PY
echo "== bench"; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("value %.0f ms %.3f loss %.5f" % (d["value"], d["ms_per_step"], d["final_loss"]))
print({k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
