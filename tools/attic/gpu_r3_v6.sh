#!/bin/bash
# visit 6: bf16-storage tower — bit-identity with the in-loop rounding path, casts, v2 parity in bf16, step time of the mlperf_v2 workload
OUT=gpurun_out/v6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=400 -p no:cacheprovider -k "bf16 and not x6" > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_bf16.log
for st in 1 0; do
  DLRM_BF16_STORAGE=$st timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 --no-parity-check > $OUT/v2_dot_storage$st.json 2> $OUT/v2_dot_storage$st.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/v2_dot_storage$st.json")); k=d["kernels"]; print("storage=$st v2 dot ms %.3f" % d["ms_per_step"], {n: round(v["ms_per_step"],3) for n,v in k.items()}, "frac fwd", round(k["linear_fwd"]["frac"],3))
except Exception as e: print("storage=$st failed", e); print(open("$OUT/v2_dot_storage$st.err").read()[-1200:])
PY
done
DLRM_BF16_STORAGE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap --mlp-arith bf16 > $OUT/tb_bf16.json 2> $OUT/tb_bf16.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/tb_bf16.json")); k=d["kernels"]; print("TB bf16 storage ms %.3f" % d["ms_per_step"], {n: round(v["ms_per_step"],3) for n,v in k.items() if n.startswith("linear")})
except Exception as e: print("tb bf16 failed", e); print(open("$OUT/tb_bf16.err").read()[-1200:])
PY
