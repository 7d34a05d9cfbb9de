#!/bin/bash
# visit 11: lookup sort issued during the forward pass on the side stream (DLRM_PRESORT=1) — parity, then A/B in the step
OUT=gpurun_out/v11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -q --timeout=400 -p no:cacheprovider -k "presorted or lookup_sort or emb_bwd_sgd_sorted" 2>&1 | tail -3
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "presort:DLRM_PRESORT=1" "inline:" "presort_b:DLRM_PRESORT=1" "inline_b:"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py $AB > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f " % d["ms_per_step"], {n: round(v["ms_per_step"],3) for n,v in k.items()})
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
