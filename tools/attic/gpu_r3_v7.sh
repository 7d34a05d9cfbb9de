#!/bin/bash
# visit 7: deferred split-K reduction — equality test, GEMM / model tests, A/B in the step
OUT=gpurun_out/v7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "deferred or linear or bf16 or training_matches_reference_golden or graphed_step_equals or terabyte_full_batch_matches" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "defer:" "perlayer:DLRM_DEFER_SPLITK=0" "defer_b:" "perlayer_b:DLRM_DEFER_SPLITK=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py $AB > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f  fwd %.3f dgrad %.3f wgrad %.3f" % (d["ms_per_step"], k["linear_fwd"]["ms_per_step"], k["linear_bwd_data"]["ms_per_step"], k["linear_bwd_weight"]["ms_per_step"]))
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
