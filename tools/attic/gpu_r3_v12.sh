#!/bin/bash
# visit 12: the fused lookup + interaction kernels with LDS-DMA'd row selectors (parity, then A/B inside the training step)
OUT=gpurun_out/v12; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gather_interaction or interact" > $OUT/pytest_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/pytest_kernels.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "terabyte_full_batch" > $OUT/pytest_model.log 2>&1; echo "model rc=$?"; tail -3 $OUT/pytest_model.log
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "two:" "fused:--fuse" "two_b:" "fused_b:--fuse"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  timeout 300 python bench.py $AB $extra > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f " % d["ms_per_step"] + " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith(("emb", "interact"))))
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
