#!/bin/bash
# visit 14: 256 x 256 tile of the bf16-storage GEMM (parity, then A/B on the two bf16 workloads)
OUT=gpurun_out/v14; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "bf16" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -3 $OUT/pytest_bf16.log
AB="--steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap --no-alt-fuse --mlp-arith bf16"
for cfg in "sq:" "fp32shape:DLRM_BF16_TN=2" "sq_b:" "fp32shape_b:DLRM_BF16_TN=2"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py $AB > $OUT/tb_bf16_$tag.json 2> $OUT/tb_bf16_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/tb_bf16_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f " % d["ms_per_step"] + " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith("linear")))
except Exception as e: print("$tag failed", e); print(open("$OUT/tb_bf16_$tag.err").read()[-800:])
PY
done
