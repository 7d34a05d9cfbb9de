#!/bin/bash
# visit 10: bf16-storage GEMM tile height (256- vs 128-row tiles), in the TB step with --mlp-arith bf16
AB="--steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap --mlp-arith bf16"
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -p no:cacheprovider -k "bf16_storage" 2>&1 | tail -2
for cfg in "tm4:" "tm2:DLRM_BF16_TM=2" "tm4_b:" "tm2_b:DLRM_BF16_TM=2"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py $AB 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$tag ms %.3f' % d['ms_per_step'], {n: round(v['ms_per_step'],3) for n,v in k.items() if n.startswith('linear')})"
done
