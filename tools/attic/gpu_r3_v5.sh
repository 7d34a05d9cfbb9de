#!/bin/bash
# visit 5: the tuned segmented sort inside the training step (A/B against rocPRIM), graph replay, Kaggle shapes
OUT=gpurun_out/v5; mkdir -p $OUT
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap"
for cfg in "own:" "rocprim:DLRM_SORT=rocprim" "own_b:" "rocprim_b:DLRM_SORT=rocprim" "own_graph:" "own_adagrad:" "rocprim_adagrad:DLRM_SORT=rocprim"; do
  tag=${cfg%%:*}; envs=${cfg#*:}; extra=""
  case $tag in *graph) extra="--graph";; *adagrad) extra="--optimizer rwsadagrad";; esac
  env $envs timeout 300 python bench.py $AB $extra > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    e = k.get("emb_bwd_sgd") or k.get("emb_bwd_adagrad") or {}
    print("$tag ms %.3f  emb_bwd %s frac %s  update=%s" % (d["ms_per_step"], e.get("ms_per_step"), e.get("frac"), d["config"]["embedding_update"][:24]))
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
for g in "" "--graph"; do
  timeout 300 python bench.py --workload criteo_kaggle --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap $g > $OUT/kaggle$g.json 2> $OUT/kaggle$g.err
  python -c "
import json
try:
    d=json.load(open('$OUT/kaggle$g.json')); print('kaggle $g ms %.3f update=%s' % (d['ms_per_step'], d['config']['embedding_update'][:40]))
except Exception as e: print('kaggle $g failed', e)"
done
