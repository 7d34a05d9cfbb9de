#!/bin/bash
# visit 15: single stream vs the 2-stream schedule with the fused forward (only the backward overlaps now), alternating, one box
OUT=gpurun_out/v15; mkdir -p $OUT
AB="--steps 30 --warmup 5 --no-cpu-baseline --no-parity-check --no-alt-arith --no-alt-overlap --no-alt-fuse"
for cfg in "single:" "overlap:--overlap" "single_b:" "overlap_b:--overlap" "single_c:" "overlap_c:--overlap"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  timeout 300 python bench.py $AB $extra > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ab_$tag.json")); k=d["kernels"]
    print("$tag ms %.3f " % d["ms_per_step"] + " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in k if n.startswith(("emb", "linear"))), "| roofline frac %.3f" % d["roofline"]["frac"])
except Exception as e: print("$tag failed", e); print(open("$OUT/ab_$tag.err").read()[-800:])
PY
done
