#!/bin/bash
# A/B inside ONE box (box-to-box spread is +-3 %): alternate bench runs of env settings "A" and "B" ($2, $3), 2 rounds each
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT
BASE_FLAGS="$BENCH_FLAGS"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-arith --no-parity-check $BENCH_FLAGS > $OUT/$tag.json 2> $OUT/$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$tag.json")); print("%-8s ms %.3f | " % ("$tag", d["ms_per_step"]) + "  ".join("%s %.3f" % (k[:14], v["ms_per_step"]) for k, v in d["kernels"].items() if v["ms_per_step"] > 0.05))
except Exception as e: print("$tag no json", e)
PY
}
# per-variant bench flags: A_FLAGS / B_FLAGS (appended to BENCH_FLAGS)
for r in 1 2; do BENCH_FLAGS="$BASE_FLAGS $A_FLAGS" run A$r $2; BENCH_FLAGS="$BASE_FLAGS $B_FLAGS" run B$r $3; done
