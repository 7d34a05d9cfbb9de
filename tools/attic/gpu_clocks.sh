#!/bin/bash
# sample sclk / power with rocm-smi while bench.py runs its timed steps
OUT=gpurun_out/${1:-clocks}; mkdir -p $OUT
( while true; do date +%s.%N; rocm-smi -c -P --csv 2>/dev/null | grep card0; sleep 0.2; done ) > $OUT/smi.log 2>&1 &
SMI=$!
timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap --no-kernel-timers > $OUT/bench.json 2> $OUT/err.log
kill $SMI
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("ms %.3f" % d["ms_per_step"])
PY
grep card0 $OUT/smi.log | awk -F, '{print $6, $10}' | sort | uniq -c | sort -rn | head -12
