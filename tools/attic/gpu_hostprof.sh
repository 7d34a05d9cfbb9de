#!/bin/bash
# host-path profile of the eager training step at Criteo-Kaggle shapes (launch-bound: the step time IS the host time)
OUT=gpurun_out/${1:-hostprof}; mkdir -p $OUT
timeout 200 python -m cProfile -o $OUT/prof.out bench.py --workload criteo_kaggle --steps 400 --warmup 20 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import pstats, json
d=json.load(open("$OUT/bench.json")); print("ms/step", d["ms_per_step"])
p=pstats.Stats("$OUT/prof.out"); p.sort_stats("tottime").print_stats(45)
PY
