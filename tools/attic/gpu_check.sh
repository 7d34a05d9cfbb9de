#!/bin/bash
# quick post-change check: a pytest selection (arg 2) and the Kaggle-shape eager step time (host path)
OUT=gpurun_out/${1:-chk}; mkdir -p $OUT
echo "== pytest"; timeout 400 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "$2" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest.log
echo "== kaggle eager"; timeout 100 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/kaggle.json 2> $OUT/kaggle.err; python - <<PY
import json
d=json.load(open("$OUT/kaggle.json")); print("kaggle ms/step %.4f loss %.5f" % (d["ms_per_step"], d["final_loss"]))
PY
