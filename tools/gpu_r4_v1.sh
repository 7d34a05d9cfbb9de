#!/bin/bash
# Round 4, visit 1: the new parity cases (config 4 at its own shapes, one-lookup-per-bag proof, ADVICE fixes), bench.py --gpus 2
# self-launch on one GPU (gloo self-test), a short default bench line.
OUT=gpurun_out/r4v1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "all_to_all_blocks or eight_rank_terabyte or ragged_batch or coo_escape or side_stream or gather_interaction or graphed_step_equals or test_terabyte_full_batch_matches" > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_new.log
echo "== bench --gpus 2 without a launcher (gloo self-test on one GPU)"
DLRM_BENCH_SELFTEST_GLOO=1 timeout 500 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 8192 --row-cap 200000 --hang-timeout 120 > $OUT/selflaunch_n2.json 2> $OUT/selflaunch_n2.err
echo "rc=$?"; grep -o '"n_gpus": [0-9]*' $OUT/selflaunch_n2.json; grep "re-executing" $OUT/selflaunch_n2.err | cut -c1-200
echo "== bench default"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
PY
