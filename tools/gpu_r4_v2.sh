#!/bin/bash
# Round 4, visit 2: the new parity cases again (visit 1 stopped at the first failure), a bench line with the box calibration
OUT=gpurun_out/r4v2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -k "all_to_all_blocks or eight_rank_terabyte or ragged_batch or coo_escape or side_stream or gather_interaction or graphed_step_equals or test_terabyte_full_batch_matches" > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_new.log
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
print("box", d["box"])
print(d["roofline"]["by_category"])
PY
