#!/bin/bash
# Round-3 build, evidence refresh at HEAD (kernel sources changed after the first final visit: a tuning knob in gemm.hip): the GEMM tests,
# the default bench line, rocprofv3 kernel statistics and the two PMC passes of the same command.
OUT=gpurun_out/final2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (GEMM + model level)"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q --timeout=600 -p no:cacheprovider -k "linear or relu or terabyte_full_batch_matches or training_matches_reference_golden" > $OUT/pytest_gemm.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gemm.log
echo "== bench (default)";  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value %.0f ms %.3f parity %s" % (d["value"], d["ms_per_step"], (d.get("parity_check") or {}).get("pass")))
for k,v in d["kernels"].items(): print("  %-18s %.3f ms  frac %s" % (k, v["ms_per_step"], v.get("frac")))
c=d.get("cpu_baseline") or {}; print("cpu", c.get("value"), c.get("ms_per_step"), c.get("threads"), c.get("iterations_run"))
s=d.get("stock_gpu_baseline") or {}; print("stock", s.get("value"), s.get("ms_per_step"), s.get("error"))
print("roofline", d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_note"])
PY
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $FLAGS > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
tr=$(find $OUT/rocprof -name "*kernel_trace.csv" | head -1); [ -n "$tr" ] && python tools/step_trace.py "$tr" 8 > $OUT/step_trace.txt 2>&1; tail -1 $OUT/step_trace.txt
db=$(find $OUT/rocprof -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py "$db" --out $OUT/rocprof_kernel_stats.md; python tools/rocpd_summary.py "$db" --by-grid --out $OUT/rocprof_kernel_stats_by_grid.md; fi
st=$(find $OUT/rocprof -name "*kernel_stats.csv" | head -1); [ -n "$st" ] && cp "$st" $OUT/rocprof_kernel_stats.csv
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete; find $OUT -name "*.db" -size +8M -delete
