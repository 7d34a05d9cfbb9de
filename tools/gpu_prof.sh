#!/bin/bash
# profiling evidence only: rocprofv3 kernel stats + the two PMC passes of the single-stream headline steps
OUT=gpurun_out/${1:-prof}; mkdir -p $OUT; export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/rocprof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 $FLAGS > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err ); echo "rocprof rc=$?"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 $FLAGS > $GRAFT_REPO_ROOT/$OUT/$c.log 2>&1
  echo "rc=$? $c"
done
cd $GRAFT_REPO_ROOT
python tools/pmc_fold.py $OUT
python tools/step_trace.py $OUT/rocprof/bench_kernel_trace.csv 8 > $OUT/step_trace.txt; tail -1 $OUT/step_trace.txt
find $OUT -name "p_counter_collection.csv" -size +8M -delete; find $OUT -name "*kernel_trace.csv" -size +8M -delete
python - <<PY
import json
d=json.load(open("$OUT/rocprof_bench.json")); print("under rocprof: ms %.3f" % d["ms_per_step"], {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
