#!/bin/bash
# one rocprofv3 kernel trace of a short bench run, folded to the per-kernel listing of the last step
OUT=gpurun_out/${1:-trace}; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/rp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt-arith --no-parity-check --no-alt-overlap $BENCH_FLAGS > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/err.log )
f=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py $f | tee $OUT/step.txt
find $OUT/rp -name "*.csv" -size +4M -delete
