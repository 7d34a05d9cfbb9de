#!/bin/bash
# extra evidence: (1) a step of the 2-stream schedule (negative gaps = kernels running concurrently), (2) kernel stats of config 5
OUT=gpurun_out/${1:-extra}; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/ov -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --overlap --no-cpu-baseline --no-alt-arith --no-parity-check --no-kernel-timers > $GRAFT_REPO_ROOT/$OUT/ov.json 2> $GRAFT_REPO_ROOT/$OUT/ov.err )
python tools/step_trace.py $(find $OUT/ov -name "*kernel_trace.csv" | head -1) 7 > $OUT/step_trace_overlap.txt; tail -1 $OUT/step_trace_overlap.txt
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $GRAFT_REPO_ROOT/$OUT/mh -o mh -- python $GRAFT_REPO_ROOT/bench.py --workload mlperf_v2_multihot --interaction dot --steps 6 --warmup 2 --no-cpu-baseline --no-alt-arith > $GRAFT_REPO_ROOT/$OUT/mh.json 2> $GRAFT_REPO_ROOT/$OUT/mh.err ); echo "mh rc=$?"
find $OUT -name "*kernel_trace.csv" -size +6M -delete
