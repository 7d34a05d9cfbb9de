# rocprofv3 kernel trace of the graphed Criteo-Kaggle step -> one step's timeline (tools/step_trace.py)
OUT=${1:-gpurun_out/kaggle_trace}; mkdir -p $OUT; ROOT=$PWD; export TMPDIR=/tmp
QUICK="--no-cpu-baseline --no-parity-check --no-box-calibration --no-rccl-selfcheck --no-high-row-check"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/rp -o k -- python $ROOT/bench.py --workload criteo_kaggle --graph --steps 60 --warmup 10 $QUICK --no-kernel-timers ${KAGGLE_ARGS} > $ROOT/$OUT/bench.json 2> $ROOT/$OUT/bench.err )
tr=$(find $OUT/rp -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py "$tr" 40 > $OUT/step_trace_kaggle_graph.txt 2>&1
tail -3 $OUT/step_trace_kaggle_graph.txt
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("kaggle graph ms", d["ms_per_step"])
PY
find $OUT -name "*.csv" -size +4M -delete
