#!/bin/bash
OUT=gpurun_out/r4v14
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "graphed or training_matches_reference_golden or rwsadagrad or coo_escape or side_stream or kernel_timers" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration"
for ov in 8192 0; do
DLRM_SMALL_BATCH_OVERLAP=$ov timeout 200 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS --graph > $OUT/bench_kaggle_graph_ov$ov.json 2>/dev/null
DLRM_SMALL_BATCH_OVERLAP=$ov timeout 200 python bench.py --workload criteo_kaggle --steps 300 --warmup 20 $FLAGS > $OUT/bench_kaggle_eager_ov$ov.json 2>/dev/null
done
python - <<PY
import json
for n in ("bench_kaggle_graph_ov8192","bench_kaggle_graph_ov0","bench_kaggle_eager_ov8192","bench_kaggle_eager_ov0"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-28s ms %.4f loss %.6f" % (n, d["ms_per_step"], d["final_loss"]))
    except Exception as e: print(n, "failed", e)
PY
