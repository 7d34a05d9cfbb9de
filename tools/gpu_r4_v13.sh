#!/bin/bash
# A/B inside one box: default fp32 step vs weight gradients on a side stream vs 256-row tiles wherever possible; graphed Kaggle step
OUT=gpurun_out/r4v13
mkdir -p $OUT
export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration --steps 30 --warmup 6"
for tag in base wgrad_side tm4 base2; do
  case $tag in
    base|base2) envs="";;
    wgrad_side) envs="DLRM_OVERLAP_WGRAD=1";;
    tm4) envs="DLRM_GEMM_TM=4";;
  esac
  env $envs timeout 300 python bench.py $FLAGS > $OUT/bench_$tag.json 2> $OUT/err_$tag.txt || tail -3 $OUT/err_$tag.txt
done
timeout 200 python bench.py --workload criteo_kaggle --steps 200 --warmup 20 --no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check --no-box-calibration --graph > $OUT/bench_kaggle_graph.json 2>/dev/null
python - <<PY
import json
for n in ("bench_base","bench_wgrad_side","bench_tm4","bench_base2","bench_kaggle_graph"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.4f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
    except Exception as e: print(n, "failed", e)
PY
