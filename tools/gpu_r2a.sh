#!/bin/bash
# r2a: dgrad mask-prefetch check (parity + per-layer microbench), Kaggle-shape step (launch-bound) timing
TAG=${1:-r2a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest linear"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=600 -p no:cacheprovider -k "linear" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
echo "== micro gemm f32"; timeout 400 python tools/microbench.py gemm --arith f32 > $OUT/micro_gemm_f32.log 2>&1; grep gemm $OUT/micro_gemm_f32.log
echo "== bench kaggle"; timeout 300 python bench.py --workload criteo_kaggle --steps 100 --warmup 10 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/bench_kaggle.json 2> $OUT/bench_kaggle.err; python - <<PY
import json
d=json.load(open("$OUT/bench_kaggle.json"))
print("kaggle value", d["value"], "ms", d["ms_per_step"])
PY
echo "== bench tb (no timers)"; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith --no-kernel-timers > $OUT/bench_tb_notimers.json 2> $OUT/bench_tb.err; python - <<PY
import json
d=json.load(open("$OUT/bench_tb_notimers.json"))
print("tb value", d["value"], "ms", d["ms_per_step"])
PY
