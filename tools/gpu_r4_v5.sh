#!/bin/bash
OUT=gpurun_out/r4v5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "bf16 or mlperf_v2 or torchrec" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -15 $OUT/pytest_bf16.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check"
for lean in 1 0; do
DLRM_BF16_LEAN=$lean timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --mlp-arith bf16 > $OUT/bench_tb_bf16_lean$lean.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
done
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_v2_dot.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 > $OUT/bench_v2_dcn.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
python - <<PY
import json
for n in ("bench_tb_bf16_lean1","bench_tb_bf16_lean0","bench_v2_dot","bench_v2_dcn"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.3f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()}, (d.get("parity_check") or {}).get("pass"))
    except Exception as e: print(n, "failed", e)
PY
