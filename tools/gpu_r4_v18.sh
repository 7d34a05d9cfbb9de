#!/bin/bash
# round 4, visit 18: full-grid elementwise cross kernels + 8-wide bf16 cast — tests of the touched paths, then the two workloads they serve
OUT=gpurun_out/r4v18
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "casts or dcn or cross or bf16_storage_tower or mlperf_v2_bench" > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-box-calibration"
timeout 300 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 $FLAGS > $OUT/bench_mlperf_v2_dcn.json 2>$OUT/err_dcn.txt
timeout 300 python bench.py --mlp-arith bf16 --steps 20 --warmup 5 $FLAGS > $OUT/bench_tb_bf16.json 2>$OUT/err_tb.txt
python - <<PY
import json
for n in ("bench_mlperf_v2_dcn","bench_tb_bf16"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-24s ms %.3f parity %s" % (n, d["ms_per_step"], (d.get("parity_check") or {}).get("pass")), {k: v[0] for k, v in d["roofline"]["by_category"].items()})
    except Exception as e: print(n, "failed", e)
PY
