#!/bin/bash
OUT=gpurun_out/r4v7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "lookup_sort or adagrad or sorted or dcn_v2_cross_network_bf16 or graphed_step_survives" > $OUT/pytest_sort.log 2>&1; echo "sort tests rc=$?"; tail -6 $OUT/pytest_sort.log
FLAGS="--no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse"
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_v2_dot.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
DLRM_SORT=rocprim timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 --no-parity-check > $OUT/bench_v2_dot_rocprim.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
timeout 300 python bench.py --steps 20 --warmup 5 $FLAGS --no-parity-check --optimizer rwsadagrad > $OUT/bench_tb_rwsadagrad.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
python - <<PY
import json
for n in ("bench_v2_dot","bench_v2_dot_rocprim","bench_tb_rwsadagrad"):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.3f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()}, (d.get("parity_check") or {}).get("pass"), d["config"].get("lookup_sort","")[:40])
    except Exception as e: print(n, "failed", e)
PY
