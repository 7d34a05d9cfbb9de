#!/bin/bash
OUT=gpurun_out/r4v6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "bf16 or mlperf_v2 or torchrec or dcn" > $OUT/pytest_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -12 $OUT/pytest_bf16.log
timeout 400 python bench.py --workload mlperf_v2_multihot --interaction dcn --steps 10 --warmup 3 > $OUT/bench_v2_dcn.json 2> $OUT/err.txt || tail -5 $OUT/err.txt
_BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 timeout 600 python tools/bf16_gemm_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-23,100-200 > $OUT/bf16_gemm_bench.txt; cat $OUT/bf16_gemm_bench.txt
python - <<PY
import json
for n in ("bench_v2_dcn",):
    try:
        d=json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print("%-22s ms %.3f" % (n, d["ms_per_step"]), {k: round(v["ms_per_step"],3) for k,v in d["kernels"].items()}, (d.get("parity_check") or {}).get("pass"))
    except Exception as e: print(n, "failed", e)
PY
