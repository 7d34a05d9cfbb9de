#!/bin/bash
# Round-3 build, last visit: the whole GPU suite at HEAD and the config-5 bench lines (kernel sources unchanged since gpu_r3_final3.sh)
OUT=gpurun_out/final4; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -16 $OUT/pytest_gpu.log
timeout 600 python bench.py --workload mlperf_v2_multihot --interaction dot --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dot.json 2> $OUT/bench_mlperf_v2_dot.err
timeout 600 python bench.py --workload mlperf_v2_multihot --steps 10 --warmup 3 > $OUT/bench_mlperf_v2_dcn.json 2> $OUT/bench_mlperf_v2_dcn.err
python - <<PY
import json
for n in ("bench_mlperf_v2_dot","bench_mlperf_v2_dcn"):
    try:
        d=json.load(open("$OUT/%s.json" % n)); p=d.get("parity_check") or {}
        print("%-24s ms %.3f parity=%s" % (n, d["ms_per_step"], p.get("pass")), {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items() if k.startswith(("emb", "linear"))})
    except Exception as e: print(n, "failed", e)
PY
