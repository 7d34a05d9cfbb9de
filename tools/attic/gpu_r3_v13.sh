#!/bin/bash
# visit 13: fused lookup + interaction as the default — whole GPU suite, then the headline bench (with its alt lines)
OUT=gpurun_out/v13; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -x -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); k=d["kernels"]
print("ms %.3f value %.0f parity %s" % (d["ms_per_step"], d["value"], d.get("parity_check", {}).get("pass")))
print(" ".join("%s %.3f/%.2f" % (n, k[n]["ms_per_step"], k[n].get("frac") or 0) for n in k))
for a in ("alt_two_kernel_lookup", "alt_mlp_arith", "alt_stream_overlap"):
    print(a, d.get(a, {}).get("ms_per_step"))
print(d["roofline"])
PY
