#!/bin/bash
# second visit of round 3: MFMA ceiling probe, the tests touched since r03a, bench with / without the stream overlap
OUT=gpurun_out/${1:-r03b}; mkdir -p $OUT
echo "== mfma peak"; timeout 120 tools/probes/mfma_peak > $OUT/mfma_peak.log 2>&1; cat $OUT/mfma_peak.log
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider \
  -k "${2:-single_output or small_reduction or terabyte or coo or multihot or out_of_range or copy_blocks or multi_rank or cat_wbce or emb_fwd_bit}" > $OUT/pytest.log 2>&1
echo "rc=$?"; grep -E "^E  |passed|failed|^FAILED" $OUT/pytest.log | cut -c1-250 | head -60
for tag in overlap no-overlap; do
  flag=""; [ $tag = no-overlap ] && flag="--no-overlap"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-arith $flag > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err; echo "== bench $tag rc=$?"
  grep -v amdgpu.ids $OUT/bench_$tag.err | tail -3
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$tag.json"))
    print("value %.0f ms %.3f loss %.6f parity %s" % (d["value"], d["ms_per_step"], d["final_loss"], {k: d["parity_check"].get(k) for k in ("pass","rel_err","error")} if d.get("parity_check") else None))
    print("  " + "  ".join("%s %.3f" % (k, v["ms_per_step"]) for k, v in d["kernels"].items()))
except Exception as e: print("no json", e)
PY
done
