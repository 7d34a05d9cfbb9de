#!/bin/bash
# Round-5 opener (prepared at the end of round 4, not yet run): what the bf16 / bf16x6 GEMM calls spend outside their k-loops, and which GPU this is.
#   1. build the tuning library ON THE CPU SIDE first (tools/build_tuning_lib.sh) — it travels with the snapshot;
#   2. DLRM_BF16_DEBUG = 0 / 8 (8 = no epilogue): the difference is the epilogue's share of a call (profiles/round4/bf16_gemm_notes.md put
#      ~85 us of a 150 us 65536 x 1024 x 1024 call outside the MFMA stream; this splits it);
#   3. a quick bench line: box.node names the GPU, mfma_bf16_random_tflops says what its matrix pipe holds under a GEMM's bit toggling
#      (profiles/round4/box_classes.md: the slow GPU of the pool has not met that probe yet).
OUT=gpurun_out/${1:-r5probe}
mkdir -p $OUT
export TMPDIR=/tmp
if [ -f dlrm_amd/libdlrm_hip_tuning.so ]; then
  export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
  for dbg in 0 8; do
    echo "== DLRM_BF16_DEBUG=$dbg (8 = no epilogue)"
    _BF16_BENCH_CHILD=1 DLRM_BF16_PHASED=1 DLRM_BF16_DEBUG=$dbg BF16_BENCH_SHAPES=2 timeout 200 python tools/bf16_gemm_bench.py 2>&1 | grep -E "^65536"
    DLRM_BF16_DEBUG=$dbg timeout 200 python tools/bf16x6_gemm_bench.py 2>&1 | grep -E "^65536x1024x1024|^65536x512x256" | cut -c1-260
  done | tee $OUT/epilogue_share.txt
  unset DLRM_HIP_LIB
else
  echo "no tuning library: run tools/build_tuning_lib.sh before gpurun"
fi
timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt-arith --no-alt-overlap --no-alt-fuse --no-parity-check > $OUT/bench_quick.json 2> /dev/null
python - <<PY
import json
d = json.load(open("$OUT/bench_quick.json")); b = d["box"]
print("ms %.3f" % d["ms_per_step"], {k: round(v, 1) for k, v in b.items() if isinstance(v, float)})
print("GPU", b["node"]["before"].get("smi", {}).get("card0", {}).get("Unique ID"), "cards at high clock", b["node"]["before"].get("cards_at_high_clock"))
PY
