#!/bin/bash
# visit 4: where the time of seg_scatter_kernel goes (timing-only debug switches; results of the switched runs are wrong by design)
OUT=gpurun_out/v4; mkdir -p $OUT
for dbg in 0 1 2 4 3 7; do
  DLRM_SEG_DEBUG=$dbg timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1 | sed "s/^/dbg=$dbg  /"
done
DLRM_SORT=rocprim timeout 120 python tools/sort_bench.py 65536 2>&1 | tail -1
