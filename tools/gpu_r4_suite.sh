#!/bin/bash
# whole GPU suite + smoke at HEAD
OUT=gpurun_out/${1:-r4suite}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "suite rc=$?"; tail -25 $OUT/pytest_gpu.log
