#!/bin/bash
OUT=gpurun_out/${1:-r2u}; mkdir -p $OUT
echo "== pytest"; timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "single_output or (linear_fwd_bwd and 64-1-256)" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -2 $OUT/pytest.log
echo "== micro last layer"; timeout 100 python - <<PY 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0, "tools"); sys.argv=["x","gemm"]
import microbench as mb
mb.gemm([(65536, 1, 256)])
PY
