export DLRM_HIP_LIB=$PWD/dlrm_amd/libdlrm_hip_tuning.so
for d in 0 4 1 5 2; do
  echo "== DLRM_GEMM_DEBUG=$d"
  DLRM_GEMM_DEBUG=$d python tools/microbench.py wgrad 2>&1 | grep -E "^wgrad"
done
