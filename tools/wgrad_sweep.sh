for wgs in 512 768 1024 1536 2048 3072; do
  DLRM_WGRAD_WGS=$wgs python tools/microbench.py wgrad 2>&1 | grep -E "^wgrad"
done
for wgs in 1024 2048; do
  DLRM_WGRAD_WGS=$wgs DLRM_WGRAD_TM=2 python tools/microbench.py wgrad 2>&1 | grep -E "^wgrad"
done
DLRM_WGRAD_WGS=2048 DLRM_WGRAD_MINROWS=256 python tools/microbench.py wgrad 2>&1 | grep -E "^wgrad"
