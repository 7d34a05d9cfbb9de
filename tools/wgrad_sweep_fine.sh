for wgs in 512 576 640 704 768 832 896 960 1024 1088 1152 1280; do
  DLRM_WGRAD_WGS=$wgs python tools/microbench.py wgrad 2>&1 | grep -E "^wgrad"
done
