set -x
mkdir -p gpurun_out/v1
python tools/vendor_vs_own.py --dtype f32 --md gpurun_out/v1/vendor_vs_own_f32.md > gpurun_out/v1/f32.log 2>&1
python tools/vendor_vs_own.py --dtype bf16 --md gpurun_out/v1/vendor_vs_own_bf16.md > gpurun_out/v1/bf16.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/v1/names_f32 -o n -- python $GRAFT_REPO_ROOT/tools/vendor_vs_own.py --dtype f32 --names > $GRAFT_REPO_ROOT/gpurun_out/v1/names_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/v1/names_bf16 -o n -- python $GRAFT_REPO_ROOT/tools/vendor_vs_own.py --dtype bf16 --names > $GRAFT_REPO_ROOT/gpurun_out/v1/names_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
tail -30 gpurun_out/v1/f32.log; tail -20 gpurun_out/v1/bf16.log
