cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python tools/exp_zero_weights.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/exp6.log
