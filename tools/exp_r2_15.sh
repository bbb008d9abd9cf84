cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r2/gputests.log
tail -3 gpurun_out/r2/gputests.log
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -3
