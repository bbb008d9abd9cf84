cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r2/gputests.log
tail -4 gpurun_out/r2/gputests.log
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -5 > gpurun_out/r2/smoke.log
cat gpurun_out/r2/smoke.log
ROUND=r2 bash tools/refresh_profiles.sh > gpurun_out/r2/refresh.log 2>&1
tail -30 gpurun_out/r2/refresh.log
