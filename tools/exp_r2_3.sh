cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r2/gputests.log
tail -15 gpurun_out/r2/gputests.log
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -5 > gpurun_out/r2/smoke.log
python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err
tail -c 3000 gpurun_out/r2/bench.json
