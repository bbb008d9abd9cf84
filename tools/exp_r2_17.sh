cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
timeout 900 python -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_reference_pins.py tests/test_gpu_sa_modules.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
python tools/bench_pointnet2.py 2>&1 | grep -v amdgpu.ids | head -12
} > gpurun_out/r2/exp17.log 2>&1
