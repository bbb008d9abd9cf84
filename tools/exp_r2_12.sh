cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
for B in 1 32; do
 echo "B=$B: "
 python bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-train-line --no-parity 2>&1 | grep -v amdgpu.ids | tail -6
done
timeout 900 python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_headline.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r2/exp12.log 2>&1
