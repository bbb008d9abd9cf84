cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_encoder_train.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r2/exp8.log
timeout 600 python tools/bench_train.py 128 2048 bf16 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200 >> gpurun_out/r2/exp8.log
python examples/train_stage1.py --iters 8 --batch 128 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r2/exp8.log
