cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
bash tools/prof_train_traffic.sh gpurun_out/r2/traffic_fused > gpurun_out/r2/traffic_fused.txt 2>&1
bash tools/prof_train_traffic.sh gpurun_out/r2/traffic_layer --layerwise > gpurun_out/r2/traffic_layer.txt 2>&1
python tools/bench_train.py 128 2048 bf16 --ab 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r2/bench_train.txt
python examples/train_stage1.py --iters 8 --batch 128 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r2/bench_train.txt
