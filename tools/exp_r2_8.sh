cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_encoder_train.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | grep "fused\|passed\|failed\|bf16 products\|Error" > gpurun_out/r2/exp8.log
timeout 600 python tools/bench_train.py 128 2048 bf16 --ab 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r2/exp8.log
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2/kstats_train --output-format csv -- python $R/tools/bench_train.py > $R/gpurun_out/r2/kstats_train.log 2>&1
cd $R
find gpurun_out/r2/kstats_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2/kernel_stats_train.csv
head -12 gpurun_out/r2/kernel_stats_train.csv >> gpurun_out/r2/exp8.log
