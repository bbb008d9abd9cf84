R=$PWD; mkdir -p gpurun_out/seq
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kst /tmp/kst1
rocprofv3 --kernel-trace -d /tmp/kst --output-format csv -- python $R/tools/bench_train.py > gpurun_out_bt.log 2>&1
F=$(find /tmp/kst -name "*kernel_trace.csv" | head -1)
python $R/tools/experiments/iter_sequence.py $F k_adam > $R/gpurun_out/seq/train_seq.txt
python $R/tools/experiments/timeline.py $F k_adam > $R/gpurun_out/seq/train_timeline.txt
rocprofv3 --kernel-trace -d /tmp/kst1 --output-format csv -- python $R/examples/train_stage1.py --iters 8 --batch 128 > /tmp/s1.log 2>&1
F=$(find /tmp/kst1 -name "*kernel_trace.csv" | head -1)
python $R/tools/experiments/iter_sequence.py $F k_adam > $R/gpurun_out/seq/stage1_seq.txt
python $R/tools/experiments/timeline.py $F k_adam > $R/gpurun_out/seq/stage1_timeline.txt
tail -3 /tmp/s1.log; tail -2 /tmp/gpurun_out_bt.log; tail -2 $R/gpurun_out/seq/train_seq.txt
