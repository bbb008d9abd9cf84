#!/bin/bash
# Launch-by-launch timeline of one stage-1 training step (GPU box): tools/experiments/stage1_sequence.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt1
rocprofv3 --kernel-trace -d /tmp/kt1 --output-format csv -- python $R/examples/train_stage1.py --iters 8 --batch 128 > /tmp/kt1.log 2>&1
F=$(find /tmp/kt1 -name "*kernel_trace.csv" | head -1)
python $R/tools/experiments/iter_sequence.py $F k_build_xin | cut -c1-140
