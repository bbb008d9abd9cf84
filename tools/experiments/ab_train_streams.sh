#!/bin/bash
# Same-box A/B of the training iteration over the side-stream sets (GPU box): bash tools/experiments/ab_train_streams.sh "0 1 2 18 ..."
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/streams
MODES=${1:-"0 1"}
for r in 1 2 3; do
  for m in $MODES; do
    python tools/bench_train.py 128 2048 bf16 --long --streams $m 2>&1 | grep "training iteration" | cut -c70-140 | sed "s/^/[streams $m] /"
  done
done | tee gpurun_out/streams/ab.txt
