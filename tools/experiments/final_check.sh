#!/bin/bash
# End-of-round check on the GPU box: the full -m gpu suite, smoke(), the profile refresh (bench line, rocprofv3 kernel stats of the
# same timed region, PMC passes) and the training measurements.  Everything lands under gpurun_out/$ROUND.
cd ${GRAFT_REPO_ROOT:-.}
export ROUND=${ROUND:-r2}
mkdir -p gpurun_out/$ROUND
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$ROUND/gputests.log
tail -3 gpurun_out/$ROUND/gputests.log
python __graft_entry__.py --smoke 2>&1 | grep -v "amdgpu.ids" | tail -3 | tee gpurun_out/$ROUND/smoke.log
bash tools/refresh_profiles.sh > gpurun_out/$ROUND/refresh.log 2>&1
python tools/bench_train.py 128 2048 bf16 --ab 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/$ROUND/bench_train.txt
python examples/train_stage1.py --iters 8 --batch 128 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/$ROUND/bench_train.txt
cat gpurun_out/$ROUND/bench_train.txt | cut -c1-200
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats -d /tmp/kst --output-format csv -- python $R/tools/bench_train.py > /dev/null 2>&1
cd $R
find /tmp/kst -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$ROUND/kernel_stats_train.csv
bash tools/prof_train_traffic.sh gpurun_out/$ROUND/traffic_fused > gpurun_out/$ROUND/traffic_fused.txt 2>&1
head -3 gpurun_out/$ROUND/traffic_fused.txt
grep '^{' gpurun_out/$ROUND/bench_T1000.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['t100']['shapes_per_s'], d['train_iteration']['ms'], d['train_iteration']['stage1']['ms'])"
head -3 gpurun_out/$ROUND/kernel_stats.csv
