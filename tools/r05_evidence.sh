#!/bin/bash
# Round-5 evidence in one gpurun call (GPU box); everything lands in gpurun_out/r05, tools/promote_profiles.py r05 copies the summaries to profiles/.
#   1 headline: bench line, rocprofv3 --kernel-trace --stats of the SAME command, PMC passes (tools/refresh_profiles.sh); the traffic model
#     (profiles/r05_traffic.json, two chain lengths) comes from tools/prof_write_size.sh and is not regenerated here
#   2 training: iteration time with dropout 0 / 0.2, HBM-side traffic (FETCH x2 + WRITE), SQ counters of k_ff<*> / k_ff_wgrad
#   3 batch / T / precision sweep; the parity gates' printed measurements
export ROUND=r05
O=gpurun_out/r05
mkdir -p $O
tools/refresh_profiles.sh > $O/refresh.log 2>&1
python tools/bench_train.py > $O/bench_train.txt 2>&1
python tools/bench_train.py --dropout 0.2 >> $O/bench_train.txt 2>&1
tools/prof_train_kernels.sh $O/kernel_stats_train.csv > $O/kernel_stats_train.txt 2>&1
tools/prof_train_traffic.sh $O/train_traffic > $O/traffic_train.txt 2>&1
PMC_PAT=k_ff tools/prof_train_pmc.sh $O/train_pmc > $O/pmc_train_ff.txt 2>&1
if [ -z "${SKIP_SWEEP:-}" ]; then
tools/sweep_bench.sh > $O/sweep_batch_T.txt 2>&1
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $O/parity_prints.txt
tail -3 $O/parity_prints.txt
fi
