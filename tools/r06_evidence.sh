#!/bin/bash
# Round-6 evidence in one gpurun call (GPU box); everything lands in gpurun_out/r06, tools/promote_profiles.py r06 copies the summaries to profiles/.
#   1 headline: bench line, rocprofv3 --kernel-trace --stats of the SAME command, PMC passes (tools/refresh_profiles.sh), the traffic model at two chain
#     lengths (tools/prof_write_size.sh -> profiles/r06_traffic.json by hand: fixed + per-step bytes, the WRITE_SIZE constant itemised)
#   2 training: iteration time with dropout 0 / 0.2, per-kernel stats, HBM-side traffic, SQ counters of k_ff<*> / k_ff_wgrad / k_ff_fwd_chain, chain phase trace
#   3 the parity gates' printed measurements (whole -m gpu suite)
export ROUND=r06
O=gpurun_out/r06
mkdir -p $O
tools/refresh_profiles.sh > $O/refresh.log 2>&1
tools/prof_write_size.sh $O/write_size > $O/write_size_scaling.txt 2>&1
python tools/bench_train.py --long > $O/bench_train.txt 2>&1
python tools/bench_train.py --long --dropout 0.2 >> $O/bench_train.txt 2>&1
tools/prof_train_dropout.sh $O/train_dropout > $O/kernel_stats_train_p0_p02.txt 2>&1
tools/prof_train_kernels.sh $O/kernel_stats_train.csv > $O/kernel_stats_train.txt 2>&1
tools/prof_train_traffic.sh $O/train_traffic > $O/traffic_train.txt 2>&1
PMC_PAT=k_ff tools/prof_train_pmc.sh $O/train_pmc > $O/pmc_train_ff.txt 2>&1
python tools/experiments/trace_ff_chain.py $O/trace_ff_chain.txt -DDFX_TRACE_FF_CHAIN > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $O/parity_prints.txt
tail -3 $O/parity_prints.txt; grep -v amdgpu $O/bench_train.txt | cut -c1-150; tail -2 $O/traffic_train.txt
