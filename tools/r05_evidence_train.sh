#!/bin/bash
# Round-5 training evidence after the byte diet of the fused block (LayerNorm3 folded into W1, bf16-pair gradient stream, h1 saved as fragments):
# the full bench line, iteration times, per-kernel stats with dropout 0 / 0.2, HBM-side traffic, SQ counters of k_ff<*> / k_ff_wgrad, phase trace.
# Everything lands in gpurun_out/r05b; copy to profiles/ by hand (names r05b_*).
O=gpurun_out/r05b
mkdir -p $O
python bench.py > $O/bench_T1000.json 2> $O/bench_T1000.err
python tools/bench_train.py --long > $O/bench_train.txt 2>&1
python tools/bench_train.py --long --dropout 0.2 >> $O/bench_train.txt 2>&1
tools/prof_train_dropout.sh $O/train_dropout > $O/kernel_stats_train_p0_p02.txt 2>&1
tools/prof_train_traffic.sh $O/train_traffic > $O/traffic_train.txt 2>&1
PMC_PAT=k_ff tools/prof_train_pmc.sh $O/train_pmc > $O/pmc_train_ff.txt 2>&1
python tools/experiments/trace_train_ff.py $O/trace_train_ff.txt > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
tail -2 $O/traffic_train.txt; grep -v amdgpu $O/bench_train.txt | cut -c1-140
