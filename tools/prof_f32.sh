#!/bin/bash
# Evidence for the exact-fp32 persistent chain (k_denoise_pipe_f32): kernel stats of the T = 1000, B = 128 launch under
# rocprofv3 --kernel-trace --stats, then the matrix-pipe counters of a T = 20 launch (separate --pmc passes).
# usage: [ROUND=r03] tools/prof_f32.sh
set -u
R=$PWD
O=$R/gpurun_out/${ROUND:-r03}/f32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kstats --output-format csv -- python $R/bench.py --precision f32 --steps 1 --warmup 1 --no-parity --no-train-line --no-cpu-baseline > $O/kstats.log 2>&1
cd $R
find $O/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -4 $O/kernel_stats.csv
grep "^{" $O/kstats.log | tail -1 > $O/bench_under_prof.json; cat $O/bench_under_prof.json
cd /tmp
for set in "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "grbm GRBM_GUI_ACTIVE GRBM_COUNT"; do
  set -- $set; name=$1; shift
  rocprofv3 --pmc "$@" -d $O/$name --output-format csv -- python $R/bench.py --precision f32 --timesteps 20 --steps 1 --warmup 1 --no-parity --no-train-line --no-cpu-baseline > $O/$name.log 2>&1
  f=$(find $O/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" k_denoise > $O/$name.summary.txt 2>&1
  cat $O/$name.summary.txt
done
