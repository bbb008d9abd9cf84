#!/bin/bash
# rocprofv3 evidence for the pointnet2 / SA / metric kernels (GPU box): kernel stats + MFMA-busy counters of the fused SA kernel
set -u
R=$PWD; OUT=$R/gpurun_out/pn2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kstats --output-format csv -- python $R/tools/bench_pointnet2.py > $OUT/kstats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc --output-format csv -- python $R/tools/bench_pointnet2.py > $OUT/pmc.log 2>&1
cd $R
find $OUT/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -16 $OUT/kernel_stats.csv
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$f" k_sa_fused > $OUT/pmc_sa_fused.txt; cat $OUT/pmc_sa_fused.txt
