cd $GRAFT_REPO_ROOT
R=$PWD
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2/kstats_train --output-format csv -- python $R/tools/bench_train.py > $R/gpurun_out/r2/kstats_train.log 2>&1
cd $R
find gpurun_out/r2/kstats_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2/kernel_stats_train.csv
head -30 gpurun_out/r2/kernel_stats_train.csv
