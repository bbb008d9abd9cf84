set -u
mkdir -p gpurun_out/r1
python bench.py > gpurun_out/r1/bench_T1000.json 2> gpurun_out/r1/bench_T1000.err
cat gpurun_out/r1/bench_T1000.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1/kstats --output-format csv -- python $R/bench.py --timesteps 100 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r1/kstats.log 2>&1
cd $R
find gpurun_out/r1/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r1/kernel_stats.csv
head -12 gpurun_out/r1/kernel_stats.csv
tools/prof_pmc.sh gpurun_out/r1/pmc --timesteps 20 --steps 1 --warmup 1
