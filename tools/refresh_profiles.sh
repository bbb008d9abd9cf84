set -u
mkdir -p gpurun_out/r1
python bench.py > gpurun_out/r1/bench_T1000.json 2> gpurun_out/r1/bench_T1000.err
cat gpurun_out/r1/bench_T1000.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
# the SAME command as the headline bench line (python bench.py: T=1000, 3 steps + 1 warm-up), under the kernel tracer
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1/kstats --output-format csv -- python $R/bench.py > $R/gpurun_out/r1/kstats.log 2>&1
cd $R
find gpurun_out/r1/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r1/kernel_stats.csv
head -6 gpurun_out/r1/kernel_stats.csv
tail -1 gpurun_out/r1/kstats.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line under rocprofv3: kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'])"
tools/prof_pmc.sh gpurun_out/r1/pmc --timesteps 20 --steps 1 --warmup 1
