set -u
# usage: [ROUND=r2] tools/refresh_profiles.sh   (bench line, kernel stats under rocprofv3 for the SAME command, PMC passes)
mkdir -p gpurun_out/${ROUND:-r03}
python bench.py > gpurun_out/${ROUND:-r03}/bench_T1000.json 2> gpurun_out/${ROUND:-r03}/bench_T1000.err
cat gpurun_out/${ROUND:-r03}/bench_T1000.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
# the SAME timed region as the headline bench line (python bench.py: T=1000, 3 steps + 1 warm-up) under the kernel tracer; the
# secondary blocks that run AFTER the timed region (parity / T=100 / cpu baseline / training line) are switched off because they
# launch the same kernel name with other shapes and would pollute its average
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${ROUND:-r03}/kstats --output-format csv -- python $R/bench.py --no-parity --no-train-line --no-cpu-baseline > $R/gpurun_out/${ROUND:-r03}/kstats.log 2>&1
cd $R
find gpurun_out/${ROUND:-r03}/kstats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${ROUND:-r03}/kernel_stats.csv
head -6 gpurun_out/${ROUND:-r03}/kernel_stats.csv
grep "^{" gpurun_out/${ROUND:-r03}/kstats.log | tail -1 | tee gpurun_out/${ROUND:-r03}/bench_under_prof.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line under rocprofv3: kernel_ms', d['roofline']['kernel_ms'], 'value', d['value'])"
tools/prof_pmc.sh gpurun_out/${ROUND:-r03}/pmc --timesteps 20 --steps 1 --warmup 1 --no-parity --no-train-line
