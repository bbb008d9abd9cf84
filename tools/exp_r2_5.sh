cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
for F in "" "-DDFX_ABL_GEMM2_BF16" ""; do
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
echo -n "[$F]: "
python bench.py --timesteps 50 --steps 3 --warmup 1 --no-cpu-baseline --no-train-line --no-parity --debug-flags 1 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f  frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
} 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|mfma_linear\|In file included\|generated" > gpurun_out/r2/exp5.log
