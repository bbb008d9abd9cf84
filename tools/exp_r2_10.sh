cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
for F in "-DDFX_FF_NW=4 -DDFX_FF_NBUF=2" "-DDFX_FF_NW=8 -DDFX_FF_NBUF=3" "-DDFX_FF_NW=8 -DDFX_FF_NBUF=2"; do
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
echo "[$F]"
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -s -k "fused or bf16_matrix" 2>&1 | grep "fused vs\|passed\|failed\|bf16 products\|Error"
R=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats -d /tmp/kst --output-format csv -- python $R/tools/bench_train.py 2>&1 | grep "training iteration" | cut -c1-130
cd $R
find /tmp/kst -name "*kernel_stats.csv" | head -1 | xargs grep "k_ff<" | cut -c1-120
done
} 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|mfma_linear\|In file included\|generated" > gpurun_out/r2/exp10.log
