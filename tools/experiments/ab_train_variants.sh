#!/bin/bash
# Same-box A/B of training-kernel variants (GPU box): every arm = a rebuild of csrc/train_kernels.hip (+ dfx_common.hip) with extra -D flags, then
# tools/prof_train_kernels.sh (rocprofv3 --kernel-trace --stats over tools/bench_train.py: both arms profiled, comparable) and one un-profiled
# tools/bench_train.py --long for the wall clock.  usage: ab_train_variants.sh out.txt "name:-DFLAG1 -DFLAG2" "name2:" ...   (an empty flag list = the shipped build)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(realpath -m $1); shift
mkdir -p $(dirname $OUT); : > $OUT
cd $R
for arm in "$@"; do
  name=${arm%%:*}; flags=${arm#*:}
  touch difffacto_amd/csrc/train_kernels.hip difffacto_amd/csrc/dfx_common.hip
  python - "$flags" > /tmp/ab_build.log 2>&1 <<'PY'
import sys
from difffacto_amd import build
build.build(verbose=False, extra_flags=sys.argv[1].split())
PY
  if [ $? -ne 0 ]; then echo "== $name [$flags]: BUILD FAILED" >> $OUT; tail -5 /tmp/ab_build.log >> $OUT; continue; fi
  echo "== $name [$flags]" >> $OUT
  bash tools/prof_train_kernels.sh /tmp/ab_$name.csv 2>&1 | grep -E "k_ff|k_stem_bwd|k_head|k_attn|in all" | head -14 >> $OUT
  python tools/bench_train.py --long 2>&1 | tail -1 | cut -c1-160 >> $OUT
  [ -n "$AB_DROPOUT" ] && python tools/bench_train.py --long --dropout 0.2 2>&1 | tail -1 | cut -c1-160 >> $OUT
done
# leave the shipped build behind
touch difffacto_amd/csrc/train_kernels.hip difffacto_amd/csrc/dfx_common.hip
python -c "from difffacto_amd import build; build.build(verbose=False)"
cat $OUT
