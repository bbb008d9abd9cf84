#!/bin/bash
# Per-kernel time of the whole stage-1 training step (GPU box): tools/prof_stage1_kernels.sh [out.csv]   (kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(realpath -m ${1:-$R/gpurun_out/stage1_kernel_stats.csv})
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst1
rocprofv3 --kernel-trace --stats -d /tmp/kst1 --output-format csv -- python $R/examples/train_stage1.py --iters 8 --batch 128 $STAGE1_ARGS > /tmp/kst1.log 2>&1
F=$(find /tmp/kst1 -name "*kernel_stats.csv" | head -1)
mkdir -p $(dirname $OUT); cp $F $OUT
python - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = 8
for r in rows[:32]:
    print("%8.1f us/iter %5d calls/iter %8.1f us/call  %s" % (float(r["TotalDurationNs"]) / it / 1e3, int(r["Calls"]) // it, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("%8.1f us/iter in all %d kernels" % (sum(float(r["TotalDurationNs"]) for r in rows) / it / 1e3, len(rows)))
PY
