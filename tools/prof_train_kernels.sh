#!/bin/bash
# Per-kernel time of the training iteration (GPU box): tools/prof_train_kernels.sh [out.csv]   (kernel-trace only, no counters)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(realpath -m ${1:-$R/gpurun_out/train_kernel_stats.csv})
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats -d /tmp/kst --output-format csv -- python $R/tools/bench_train.py $BENCH_ARGS > /tmp/kst.log 2>&1
F=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
mkdir -p $(dirname $OUT); cp $F $OUT
python - "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = 7   # iterations traced by tools/bench_train.py (2 warm-up + 5 timed)
tot = 0
for r in rows[:24]:
    us = float(r["TotalDurationNs"]) / it / 1e3
    tot += us
    print("%8.1f us/iter %5d calls/iter %8.1f us/call  %s" % (us, int(r["Calls"]) // it, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("%8.1f us/iter in all %d kernels" % (sum(float(r["TotalDurationNs"]) for r in rows) / it / 1e3, len(rows)))
print("(the parameter-gradient reductions run on a side stream beside other kernels — k_stem_bwd, k_sum_parts, k_ff_wgrad_finish and whatever they overlap are stretched: the sum exceeds the wall time of an iteration; tools/experiments/iter_sequence.py prints the timeline, tools/bench_train.py --streams 0 the single-stream order)")
PY
