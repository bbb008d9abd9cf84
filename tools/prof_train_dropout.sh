#!/bin/bash
# Per-kernel time of the training iteration with dropout 0 and with the shipped dropout 0.2 (GPU box; kernel-trace only, no counters):
#   tools/prof_train_dropout.sh <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$(realpath -m ${1:-$R/gpurun_out/train_dropout})
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for P in 0 0.2; do
  rm -rf /tmp/kst_$P
  rocprofv3 --kernel-trace --stats -d /tmp/kst_$P --output-format csv -- python $R/tools/bench_train.py 128 2048 bf16 --dropout $P > $OUT/bench_p$P.log 2>&1
  F=$(find /tmp/kst_$P -name "*kernel_stats.csv" | head -1)
  cp $F $OUT/kernel_stats_p$P.csv
  echo "== dropout $P: $(tail -1 $OUT/bench_p$P.log)"
  python - "$OUT/kernel_stats_p$P.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = 7   # iterations traced by tools/bench_train.py (2 warm-up + 5 timed)
for r in rows[:12]:
    us = float(r["TotalDurationNs"]) / it / 1e3
    print("%8.1f us/iter %5d calls/iter %8.1f us/call  %s" % (us, int(r["Calls"]) // it, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("%8.1f us/iter in all %d kernels" % (sum(float(r["TotalDurationNs"]) for r in rows) / it / 1e3, len(rows)))
PY
done
