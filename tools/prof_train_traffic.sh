#!/bin/bash
# HBM-side traffic of one training iteration from rocprofv3 PMC counters, separate passes for FETCH_SIZE and
# WRITE_SIZE (never combined with tracing), summed over ALL kernels of tools/bench_train.py (2 warm-up + 5 timed iterations = 7):
#   tools/prof_train_traffic.sh <outdir> [extra bench_train args, e.g. --layerwise]
set -u
REPO=$PWD
OUT=$(realpath -m $1); shift
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d $OUT/$C --output-format csv -- python $REPO/tools/bench_train.py 128 2048 bf16 "$@" > $OUT/$C.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    s = 0.0
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == c:
            v = float(row["Counter_Value"]); s += v
            per[row["Kernel_Name"].split("(")[0][-48:]][c] += v
    tot[c] = s
it = 7.0
# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section)
fetch, write = tot["FETCH_SIZE"] * 1024 * 2 / it, tot["WRITE_SIZE"] * 1024 / it
print(f"per training iteration (B=128 x 2048, bf16 products): FETCH_SIZE x2 = {fetch / 1e9:.2f} GB, WRITE_SIZE = {write / 1e9:.2f} GB, total {(fetch + write) / 1e9:.2f} GB")
for k, d in sorted(per.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] * 2 + kv[1]['WRITE_SIZE']))[:12]:
    print(f"  {k:50s} read {d['FETCH_SIZE'] * 2048 / it / 1e9:7.3f} GB  write {d['WRITE_SIZE'] * 1024 / it / 1e9:7.3f} GB")
PY
