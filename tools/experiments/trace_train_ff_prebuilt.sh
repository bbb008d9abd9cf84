#!/bin/bash
# Phase trace with a library built beforehand (dev container: python -c "from difffacto_amd import build; build.build(force=True, extra_flags=['-DDFX_TRACE_FF'])",
# copy libdfx.so to <trace.so>, rebuild the plain one):  tools/experiments/trace_train_ff_prebuilt.sh <trace.so> <out.txt> [bench_train args]
T=$1; OUT=$2; shift 2
L=difffacto_amd/libdfx.so; mkdir -p $(dirname $OUT)
cp $L /tmp/libdfx_plain.so; cp $T $L
DFX_TRACE_FF_OUT=/tmp/ff_trace_raw.txt python tools/bench_train.py "$@" | tail -1 | cut -c1-160 > $OUT
python - >> $OUT <<PY
import sys
sys.path.insert(0, "tools")
import trace_train_ff as t
t.analyse("/tmp/ff_trace_raw.txt", sys.stdout)
PY
cp /tmp/libdfx_plain.so $L
