#!/bin/bash
# Launch-by-launch timeline of one stage-1 training step in bench.py's loop (no host sync between iterations); GPU box: tools/experiments/stage1_sequence.sh [out.txt]
R=$PWD; OUT=$(realpath -m ${1:-$R/gpurun_out/stage1_seq.txt}); cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt1
cat > /tmp/s1loop.py <<PY
import importlib.util, sys
sys.path.insert(0, "$R")
spec = importlib.util.spec_from_file_location("bench", "$R/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.stage1_iteration(128, 2048, 8)["ms"])
PY
rocprofv3 --kernel-trace -d /tmp/kt1 --output-format csv -- python /tmp/s1loop.py > /tmp/kt1.log 2>&1
F=$(find /tmp/kt1 -name "*kernel_trace.csv" | head -1)
python $R/tools/experiments/iter_sequence.py $F k_build_xin | cut -c1-150 > $OUT
tail -1 /tmp/kt1.log; tail -1 $OUT
