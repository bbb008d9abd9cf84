#!/bin/bash
# Socket power and shader clock while the chain runs (evidence for DESIGN 5.1's "power-limited"): tools/experiments/power_trace.sh
# Samples rocm-smi beside `python bench.py --steps 8` (about 5 s of chain launches) and beside an idle GPU.
smi() { rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk\|mclk" | sed 's/^/    /'; }
echo "== idle"; smi
python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-train-line > /tmp/bench_power.json 2>/dev/null &
BP=$!
for i in $(seq 1 30); do            # ~0.5 s apart from process start to exit: import, setup, then 9 chain launches
  kill -0 $BP 2>/dev/null || break
  echo "== sample $i (bench process alive)"; smi
  sleep 0.3
done
wait $BP
echo "== bench line"; cut -c1-260 /tmp/bench_power.json
