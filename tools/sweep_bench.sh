#!/bin/bash
# SURVEY 8(d) config 2 sweep on one GPU: batch size, chain length and the fp32 variant.  tools/sweep_bench.sh > gpurun_out/sweep.txt
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-line "$@" 2>/dev/null | tail -1 | python -c '
import json, sys
r = json.loads(sys.stdin.readline())
c = r["config"]
print("B=%-5d N=%-5d T=%-5d %s: %9.2f shapes/s  %9.2f ms per batch  %7.1f TFLOP/s (frac %.3f)" % (c["batch_per_gpu"], c["npoints"], c["num_timesteps"], r["dtype"], r["value"], r["ms_per_step"], r["roofline"]["achieved"], r["roofline"]["frac"]))'; }
for b in 1 8 32 128 512 1024; do run --batch $b; done
run --batch 128 --timesteps 100
run --batch 128 --npoints 8192 --timesteps 100
run --batch 128 --precision f32 --timesteps 100
