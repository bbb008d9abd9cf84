#!/bin/bash
# SURVEY 8(d) configs 2-3 on one GPU: batch size, chain length, the exact-fp32 chain, and the gen_airplane / gen_car / gen_lamp
# configs (noise_scale, npoints) at the headline T.  tools/sweep_bench.sh > gpurun_out/sweep.txt
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-line --no-parity "$@" 2>/dev/null | tail -1 | python -c '
import json, sys
r = json.loads(sys.stdin.readline())
c = r["config"]
print("B=%-5d N=%-5d T=%-5d %-4s: %9.2f shapes/s  %9.2f ms per batch  kernel %9.2f ms  %7.1f TFLOP/s (frac %.3f)" % (c["batch_per_gpu"], c["npoints"], c["num_timesteps"], r["dtype"], r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"], r["roofline"]["frac"]))'; }
echo "# bf16 (k_denoise_pipe / k_denoise_coop), gen_chair, T = 1000"
for b in 1 8 32 128 512 1024; do run --batch $b; done
echo "# shipped chain length T = 100"
run --batch 128 --timesteps 100
run --batch 128 --npoints 8192 --timesteps 100
echo "# exact fp32 (k_denoise_pipe_f32), T = 1000 (B = 1024: one step)"
run --batch 1 --precision f32
run --batch 128 --precision f32
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-line --no-parity --batch 1024 --precision f32 2>/dev/null | tail -1 | python -c '
import json, sys
r = json.loads(sys.stdin.readline()); c = r["config"]
print("B=%-5d N=%-5d T=%-5d %-4s: %9.2f shapes/s  %9.2f ms per batch  kernel %9.2f ms  %7.1f TFLOP/s (frac %.3f)" % (c["batch_per_gpu"], c["npoints"], c["num_timesteps"], r["dtype"], r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"], r["roofline"]["frac"]))'
run --batch 128 --precision f32 --timesteps 100
echo "# exact fp32, direct kernel (r02's path), T = 100"
run --batch 128 --precision f32 --timesteps 100 --force-direct
echo "# configs[2]: gen_airplane / gen_car / gen_lamp at T = 1000 (bench.py's sweep block: latents with the config's noise_scale)"
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-line 2>/dev/null | tail -1 | python -c '
import json, sys
r = json.loads(sys.stdin.readline())
for k, v in r["sweep"].items():
    print("%-16s B=%-5d N=%-5d T=%-5d %-4s noise_scale %5.1f: kernel %9.2f ms  %8.2f shapes/s  (wall %8.2f)  frac %.3f" % (k, v["batch"], v["npoints"], v["num_timesteps"], v["dtype"], v["noise_scale"], v["kernel_ms"], v["shapes_per_s"], v["wall_shapes_per_s"], v["frac"]))
v = r["f32"]; print("%-16s B=%-5d N=%-5d T=%-5d %-4s: kernel %9.2f ms  %8.2f shapes/s  frac %.3f" % ("f32 chain", v["batch"], v["npoints"], v["num_timesteps"], v["dtype"], v["kernel_ms"], v["shapes_per_s"], v["frac"]))
v = r["t100"]; print("%-16s kernel %.2f ms = %.1f shapes/s; wall (latents + context + chain) %.2f ms = %.1f shapes/s" % ("T=100", v["kernel_ms"], v["shapes_per_s"], v["wall_ms"], v["wall_shapes_per_s"]))'
