#!/bin/bash
# Round-3 evidence in one gpurun call: headline bench + rocprofv3 stats + PMC (tools/refresh_profiles.sh), the config / batch / fp32
# sweep, the pipe<8> vs pipe2 A/B with pipe2's phase trace, the one-wave-per-SIMD issue micro-benchmark, the pointnet2 timings and the
# printed measurements of the parity gates.
export ROUND=r03
O=gpurun_out/r03
mkdir -p $O
tools/refresh_profiles.sh > $O/refresh.log 2>&1
tools/sweep_bench.sh > $O/sweep_batch_T.txt 2>&1
(echo "# same-box A/B, T = 50, B = 128: k_denoise_pipe<8> (pipe-waves 8) vs k_denoise_pipe2 (64 = two point tiles per wavefront)"; tools/experiments/ab_variants.sh 8 64; echo "# phase trace of k_denoise_pipe2 (wave 0 of workgroup 0; shader cycles)"; tools/experiments/run_trace2.sh | cut -c1-1700) > $O/ab_pipe2.txt 2>&1
tools/ubench/_build/pair_issue > $O/ubench_pair_issue.txt 2>&1
python tools/bench_pointnet2.py > $O/bench_pointnet2.txt 2>&1
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" | tail -120 > $O/parity_prints.txt
tail -3 $O/parity_prints.txt
