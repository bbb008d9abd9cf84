#!/bin/bash
# Round-4 evidence in one gpurun call (GPU box); everything lands in gpurun_out/r04, tools/promote_profiles.py copies the summaries to profiles/.
#   1 headline: bench line, rocprofv3 --kernel-trace --stats of the SAME command, PMC passes incl. FETCH/WRITE_SIZE (tools/refresh_profiles.sh)
#   2 training: per-kernel stats, HBM-side traffic (FETCH x2 + WRITE), SQ counters of k_ff<*> / k_ff_wgrad, phase trace
#   3 pointnet2 / SA / metric kernels: timings + kernel stats
#   4 batch / T / precision sweep; the parity gates' printed measurements
export ROUND=r04
O=gpurun_out/r04
mkdir -p $O
tools/refresh_profiles.sh > $O/refresh.log 2>&1
python - <<'PY'
import csv, glob, json, os
o = "gpurun_out/r04/pmc"
def total(name, counter):
    f = glob.glob(f"{o}/{name}/**/*counter_collection.csv", recursive=True)
    s = 0.0
    for row in csv.DictReader(open(f[0])):
        if row["Counter_Name"] == counter and "k_denoise_pipe" in row["Kernel_Name"]:
            s += float(row["Counter_Value"])
    return s
try:
    T, launches = 20, 2   # tools/prof_pmc.sh runs --timesteps 20 --steps 1 --warmup 1: two launches of the chain kernel
    fetch, write = total("tcc1", "FETCH_SIZE") * 1024 / launches, total("tcc2", "WRITE_SIZE") * 1024 / launches
    d = {"B": 128, "N": 2048, "T_profiled": T, "fetch_size_bytes_reported": fetch, "fetch_size_bytes_corrected_x2": 2 * fetch, "write_size_bytes": write,
         "bytes_per_launch_T20": 2 * fetch + write, "bytes_per_step": (2 * fetch + write) / T, "round": "r04",
         "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.sh via tools/refresh_profiles.sh), T=20 chain, B=128 x 2048, per launch of "
                "k_denoise_pipe<8>; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (128-B requests tallied at 64 B); counter values are KiB"}
    json.dump(d, open("gpurun_out/r04/traffic.json", "w"), indent=1)
    print("traffic per diffusion step:", d["bytes_per_step"])
except Exception as e:
    print("traffic summary failed:", repr(e))
PY
python tools/bench_train.py > $O/bench_train.txt 2>&1
tools/prof_train_kernels.sh $O/kernel_stats_train.csv > $O/kernel_stats_train.txt 2>&1
tools/prof_train_traffic.sh $O/train_traffic > $O/traffic_train.txt 2>&1
PMC_PAT=k_ff tools/prof_train_pmc.sh $O/train_pmc > $O/pmc_train_ff.txt 2>&1
python tools/experiments/trace_train_ff.py $O/trace_train_ff.txt > /dev/null 2>&1
python tools/bench_pointnet2.py > $O/bench_pointnet2.txt 2>&1
tools/prof_pointnet2.sh > $O/prof_pointnet2.log 2>&1
cp gpurun_out/pn2/kernel_stats.csv $O/kernel_stats_pointnet2.csv 2>/dev/null
cp gpurun_out/pn2/pmc_sa_fused.txt $O/pmc_sa_fused.txt 2>/dev/null
tools/sweep_bench.sh > $O/sweep_batch_T.txt 2>&1
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $O/parity_prints.txt
tail -3 $O/parity_prints.txt
