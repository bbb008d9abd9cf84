"""Copy the summaries tools/r04_evidence.sh left under gpurun_out/r04 into profiles/ (tracked): python tools/promote_profiles.py [round]"""
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", R)
MAP = {"bench_T1000.json": "bench_T1000_B128.json", "bench_under_prof.json": "bench_under_rocprof_T1000_B128.json",
       "kernel_stats.csv": "kernel_stats_T1000_B128.csv", "traffic.json": "traffic.json", "bench_train.txt": "bench_train.txt",
       "kernel_stats_train.csv": "kernel_stats_train.csv", "kernel_stats_train.txt": "kernel_stats_train.txt", "traffic_train.txt": "traffic_train.txt",
       "pmc_train_ff.txt": "pmc_train_ff.txt", "bench_pointnet2.txt": "bench_pointnet2.txt", "kernel_stats_pointnet2.csv": "kernel_stats_pointnet2.csv",
       "pmc_sa_fused.txt": "pmc_sa_fused.txt", "sweep_batch_T.txt": "sweep_batch_T.txt", "parity_prints.txt": "parity_prints.txt"}
for a, b in MAP.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(ROOT, "profiles", f"{R}_{b}"))
        print("promoted", a, "->", f"profiles/{R}_{b}")
    else:
        print("MISSING", a)
pm = os.path.join(src, "pmc")
if os.path.isdir(pm):
    with open(os.path.join(ROOT, "profiles", f"{R}_pmc_chain_T20_B128.txt"), "w") as out:
        for name in ("sq1", "sq2", "grbm", "tcc1", "tcc2"):
            f = os.path.join(pm, name + ".summary.txt")
            if os.path.exists(f):
                out.write(f"== pass {name}\n" + open(f).read())
    print("promoted pmc summaries")
