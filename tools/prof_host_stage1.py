"""Which torch ops (host side) launch the small copy / fill kernels of a stage-1 step: python tools/prof_host_stage1.py"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["train_stage1.py", "--iters", "4", "--batch", "128"]
import torch
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    runpy.run_path(os.path.join(ROOT, "examples", "train_stage1.py"), run_name="__main__")
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=50))
