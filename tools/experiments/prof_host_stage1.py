"""Which torch ops (host side) launch the small copy / fill kernels of a stage-1 step, by operand shapes:
python tools/experiments/prof_host_stage1.py [op ...]   (default: aten::copy_ aten::fill_ aten::zero_ aten::zeros aten::clone aten::contiguous)"""
import os, sys, runpy
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ops = sys.argv[1:] or ["aten::copy_", "aten::fill_", "aten::zero_"]
ITERS = 4
sys.argv = ["train_stage1.py", "--iters", str(ITERS), "--batch", "128"]
import torch
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    runpy.run_path(os.path.join(ROOT, "examples", "train_stage1.py"), run_name="__main__")
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
avg = prof.key_averages(group_by_input_shape=True)
for op in ops:
    rows = sorted((e for e in avg if e.key == op), key=lambda e: -e.count)
    print(f"\n{op}: calls per {ITERS}-iteration run (+ model construction) by input shapes")
    for e in rows[:40]:
        print(f"  {e.count:5d}  {e.device_time_total / max(e.count, 1):8.1f} us each  {e.input_shapes}")
