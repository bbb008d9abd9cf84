import sys, os, time
sys.path.insert(0, "/root/repo")
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
for p in (0.0, 0.2):
    print("dropout", p, b.stage1_iteration(128, 2048, 16, dropout=p)["ms"])
