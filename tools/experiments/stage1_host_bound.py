import importlib.util, os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
for B in (16, 32, 16, 24, 8):
    r = b.stage1_iteration(B, 2048, 16)
    print(B, round(r["ms"], 3), [round(x, 2) for x in r["samples_ms"]])
