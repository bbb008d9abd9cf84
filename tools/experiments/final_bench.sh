python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>/dev/null | grep "^{" > gpurun_out/bench_final.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench_final.json"))
print(d["value"], d["roofline"]["frac"], d["t100"]["shapes_per_s"], d["t100"]["wall_shapes_per_s"], d["train_iteration"]["ms"], d["train_iteration"]["stage1"]["ms"], d["train_iteration"]["stage1"]["encoder_bf16_ms"])
PY
