"""Re-run one case of tools/fuzz_parity.py's fused-training family with its numbers printed (GPU box)."""
import sys
import traceback

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import test_gpu_train as tt

B, N, drop = 5, 160, (0.2, 717135)
try:
    tt.test_fused_feed_forward_matches_the_layer_by_layer_bf16_path(B, N, drop)
    print("passed")
except AssertionError:
    traceback.print_exc()
