"""nmf_topk_select (radix select) against nmf_argsort_f32 (rocPRIM radix sort of all keys) on the key counts of a training step:
    python tools/select_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nmf_amd import hip  # noqa: E402

dev = torch.device("cuda", 0)
for n in (242000, 500000):
    keys = torch.rand(n, device=dev) * 2
    for name, fn in [("argsort_f32", lambda: hip.argsort_f32(keys))] + [
            (f"topk_select k={k}", (lambda k: (lambda: hip.topk_select(keys, k)))(k)) for k in (1000, 4096, 20000, n // 2)]:
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"n={n} {name:28s} {a.elapsed_time(b) / 50 * 1e3:8.1f} us per call")
