"""HBM roofline of the fused denormalise + relative-L2 metric at the bench rollout shape."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lns_amd import metrics  # noqa: E402

B, T, C, H, W = 64, 64, 3, 128, 128
y = torch.randn(B, T, C, H, W, device="cuda")
yh = y + 0.01 * torch.randn_like(y)
for _ in range(3):
    metrics.relative_l2(yh, y, 0.1, 1.3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 10
for _ in range(N):
    f, s = metrics.relative_l2(yh, y, 0.1, 1.3)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
gb = 2 * y.numel() * 4 / 1e9
print("relative_l2 on [%d,%d,%d,%d,%d]: %.3f ms, %.2f GB read -> %.2f TB/s (HBM peak 8.0); torch reference:" % (B, T, C, H, W, ms, gb, gb / ms))
t0 = time.perf_counter()
yd, gd = yh * 1.3 + 0.1, y * 1.3 + 0.1
ref = (((yd - gd) ** 2).sum((3, 4)) / (gd ** 2).sum((3, 4))).sqrt()
torch.cuda.synchronize()
print("  eager torch %.1f ms ; max rel diff %.2e" % ((time.perf_counter() - t0) * 1e3, ((f - ref).abs() / ref).max().item()))
