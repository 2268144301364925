#!/usr/bin/env python3
"""Does keeping the conv weights in channels_last remove the per-call layout conversions?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from hdrnet_amd import models
dev = torch.device("cuda:0")
torch.manual_seed(0)
low = torch.rand(1, 256, 256, 3, device=dev)
def bench(m, tag):
    with torch.no_grad():
        for _ in range(5): m.coefficients(low)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(50): m.coefficients(low)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10): m.coefficients(low)
            torch.cuda.synchronize()
    ev = prof.key_averages()
    n = sum(e.count for e in ev if e.self_device_time_total > 0) / 10
    tot = sum(e.self_device_time_total for e in ev) / 10
    print(f"{tag}: wall {dt * 1e3:.3f} ms, {n:.0f} GPU ops / call, {tot:.0f} us GPU time / call")
    return m.coefficients(low)
m = models.HDRNetPointwiseNNGuide().to(dev).eval()
a = bench(m, "weights contiguous (NCHW)")
m2 = models.HDRNetPointwiseNNGuide().to(dev).eval()
m2.load_state_dict(m.state_dict())
m2 = m2.to(memory_format=torch.channels_last)
b = bench(m2, "weights channels_last")
print("max diff", (a - b).abs().max().item())
with torch.no_grad():
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        m2.coefficients(low); torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0]
for k, c, t in sorted(rows, key=lambda r: -r[2])[:30]:
    print(f"{c:3d} x {t / max(c, 1):6.1f} us  {k[:110]}")
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cpu_time_total", row_limit=30, max_name_column_width=50)[:6000])
