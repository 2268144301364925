#!/usr/bin/env python3
"""Training-step time of HDRNetCurves (the reference's default model) at config #4's size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from hdrnet_amd import models
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = models.HDRNetCurves(dict(batch_norm=True)).to(dev).train()
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
B = 4
low = torch.rand(B, 256, 256, 3, device=dev)
full = torch.rand(B, 1080, 1920, 3, device=dev)
target = torch.rand(B, 1080, 1920, 3, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = (m(low, full) - target).square().mean()
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"HDRNetCurves training step 4 x 1080p: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
