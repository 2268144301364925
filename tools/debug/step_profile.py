#!/usr/bin/env python3
"""Kernel-level breakdown of the config #4 training step (torch.profiler, one MI355X)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from hdrnet_amd import models  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
mt = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
opt = torch.optim.Adam([p for p in mt.parameters() if p.requires_grad], lr=1e-4)
B = 4
low = torch.rand(B, 256, 256, 3, device=dev)
full = torch.rand(B, 1080, 1920, 3, device=dev)
target = torch.rand(B, 1080, 1920, 3, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss = (mt(low, full) - target).square().mean()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
