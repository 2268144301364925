#!/usr/bin/env python3
"""Per-kernel durations of the coefficient network (256 x 256 input), inference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile
from hdrnet_amd import models
dev = torch.device("cuda:0")
m = models.HDRNetPointwiseNNGuide().to(dev).eval()
low = torch.rand(1, 256, 256, 3, device=dev)
with torch.no_grad():
    for _ in range(5): m.coefficients(low)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10): m.coefficients(low)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=90))
