#!/usr/bin/env python3
"""Per-parameter gradient errors of the HIP coefficient-network backward against float64 autograd (debug aid)."""
import copy
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdrnet_amd import models

params = dict(batch_norm=False)
for a in sys.argv[2:]:
    k, v = a.split("=")
    params[k] = int(v)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(21)
m = models.HDRNetPointwiseNNGuide(params)
g = torch.Generator().manual_seed(7)
with torch.no_grad():
    for name, p in m.named_parameters():
        if p.dim() == 1:
            p.copy_(0.2 * torch.randn(p.shape, generator=g))
N = m.params["net_input_size"]
low = torch.rand(B, N, N, 3)
ref = copy.deepcopy(m.coefficients).double()
out64 = ref(low.double())
wts = torch.randn(out64.shape, dtype=torch.float64)
(out64 * wts).sum().backward()
net = m.coefficients.to("cuda:0")
lowd, wd = low.cuda(), wts.float().cuda()
print("native training:", net._use_native_training(lowd))
out = net(lowd)
(out * wd).sum().backward()
print("forward err", float((out.detach().cpu().double() - out64).abs().max()) / float(out64.abs().max()))
for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
    g64 = q.grad
    scale = float(g64.abs().max()) + 1e-30
    e = float((p.grad.cpu().double() - g64).abs().max()) / scale
    print(f"{name:28s} {tuple(p.shape)!s:20s} rel err {e:.3e}   |g| {scale:.3e}")
