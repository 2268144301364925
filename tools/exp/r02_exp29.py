"""Phase timeline of the fused backward (tools variant 9): where does a wave's chunk time go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import WORKLOADS
from hdrnet_amd import _lib
lib = _lib.load_tools()
H, W, GH, GW, GD, desc = WORKLOADS['4k']
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev).manual_seed(1)
grid = torch.rand((1, GH, GW, GD, 12), device=dev, generator=gen)
guide = torch.rand((1, H, W), device=dev, generator=gen)
inp = torch.rand((1, H, W, 3), device=dev, generator=gen)
dout = torch.randn((1, H, W, 3), device=dev, generator=gen)
dgrid = torch.empty_like(grid); dguide = torch.empty_like(guide); dinput = torch.empty_like(inp)
st = torch.cuda.current_stream(dev).cuda_stream
wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(1, H, W, GH, GW, GD, 3, 3, 1)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
NT, NC = 17 * 540, 16   # upper bound on tasks (rg >= 4), chunks recorded per task
trace = torch.zeros((NT * NC * 5,), dtype=torch.int64, device=dev)
def run(case, variant):
    dg, dgu, di = {"all": (1, 1, 1), "gg": (1, 1, 0), "g": (1, 0, 0)}[case]
    rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
        dgrid.data_ptr(), dguide.data_ptr() if dgu else None, dinput.data_ptr() if di else None, 1, H, W, GH, GW, GD, 3, 3, 1,
        ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (variant << 8), st)
    assert rc == 0, lib.hdrnet_last_error()
for case in ("g", "gg", "all"):
    for _ in range(30): run(case, 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(case, 0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    trace.zero_()
    lib.hdrnet_tools_set_trace(trace.data_ptr())
    run(case, 9); torch.cuda.synchronize()
    lib.hdrnet_tools_set_trace(None)
    t = trace.cpu().numpy().reshape(NT, NC, 5)
    live = t[:, 0, 0] != 0
    t = t[live]
    full = t[(t[:, :, 0] != 0).all(axis=1)]   # tasks whose wave 0 recorded all NC chunks
    print(f"== case {case}: product {us:.1f} us; traced tasks {live.sum()}, with >= {NC} chunks {len(full)}")
    if len(full) == 0: full = t[:, :8, :][(t[:, :8, 0] != 0).all(axis=1)]
    d = np.diff(full, axis=2).astype(np.float64)        # phases within a chunk
    gap = (full[:, 1:, 0] - full[:, :-1, 4]).astype(np.float64)  # between chunks
    span = (full[:, -1, 4] - full[:, 0, 0]).astype(np.float64)
    names = ["VALU + staging writes", "LDS turnaround + 1st reads", "1st MFMAs issue + 2nd reads", "2nd MFMAs issue + re-zero"]
    tot = d.sum(axis=2).mean() + gap.mean()
    for k, nm in enumerate(names):
        print(f"   {nm:30s} mean {d[:, :, k].mean():8.0f}  p10 {np.percentile(d[:, :, k], 10):7.0f}  p90 {np.percentile(d[:, :, k], 90):7.0f} ticks  ({d[:, :, k].mean() / tot * 100:4.1f} %)")
    print(f"   {'between chunks (loads, row end)':30s} mean {gap.mean():8.0f}  p10 {np.percentile(gap, 10):7.0f}  p90 {np.percentile(gap, 90):7.0f} ticks  ({gap.mean() / tot * 100:4.1f} %)")
    print(f"   per chunk {tot:.0f} ticks; {full.shape[1]} chunks span {span.mean():.0f} ticks")
    # even / odd chunk of a row-batch differ (x-weights, row end): show by chunk ordinal
    print("   per-chunk total by ordinal:", [int(x) for x in (d.sum(axis=2).mean(axis=0))])
    print("   gap before chunk by ordinal:", [int(x) for x in gap.mean(axis=0)])
