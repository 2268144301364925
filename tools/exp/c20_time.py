"""4 -> 4 with offset (C = 20) at 4K: forward, all three gradients (apply_vjp_seg + two channel windows of the MFMA
pass), dgrid alone; against the generic gather on a quarter-size frame.
    python tools/exp/c20_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib, hdrnet_ops as ops
dev = torch.device("cuda:0")
_lib.enable_kernel_names(True)


def run(H, W, GD, need_gu, need_in, override=None, steps=30):
    gen = torch.Generator(device=dev).manual_seed(2)
    grid = torch.rand((1, 16, 16, GD, 20), device=dev, generator=gen).requires_grad_(True)
    guide = torch.rand((1, H, W), device=dev, generator=gen).requires_grad_(need_gu)
    inp = torch.rand((1, H, W, 4), device=dev, generator=gen).requires_grad_(need_in)
    dout = torch.randn((1, H, W, 4), device=dev, generator=gen)
    lib = _lib.load()
    wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(1, H, W, 16, 16, GD, 4, 4, 1)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    dg, dgu, di = torch.empty_like(grid), torch.empty_like(guide), torch.empty_like(inp)
    stream = torch.cuda.current_stream(dev).cuda_stream
    flags = _lib.KERNEL_GENERIC if override == "generic" else _lib.KERNEL_AUTO

    def call():
        rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
            grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(), dg.data_ptr(),
            dgu.data_ptr() if need_gu else None, di.data_ptr() if need_in else None, 1, H, W, 16, 16, GD, 4, 4, 1,
            ws.data_ptr(), wsb, flags, stream)
        assert rc == 0, lib.hdrnet_last_error().decode()
    call(); torch.cuda.synchronize()
    kern = lib.hdrnet_last_kernel().decode()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        call()
    e0.record()
    for _ in range(steps):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps, kern


for GD in (8, 16):
    for nm, gu, gi in (("all three", True, True), ("dgrid + dguide", True, False), ("dgrid", False, False)):
        us, k = run(2160, 3840, GD, gu, gi)
        print(f"4K 4->4+offset GD={GD} {nm:15s} {k:40s} {us:9.1f} us")
us, k = run(1080, 1920, 8, False, False, override="generic", steps=2)
print(f"1080p 4->4+offset GD=8 dgrid           {k:40s} {us:9.1f} us   (the generic gather this shape took before)")
us, k = run(1080, 1920, 8, False, False)
print(f"1080p 4->4+offset GD=8 dgrid           {k:40s} {us:9.1f} us")
