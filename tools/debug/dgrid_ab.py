#!/usr/bin/env python3
"""Times the grid gradient (or, with --all3, the whole BilateralSliceApplyGrad) at 4K through the
C-ABI, after checking it against the generic (bit-exact) kernels.  `--variants` passes flag bits
8..15 through; the backward kernels currently define none, so 0 is the only meaningful value (the
hook stays for the next round of stage-1 experiments)."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hdrnet_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--all3", action="store_true")
ap.add_argument("--smooth", action="store_true")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
dev = torch.device("cuda:0")
lib = _lib.load()
H, W, GH, GW, GD, Cin, Cout, C = 2160, 3840, 16, 16, 8, 3, 3, 12
gen = torch.Generator(device=dev).manual_seed(1)
nsets = 3
S = []
for _ in range(nsets):
    guide = torch.rand((1, H, W), device=dev, generator=gen)
    if args.smooth:
        yy = torch.linspace(0, 1, H, device=dev)[:, None]
        xx = torch.linspace(0, 1, W, device=dev)[None, :]
        guide = (0.5 + 0.5 * torch.sin(6 * xx + 4 * yy) * torch.cos(3 * yy))[None].contiguous()
    S.append(dict(grid=torch.rand((1, GH, GW, GD, C), device=dev, generator=gen), guide=guide,
                  inp=torch.rand((1, H, W, Cin), device=dev, generator=gen),
                  dout=torch.randn((1, H, W, Cout), device=dev, generator=gen),
                  dgrid=torch.empty((1, GH, GW, GD, C), device=dev),
                  dguide=torch.empty((1, H, W), device=dev),
                  dinput=torch.empty((1, H, W, Cin), device=dev)))
wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(1, H, W, GH, GW, GD, Cin, Cout, 1)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream


def call(k, flags):
    s = S[k % nsets]
    rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
        s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
        s["dgrid"].data_ptr(), s["dguide"].data_ptr() if args.all3 else None,
        s["dinput"].data_ptr() if args.all3 else None, 1, H, W, GH, GW, GD, Cin, Cout, 1,
        ws.data_ptr(), wsb, flags, stream)
    assert rc == 0, lib.hdrnet_last_error()


call(0, _lib.KERNEL_GENERIC)
ref = S[0]["dgrid"].clone()
for v in variants:
    S[0]["dgrid"].zero_()
    call(0, _lib.KERNEL_FAST | (v << 8))
    got = S[0]["dgrid"]
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"variant {v}: {lib.hdrnet_last_kernel().decode()}  max|err| / max|ref| = {err:.2e}")
    assert err < 1e-4 or v >= 9

res = {v: [] for v in variants}
for r in range(args.rounds):
    for v in variants:
        fl = _lib.KERNEL_FAST | (v << 8)
        for k in range(3):
            call(k, fl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(args.steps):
            call(k, fl)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) * 1e3 / args.steps)
for v in variants:
    t = res[v]
    print(f"variant {v:2d}: median {statistics.median(t):7.2f} us  min {min(t):7.2f}  all {[round(x, 1) for x in t]}")
