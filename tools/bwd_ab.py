#!/usr/bin/env python
"""Interleaved A/B of the gradient entry point's kernel variants (tools build), same process.

    python tools/bwd_ab.py [--workload 4k] [--rounds 5] [--steps 50] [--cases all,gg,g] [--variants 0,3,2,4,5]

cases: all = dgrid + dguide + dinput, gg = dgrid + dguide, g = dgrid only, v = dguide + dinput only,
       sl = BilateralSlice grads (C = 12).
variants (include/hdrnet_amd_tools.h): 0 product, 2 bf16-split contraction, 3 un-fused kernels,
4 ABLATION: pixels loaded once per wave (no memory waits), 5 ABLATION: no MFMAs.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS, make_sets  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


def time_launches(fn, steps):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for k in range(steps):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--cases", default="all,gg,g")
    ap.add_argument("--variants", default="0,3,4,5")
    ap.add_argument("--smooth", action="store_true", help="a smooth luminance-like guide (bench.make_sets) instead of U[0,1)")
    ap.add_argument("--guide", default="", choices=["", "const", "smooth0", "photo"],
                    help="const: 0.4 everywhere (every chunk in one plane); smooth0: the smooth guide without its 2 %% noise; "
                         "photo: smooth + 0.5 %% noise (a denoised image)")
    ap.add_argument("--lds-pad", default="", help="comma-separated bytes of unused dynamic LDS per stage-1 workgroup (tools knob 1): "
                    "every (case, variant) is timed at each value, interleaved -- what a resident wave is worth")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load_tools()
    lib.hdrnet_enable_kernel_names(1)
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    Cin, Cout, C = 3, 3, 12
    npx = B * H * W
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // (4 * npx * 11)))
    gen = torch.Generator(device=dev).manual_seed(1)
    S = [dict(grid=torch.rand((B, GH, GW, GD, C), device=dev, generator=gen),
              guide=(make_sets(dev, 1, B, H, W, GH, GW, GD, seed=3 + i, smooth_guide=True)[0][1] if args.smooth
                     else torch.rand((B, H, W), device=dev, generator=gen)),
              inp=torch.rand((B, H, W, Cin), device=dev, generator=gen),
              dout=torch.randn((B, H, W, Cout), device=dev, generator=gen),
              dgrid=torch.empty((B, GH, GW, GD, C), device=dev),
              dguide=torch.empty((B, H, W), device=dev),
              dinput=torch.empty((B, H, W, Cin), device=dev)) for i in range(nsets)]
    if args.guide:
        yy = torch.linspace(0, 1, H, device=dev)[:, None]
        xx = torch.linspace(0, 1, W, device=dev)[None, :]
        for i, s in enumerate(S):
            base = 0.5 + 0.4 * torch.sin(5.0 * xx + 3.0 * yy + i) * torch.cos(2.0 * yy - xx)
            if args.guide == "const":
                s["guide"] = torch.full((B, H, W), 0.4, device=dev)
            elif args.guide == "smooth0":
                s["guide"] = base[None].expand(B, H, W).contiguous()
            else:
                s["guide"] = (base[None] + 0.005 * torch.randn((B, H, W), device=dev, generator=gen)).clamp(0, 1).contiguous()
    sl = [torch.randn((B, H, W, C), device=dev, generator=gen) for _ in range(2)]
    stream = torch.cuda.current_stream(dev).cuda_stream
    wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, Cin, Cout, 1)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    wsb2 = lib.hdrnet_bilateral_slice_grad_workspace_bytes(B, H, W, GH, GW, GD, C)
    ws2 = torch.empty((max(wsb2, 16),), dtype=torch.uint8, device=dev)

    def chk(rc):
        if rc:
            raise RuntimeError(lib.hdrnet_last_error().decode())

    def make(case, variant):
        dg, dgu, di = {"all": (1, 1, 1), "gg": (1, 1, 0), "g": (1, 0, 0), "v": (0, 1, 1), "sl": (1, 1, 0)}[case]

        def fn(k):
            s = S[k % nsets]
            if case == "sl":
                chk(lib.hdrnet_bilateral_slice_grad_f32_ex(
                    s["grid"].data_ptr(), s["guide"].data_ptr(), sl[k % 2].data_ptr(), s["dgrid"].data_ptr(),
                    s["dguide"].data_ptr(), B, H, W, GH, GW, GD, C, ws2.data_ptr(), wsb2,
                    _lib.KERNEL_AUTO | (variant << 8), stream))
            else:
                chk(lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
                    s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
                    s["dgrid"].data_ptr() if dg else None, s["dguide"].data_ptr() if dgu else None,
                    s["dinput"].data_ptr() if di else None, B, H, W, GH, GW, GD, Cin, Cout, 1,
                    ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (variant << 8), stream))
        return fn

    cases = args.cases.split(",")
    variants = [int(v) for v in args.variants.split(",")]
    pads = [int(v) for v in args.lds_pad.split(",")] if args.lds_pad else [None]
    fns, names = {}, {}

    def padded(f, pad):
        if pad is None:
            return f

        def g(k):
            lib.hdrnet_tools_set_knob(1, pad)
            f(k)
            lib.hdrnet_tools_set_knob(1, 0)
        return g

    for c in cases:
        for v in variants:
            for pad in pads:
                try:
                    f = padded(make(c, v), pad)
                    f(0)
                    torch.cuda.synchronize()
                    key = (c, v) if pad is None else (c, v, pad)
                    fns[key] = f
                    names[key] = lib.hdrnet_last_kernel().decode() + ("" if pad is None else f" +{pad} B LDS")
                except Exception as e:  # noqa: BLE001
                    print(f"case {c} variant {v}: unavailable ({e})")
    time_launches(next(iter(fns.values())), 400)  # power-state pre-roll
    res = {k: [] for k in fns}
    for _ in range(args.rounds):
        for k, f in fns.items():
            time_launches(f, 20)  # settle (see tools/ab_bench.py)
            res[k].append(time_launches(f, args.steps))
    print(desc + ("; SMOOTH guide" if args.smooth else "") + (f"; guide {args.guide}" if args.guide else ""))
    for key, t in res.items():
        c, v = key[0], key[1]
        print(f"case {c:4s} variant {v:3d} {names[key]:34s} median {statistics.median(t):8.2f} us  min {min(t):8.2f}"
              f"   all: {[round(x, 1) for x in t]}")


if __name__ == "__main__":
    main()
