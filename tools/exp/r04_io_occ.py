#!/usr/bin/env python3
"""Wire-format kernels (apply_fwd_io) with fewer resident workgroups (tools knob 0 = extra LDS per workgroup), interleaved."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load_tools()
stream = torch.cuda.current_stream(dev).cuda_stream
gen = torch.Generator(device=dev).manual_seed(1)


def run(name, B, H, W, GH, GW, GD, in_dtype, out_u8, pads, nsets=3, steps=150, rounds=7):
    S = []
    for _ in range(nsets):
        hi = 256 if in_dtype == torch.uint8 else 32768
        S.append((torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen), torch.rand((B, H, W), device=dev, generator=gen),
                  torch.randint(0, hi, (B, H, W, 3), device=dev, generator=gen, dtype=torch.int32).to(in_dtype),
                  torch.empty((B, H, W, 3), device=dev, dtype=torch.uint8 if out_u8 else torch.float32)))
    code = 1 if in_dtype == torch.uint8 else 2
    wl = 255.0 if in_dtype == torch.uint8 else 32767.0

    def fn(k):
        g, gu, i, o = S[k % nsets]
        rc = lib.hdrnet_bilateral_slice_apply_io(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(), B, H, W, GH, GW, GD, 3, 3, 1,
                                                 code, wl, 1 if out_u8 else 0, None, None, 0, None, stream)
        assert rc == 0, lib.hdrnet_last_error()

    def t(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(n):
            fn(k)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    res = {p: [] for p in pads}
    for _ in range(rounds):
        for p in pads:
            lib.hdrnet_tools_set_knob(0, p)
            t(40)
            res[p].append(t(steps))
    lib.hdrnet_tools_set_knob(0, 0)
    print(name)
    for p in pads:
        print(f"  +{p:6d} B LDS: median {statistics.median(res[p]):7.2f} us  min {min(res[p]):7.2f}   {[round(x, 1) for x in res[p]]}")


run("u16 / 32767 -> f32 @ 4000x3000 (grid 32x32x8)", 1, 3000, 4000, 32, 32, 8, torch.uint16, False, [0, 3000, 6000, 9000, 13000])
run("u8 -> u8 @ 4K", 1, 2160, 3840, 16, 16, 8, torch.uint8, True, [0, 4000, 8000, 12000, 16000])
