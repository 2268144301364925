#!/usr/bin/env python3
"""Interleaved timing of the wire-format / fused-guide forwards of two library builds (4K)."""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
sys.path.insert(0, os.path.join(ROOT, "tools"))
from prev_vs_new import bind

dev = torch.device("cuda:0")
libs = {"prev": bind(os.path.abspath(sys.argv[1])), "new": _lib.load()}
B, H, W, GH, GW, GD = 1, 2160, 3840, 16, 16, 8
gen = torch.Generator(device=dev).manual_seed(1)
nsets = 6
S = [dict(grid=torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen), guide=torch.rand((B, H, W), device=dev, generator=gen),
          inp=torch.rand((B, H, W, 3), device=dev, generator=gen), out=torch.empty((B, H, W, 3), device=dev),
          u8=torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8), o8=torch.empty((B, H, W, 3), device=dev, dtype=torch.uint8))
     for _ in range(nsets)]
conv1 = (torch.randn((16, 4), device=dev, generator=gen) * 0.8).contiguous()
conv2 = (torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous()
stream = torch.cuda.current_stream(dev).cuda_stream

def make(case, lib):
    def fn(k):
        s = S[k % nsets]
        if case == "u8+map->u8":
            rc = lib.hdrnet_bilateral_slice_apply_io(s["grid"].data_ptr(), s["guide"].data_ptr(), s["u8"].data_ptr(), s["o8"].data_ptr(),
                                                     B, H, W, GH, GW, GD, 3, 3, 1, 1, 255.0, 1, None, None, 0, None, stream)
        elif case == "u8->nn->u8":
            rc = lib.hdrnet_bilateral_slice_apply_io(s["grid"].data_ptr(), None, s["u8"].data_ptr(), s["o8"].data_ptr(),
                                                     B, H, W, GH, GW, GD, 3, 3, 1, 1, 255.0, 1, conv1.data_ptr(), conv2.data_ptr(), 16, None, stream)
        elif case == "f32 nnguide":
            rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32(s["grid"].data_ptr(), s["inp"].data_ptr(), conv1.data_ptr(), conv2.data_ptr(),
                                                              s["out"].data_ptr(), None, B, H, W, GH, GW, GD, 3, 3, 1, 16, stream)
        else:
            rc = lib.hdrnet_bilateral_slice_apply_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["out"].data_ptr(),
                                                      B, H, W, GH, GW, GD, 3, 3, 1, stream)
        assert rc == 0
    return fn

def t(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for case in ("fwd", "u8+map->u8", "u8->nn->u8", "f32 nnguide"):
    fns = {w: make(case, l) for w, l in libs.items()}
    res = {"prev": [], "new": []}
    for _ in range(9):
        for w in ("prev", "new"):
            t(fns[w], 30); res[w].append(t(fns[w], 100))
    mp, mn = statistics.median(res["prev"]), statistics.median(res["new"])
    print(f"{case:14s} prev {mp:7.2f} (min {min(res['prev']):7.2f})  new {mn:7.2f} (min {min(res['new']):7.2f})  new/prev {mn/mp:.3f}")
