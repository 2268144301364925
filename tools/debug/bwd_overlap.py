#!/usr/bin/env python3
"""Does the grid-gradient kernel (VALU/LDS/MFMA-bound) overlap with the dguide+dinput kernel
(HBM-bound) when the two are enqueued on different streams?  Times serial vs forked."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from hdrnet_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
H, W, GH, GW, GD, Cin, Cout, C = 2160, 3840, 16, 16, 8, 3, 3, 12
gen = torch.Generator(device=dev).manual_seed(1)
nsets = 3
S = [dict(grid=torch.rand((1, GH, GW, GD, C), device=dev, generator=gen),
          guide=torch.rand((1, H, W), device=dev, generator=gen),
          inp=torch.rand((1, H, W, Cin), device=dev, generator=gen),
          dout=torch.randn((1, H, W, Cout), device=dev, generator=gen),
          dgrid=torch.empty((1, GH, GW, GD, C), device=dev),
          dguide=torch.empty((1, H, W), device=dev),
          dinput=torch.empty((1, H, W, Cin), device=dev)) for _ in range(nsets)]
wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(1, H, W, GH, GW, GD, Cin, Cout, 1)
ws = [torch.empty((wsb,), dtype=torch.uint8, device=dev) for _ in range(nsets)]
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(dev)


def call(s, k, dg, rest, stream):
    rc = lib.hdrnet_bilateral_slice_apply_grad_f32(
        s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
        s["dgrid"].data_ptr() if dg else None, s["dguide"].data_ptr() if rest else None,
        s["dinput"].data_ptr() if rest else None, 1, H, W, GH, GW, GD, Cin, Cout, 1,
        ws[k % nsets].data_ptr(), wsb, stream.cuda_stream)
    assert rc == 0, lib.hdrnet_last_error()


def serial(k):
    call(S[k % nsets], k, True, True, main)


def forked(k):
    s = S[k % nsets]
    side.wait_stream(main)
    call(s, k, True, False, side)
    call(s, k, False, True, main)
    main.wait_stream(side)


def timeit(fn, steps=50, rounds=5):
    out = []
    for _ in range(rounds):
        for k in range(3):
            fn(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(steps):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / steps)
    return statistics.median(out), min(out)


for name, fn in (("serial (one call, one stream)", serial), ("forked (dgrid on a side stream)", forked),
                 ("serial again", serial)):
    med, mn = timeit(fn)
    print(f"{name:36s} median {med:7.2f} us  min {mn:7.2f} us")
