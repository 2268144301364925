#!/usr/bin/env python3
"""The multi-scale output of HDRNetGaussianPyrNN (hdrnet/models.py:277-289): ONE pass (tools kernel,
csrc/pyramid_onepass.hip) against the product's per-level chain, same inputs, same process, interleaved.

    python tools/pyramid_onepass_bench.py [--workload 4k] [--segs 512,768,1024] [--rounds 7] [--steps 60]

chain   = level 2: guide network + slice-apply; level 1: + up-add of level 2; level 0: + up-add of level 1
          (hdrnet_bilateral_slice_apply_nnguide_f32_ex, 2 x hdrnet_bilateral_slice_apply_upadd_f32_ex): 3 launches,
          the two coarse results in memory -- what models.HDRNetGaussianPyrNN._forward_fused issues after its two
          resizes (which both forms need: the one-pass kernel reads the same down-sampled inputs).
onepass = hdrnet_tools_pyramid_onepass_f32: 1 launch.
Parity first: max|onepass - chain| must be rounding-level.
"""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--segs", default="512,768,1024,1280")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load_tools()
    lib.hdrnet_enable_kernel_names(1)
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    stream = torch.cuda.current_stream(dev).cuda_stream
    gen = torch.Generator(device=dev).manual_seed(3)
    FAST = _lib.GUIDE_SIGMOID_FAST
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // (4 * B * H * W * 8)))

    def chk(rc):
        if rc:
            raise RuntimeError(lib.hdrnet_last_error().decode() or f"rc={rc}")

    sets = []
    for _ in range(nsets):
        full = torch.rand((B, H, W, 3), device=dev, generator=gen)
        half = torch.empty((B, H // 2, W // 2, 3), device=dev)
        quarter = torch.empty((B, H // 4, W // 4, 3), device=dev)
        chk(lib.hdrnet_resize_bilinear_f32(full.data_ptr(), half.data_ptr(), B, H, W, H // 2, W // 2, 3, stream))
        chk(lib.hdrnet_resize_bilinear_f32(half.data_ptr(), quarter.data_ptr(), B, H // 2, W // 2, H // 4, W // 4, 3, stream))
        # an identity-ish affine per level + noise (so that the three levels' contributions are of comparable size)
        grids = []
        for _l in range(3):
            g6 = torch.zeros((B, GH, GW, GD, 3, 4), device=dev)
            for i in range(3):
                g6[..., i, i] = 0.4
            grids.append((g6 + 0.15 * torch.randn(g6.shape, device=dev, generator=gen)).reshape(B, GH, GW, GD, 12).contiguous())
        sets.append(dict(ins=[full, half, quarter], grids=grids,
                         l2=torch.empty((B, H // 4, W // 4, 3), device=dev), l1=torch.empty((B, H // 2, W // 2, 3), device=dev),
                         out_chain=torch.empty((B, H, W, 3), device=dev), out_one=torch.empty((B, H, W, 3), device=dev)))
    conv1 = [(torch.randn((16, 4), device=dev, generator=gen) * 0.8).contiguous() for _ in range(3)]
    conv2 = [(torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous() for _ in range(3)]

    def chain(k):
        s = sets[k % nsets]
        g, i = s["grids"], s["ins"]
        chk(lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(
            g[2].data_ptr(), i[2].data_ptr(), conv1[2].data_ptr(), conv2[2].data_ptr(), s["l2"].data_ptr(), None,
            B, H // 4, W // 4, GH, GW, GD, 3, 3, 1, 16, FAST, stream))
        chk(lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(
            g[1].data_ptr(), None, i[1].data_ptr(), s["l2"].data_ptr(), H // 4, W // 4, s["l1"].data_ptr(),
            B, H // 2, W // 2, GH, GW, GD, 3, 3, 1, conv1[1].data_ptr(), conv2[1].data_ptr(), 16, FAST, stream))
        chk(lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(
            g[0].data_ptr(), None, i[0].data_ptr(), s["l1"].data_ptr(), H // 2, W // 2, s["out_chain"].data_ptr(),
            B, H, W, GH, GW, GD, 3, 3, 1, conv1[0].data_ptr(), conv2[0].data_ptr(), 16, FAST, stream))

    P3 = ctypes.c_void_p * 3
    c1 = P3(*[t.data_ptr() for t in conv1])
    c2 = P3(*[t.data_ptr() for t in conv2])
    bound = [(P3(*[t.data_ptr() for t in s["grids"]]), P3(*[t.data_ptr() for t in s["ins"]])) for s in sets]

    def make_onepass(seg):
        def fn(k):
            s = sets[k % nsets]
            gp, ip = bound[k % nsets]
            chk(lib.hdrnet_tools_pyramid_onepass_f32(gp, ip, c1, c2, 16, s["out_one"].data_ptr(), B, H, W, GH, GW, GD, seg,
                                                     FAST, stream))
        return fn

    def timeit(fn, steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(steps):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps

    segs = [int(v) for v in args.segs.split(",")]
    fns = {"chain (3 launches)": chain}
    chain(0)
    torch.cuda.synchronize()
    print(desc + "; pyramid output, 3 levels, guide networks of 16 features fused; " + f"{nsets} rotating sets")
    for seg in segs:
        f = make_onepass(seg)
        try:
            f(0)
            torch.cuda.synchronize()
        except RuntimeError as e:
            print(f"onepass seg={seg}: unavailable ({e})")
            continue
        a, b2 = sets[0]["out_one"], sets[0]["out_chain"]
        err = float((a - b2).abs().max())
        print(f"onepass seg={seg}: max|onepass - chain| = {err:.3e} on values up to {float(b2.abs().max()):.3g}")
        if not err < 2e-5:
            print("  PARITY FAILED; not timed")
            continue
        fns[f"onepass seg={seg}"] = f
    timeit(chain, 300)  # power-state pre-roll
    res = {k: [] for k in fns}
    for _ in range(args.rounds):
        for k, f in fns.items():
            timeit(f, 20)
            res[k].append(timeit(f, args.steps))
    for k, t in res.items():
        print(f"{k:22s} median {statistics.median(t):8.2f} us  min {min(t):8.2f}   all: {[round(x, 1) for x in t]}")


if __name__ == "__main__":
    main()
