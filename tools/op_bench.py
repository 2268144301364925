#!/usr/bin/env python3
"""Times every entry point of the C-ABI on one MI355X (HIP events, rotating buffer sets).

    python tools/op_bench.py [--workload 4k|1080p|hdrp|refbench] [--steps 50] [--json out.json] [--tools]

`hdrp` = BASELINE config #5 per GPU (4000x3000, grid 32x32x8x12; also the uint16 / 32767 -> f32 wire
format of hdrnet/data_pipeline.py:267-274); `refbench` = the reference's own micro-benchmark shape
(hdrnet/hdrnet_ops_jax_tf2_test.py:56-65: batch 4, guide 4 x 1024 x 768 (h x w), grid 16 x 12 x 8 (gh x gw x gd), 2 channels, BilateralSlice,
10 burn-in + 100 timed iterations there).  --tools loads the tools build and adds the round-1
kernels (variant 1 of the gradient entry points: dense-tile dgrid) for A/B.

Reports per-launch microseconds and algorithmic GB/s (SURVEY.md section 8d byte counts):
  apply fwd   4*[HW(1+Cin+Cout) + grid]
  apply bwd   reads grid, guide, input, dout; writes dgrid, dguide, dinput
  slice fwd   4*[HW(1+C) + grid]
  slice bwd   reads grid, guide, dout (C ch); writes dgrid, dguide
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


def timeit(fn, steps, rounds=5):
    out = []
    for _ in range(rounds):
        for k in range(3):
            fn(k)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(steps):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / steps)
    return statistics.median(out), min(out)


def smooth_guide(dev, gen, B, H, W):
    """Luminance-like guide: a low-pass ramp + 2 % noise (the form of bench.make_sets(smooth_guide=True))."""
    yy = torch.linspace(0, 1, H, device=dev)[:, None]
    xx = torch.linspace(0, 1, W, device=dev)[None, :]
    base = 0.5 + 0.25 * torch.sin(6.28318 * (xx * 1.5 + yy)) + 0.2 * (xx - 0.5)
    return (base[None] + 0.02 * torch.rand((B, H, W), device=dev, generator=gen)).clamp(0, 1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--json", default=None)
    ap.add_argument("--tools", action="store_true")
    ap.add_argument("--luma-bins", type=int, default=None, help="override the grid depth GD (hdrnet/bin/train.py:235)")
    ap.add_argument("--smooth-guide", action="store_true",
                    help="an image-like guide (bench.make_sets' low-pass ramp + 2 %% noise) instead of U[0, 1)")
    ap.add_argument("--only", default=None, help="comma-separated substrings: run only the ops whose name contains one")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load_tools() if args.tools else _lib.load()
    lib.hdrnet_enable_kernel_names(1)
    if args.workload == "refbench":
        return refbench(lib, dev, args)
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    if args.luma_bins:
        desc = desc.replace(f"x{GD}x12", f"x{args.luma_bins}x12")
        GD = args.luma_bins
    if args.smooth_guide:
        desc += ", smooth (image-like) guide"
    Cin, Cout, C = 3, 3, 12
    npx = B * H * W
    gridb = 4 * B * GH * GW * GD * C
    # enough buffer sets that even the op with the smallest footprint per set -- the forward, 28 B/px -- cycles through
    # 1.5 x the Infinity Cache (bench.py's rule; until round 6 the count was taken from the backward's 44 B/px, which at
    # 1080p left the forward 290 MB for a 256-MB cache and read 10.9 us where bench.py's seven sets read 11.6-12.0)
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // (4 * npx * 7)))
    gen = torch.Generator(device=dev).manual_seed(1)
    S = []
    for _ in range(nsets):
        S.append(dict(
            grid=torch.rand((B, GH, GW, GD, C), device=dev, generator=gen),
            guide=smooth_guide(dev, gen, B, H, W) if args.smooth_guide else torch.rand((B, H, W), device=dev, generator=gen),
            inp=torch.rand((B, H, W, Cin), device=dev, generator=gen),
            dout=torch.randn((B, H, W, Cout), device=dev, generator=gen),
            out=torch.empty((B, H, W, Cout), device=dev),
            dgrid=torch.empty((B, GH, GW, GD, C), device=dev),
            dguide=torch.empty((B, H, W), device=dev),
            dinput=torch.empty((B, H, W, Cin), device=dev)))
    # the un-fused slice moves 4*C B/px: two sets suffice to exceed the cache
    sl = [dict(dout=torch.randn((B, H, W, C), device=dev, generator=gen),
               out=torch.empty((B, H, W, C), device=dev)) for _ in range(2)]
    stream = torch.cuda.current_stream(dev).cuda_stream
    wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, Cin, Cout, 1)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    wsb2 = lib.hdrnet_bilateral_slice_grad_workspace_bytes(B, H, W, GH, GW, GD, C)
    ws2 = torch.empty((max(wsb2, 16),), dtype=torch.uint8, device=dev)

    def chk(rc):
        if rc:
            raise RuntimeError(lib.hdrnet_last_error().decode())

    def apply_fwd(k):
        s = S[k % nsets]
        chk(lib.hdrnet_bilateral_slice_apply_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(),
                                                 s["out"].data_ptr(), B, H, W, GH, GW, GD, Cin, Cout, 1, stream))

    conv1 = (torch.randn((16, Cin + 1), device=dev, generator=gen) * 0.8).contiguous()
    conv2 = (torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous()

    FAST = _lib.GUIDE_SIGMOID_FAST  # the inference lines: the models' explicit choice (HDRNET_GUIDE_SIGMOID_FAST)
    # ... and the guide network's parameters in their PRESCALED form (HDRNET_GUIDE_RELU_PRESCALED, what the models' inference
    # passes since round 5: the same guide bit for bit); `pre=False` lines time the exported layout
    PRE = _lib.GUIDE_RELU_PRESCALED
    pconv1, pconv2 = torch.empty_like(conv1), torch.empty_like(conv2)
    chk(lib.hdrnet_guide_nn_prescale_f32(conv1.data_ptr(), conv2.data_ptr(), 16, 3, 65536.0, pconv1.data_ptr(),
                                         pconv2.data_ptr(), stream))

    def nnp(pre):
        return (pconv1.data_ptr(), pconv2.data_ptr(), FAST | PRE) if pre else (conv1.data_ptr(), conv2.data_ptr(), FAST)

    def apply_fwd_nnguide(k, pre=True):
        s = S[k % nsets]
        c1, c2, fl = nnp(pre)
        chk(lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(
            s["grid"].data_ptr(), s["inp"].data_ptr(), c1, c2, s["out"].data_ptr(),
            None, B, H, W, GH, GW, GD, Cin, Cout, 1, 16, fl, stream))

    u8 = [dict(inp=torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8),
               out=torch.empty((B, H, W, 3), device=dev, dtype=torch.uint8)) for _ in range(nsets)]

    def apply_io_u8(k, nn=True, pre=True):
        s, t = S[k % nsets], u8[k % nsets]
        c1, c2, fl = nnp(pre)
        chk(lib.hdrnet_bilateral_slice_apply_io_ex(
            s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), t["inp"].data_ptr(), t["out"].data_ptr(),
            B, H, W, GH, GW, GD, 3, 3, 1, 1, 255.0, 1, c1 if nn else None,
            c2 if nn else None, 16 if nn else 0, None, fl if nn else 0, stream))

    ccm = torch.cat([torch.eye(3, device=dev), torch.zeros((3, 1), device=dev)], 1) + 0.1 * torch.randn((3, 4), device=dev, generator=gen)
    shifts = torch.linspace(0, 1, 17, device=dev)[:-1, None].repeat(1, 3).contiguous()
    slopes = (0.2 * torch.randn((16, 3), device=dev, generator=gen)).contiguous()
    mixv = torch.tensor([0.4, 0.35, 0.25, 0.0], device=dev)

    # the curves' lookup tables prepared once per parameter set (hdrnet_curves_guide_prepare_f32; what the models' inference
    # passes since round 5); `pre=False` lines time the exported arrays alone (every workgroup sorts the knots itself)
    nprep = lib.hdrnet_curves_guide_prepared_bytes(3)
    cprep = torch.empty((nprep // 4,), device=dev)
    import ctypes
    usable = ctypes.c_int(0)
    chk(lib.hdrnet_curves_guide_prepare_f32(shifts.data_ptr(), slopes.data_ptr(), 16, 3, cprep.data_ptr(), nprep,
                                            ctypes.byref(usable), stream))
    assert usable.value == 1, "the bench's equidistant knots fit the cell tables"

    def apply_io_curves(k, u8io=True, pre=True):
        s, t = S[k % nsets], u8[k % nsets]
        chk(lib.hdrnet_bilateral_slice_apply_io_curves_prepared(
            s["grid"].data_ptr(), (t["inp"] if u8io else s["inp"]).data_ptr(), (t["out"] if u8io else s["out"]).data_ptr(),
            B, H, W, GH, GW, GD, 3, 3, 1, 1 if u8io else 0, 255.0 if u8io else 1.0, 1 if u8io else 0,
            ccm.data_ptr(), shifts.data_ptr(), slopes.data_ptr(), mixv.data_ptr(), 16, cprep.data_ptr() if pre else None,
            None, stream))

    coarse = [torch.randn((B, H // 2, W // 2, 3), device=dev, generator=gen) for _ in range(nsets)]
    half = [torch.empty((B, H // 2, W // 2, 3), device=dev) for _ in range(nsets)]

    def apply_upadd(k, nn=True, pre=True):
        s = S[k % nsets]
        c1, c2, fl = nnp(pre)
        chk(lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(
            s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), s["inp"].data_ptr(),
            coarse[k % nsets].data_ptr(), H // 2, W // 2, s["out"].data_ptr(), B, H, W, GH, GW, GD, 3, 3, 1,
            c1 if nn else None, c2 if nn else None, 16 if nn else 0, fl if nn else 0,
            stream))

    def resize_half(k):
        chk(lib.hdrnet_resize_bilinear_f32(S[k % nsets]["inp"].data_ptr(), half[k % nsets].data_ptr(),
                                           B, H, W, H // 2, W // 2, 3, stream))

    def apply_bwd(k, dg=True, dgu=True, di=True, variant=0):
        s = S[k % nsets]
        chk(lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
            s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
            s["dgrid"].data_ptr() if dg else None, s["dguide"].data_ptr() if dgu else None,
            s["dinput"].data_ptr() if di else None, B, H, W, GH, GW, GD, Cin, Cout, 1,
            ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (variant << 8), stream))

    u16 = None
    if args.workload == "hdrp":
        u16 = [torch.randint(0, 32768, (B, H, W, 3), device=dev, dtype=torch.int32).to(torch.uint16)
               for _ in range(nsets)]

    def apply_io_u16(k):
        s = S[k % nsets]
        chk(lib.hdrnet_bilateral_slice_apply_io(
            s["grid"].data_ptr(), s["guide"].data_ptr(), u16[k % nsets].data_ptr(), s["out"].data_ptr(),
            B, H, W, GH, GW, GD, 3, 3, 1, 2, 32767.0, 0, None, None, 0, None, stream))

    def slice_fwd(k):
        s, t = S[k % nsets], sl[k % 2]
        chk(lib.hdrnet_bilateral_slice_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), t["out"].data_ptr(),
                                           B, H, W, GH, GW, GD, C, stream))

    def slice_bwd(k):
        s, t = S[k % nsets], sl[k % 2]
        chk(lib.hdrnet_bilateral_slice_grad_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), t["dout"].data_ptr(),
                                                s["dgrid"].data_ptr(), s["dguide"].data_ptr(),
                                                B, H, W, GH, GW, GD, C, ws2.data_ptr(), wsb2, stream))

    rows = []

    only = [t.strip() for t in args.only.split(",")] if args.only else None

    def run(name, fn, nbytes):
        if only and not any(t in name for t in only):
            return
        fn(0)
        torch.cuda.synchronize()
        kern = lib.hdrnet_last_kernel().decode()
        med, mn = timeit(fn, args.steps)
        rows.append(dict(op=name, kernel=kern, us=round(med, 2), us_min=round(mn, 2),
                         algorithmic_MB=round(nbytes / 1e6, 1), GBps=round(nbytes / med / 1e3, 1),
                         hbm_frac=round(nbytes / med / 1e3 / 8000, 4), MPps=round(npx / med, 0)))
        print(f"{name:28s} {kern:40s} {med:8.2f} us (min {mn:7.2f})  {nbytes / 1e6:7.1f} MB  "
              f"{nbytes / med / 1e3:7.1f} GB/s  {nbytes / med / 1e3 / 80:5.1f}% of 8 TB/s")

    print(f"{desc}; {nsets} rotating sets; workspace apply-grad {wsb / 1e6:.1f} MB")
    for k in range(1500):  # pre-roll: ~60 ms of launches take the device out of the idle power state
        apply_fwd(k)
    torch.cuda.synchronize()
    run("apply fwd", apply_fwd, 4 * npx * (1 + Cin + Cout) + gridb)
    if u16 is not None:
        run("u16 / 32767 + guide map -> apply -> f32", apply_io_u16, npx * (4 + 6 + 12) + gridb)
    run("guide-NN(16) + apply fwd fused", apply_fwd_nnguide, 4 * npx * (Cin + Cout) + gridb)
    run("... exported layout (no prescale)", lambda k: apply_fwd_nnguide(k, pre=False), 4 * npx * (Cin + Cout) + gridb)
    run("u8 -> guide-NN + apply -> u8", apply_io_u8, npx * 6 + gridb)
    run("... exported layout (no prescale)", lambda k: apply_io_u8(k, pre=False), npx * 6 + gridb)
    run("u8 + guide map -> apply -> u8", lambda k: apply_io_u8(k, nn=False), npx * 10 + gridb)
    run("curves guide + apply fwd fused", lambda k: apply_io_curves(k, u8io=False), 4 * npx * (Cin + Cout) + gridb)
    run("... exported arrays only (no prepared tables)", lambda k: apply_io_curves(k, u8io=False, pre=False),
        4 * npx * (Cin + Cout) + gridb)
    run("u8 -> curves guide + apply -> u8", apply_io_curves, npx * 6 + gridb)
    run("... exported arrays only (no prepared tables)", lambda k: apply_io_curves(k, pre=False), npx * 6 + gridb)
    run("apply + up-add of coarse level", lambda k: apply_upadd(k, nn=False),
        4 * npx * (1 + Cin + Cout) + gridb + 4 * npx * 3 // 4)
    run("guide-NN + apply + up-add", apply_upadd, 4 * npx * (Cin + Cout) + gridb + 4 * npx * 3 // 4)
    run("... exported layout (no prescale)", lambda k: apply_upadd(k, pre=False),
        4 * npx * (Cin + Cout) + gridb + 4 * npx * 3 // 4)
    run("resize bilinear 4K -> 1080p", resize_half, 4 * npx * 3 + 4 * npx * 3 // 4)
    run("apply bwd (all three)", apply_bwd, 4 * npx * (1 + Cin + Cout) + 4 * npx * (1 + Cin) + 2 * gridb)
    run("apply bwd dguide+dinput", lambda k: apply_bwd(k, dg=False), 4 * npx * (1 + Cin + Cout) + 4 * npx * (1 + Cin) + gridb)
    run("apply bwd dgrid only", lambda k: apply_bwd(k, dgu=False, di=False), 4 * npx * (1 + Cin + Cout) + gridb)
    run("apply bwd dgrid+dguide", lambda k: apply_bwd(k, di=False), 4 * npx * (1 + Cin + Cout) + 4 * npx + 2 * gridb)
    if args.tools:
        run("apply bwd (all three), un-fused kernels", lambda k: apply_bwd(k, variant=3),
            4 * npx * (1 + Cin + Cout) + 4 * npx * (1 + Cin) + 2 * gridb)
        run("apply bwd (all three), bf16-split MFMA", lambda k: apply_bwd(k, variant=2),
            4 * npx * (1 + Cin + Cout) + 4 * npx * (1 + Cin) + 2 * gridb)
        run("apply bwd dgrid+dguide, bf16-split MFMA", lambda k: apply_bwd(k, di=False, variant=2),
            4 * npx * (1 + Cin + Cout) + 4 * npx + 2 * gridb)
        run("apply bwd dgrid only (bf16-split MFMA)", lambda k: apply_bwd(k, dgu=False, di=False, variant=2),
            4 * npx * (1 + Cin + Cout) + gridb)
    run("slice fwd", slice_fwd, 4 * npx * (1 + C) + gridb)
    run("slice bwd (both)", slice_bwd, 4 * npx * (1 + C) + 4 * npx + 2 * gridb)
    if args.json:
        json.dump(dict(workload=desc, rows=rows), open(args.json, "w"), indent=1)


def refbench(lib, dev, args):
    """BilateralSlice at the reference's micro-benchmark shape, its iteration counts."""
    B, H, W, GH, GW, GD, C = 4, 1024, 768, 16, 12, 8, 2  # guide (4, 1024, 768), grid (4, 16, 12, 8, 2)
    gen = torch.Generator(device=dev).manual_seed(1)
    nsets = 16  # 4 * 768 * 1024 * (1 + 2) * 4 B = 37.7 MB per set
    S = [dict(grid=torch.rand((B, GH, GW, GD, C), device=dev, generator=gen),
              guide=torch.rand((B, H, W), device=dev, generator=gen),
              out=torch.empty((B, H, W, C), device=dev)) for _ in range(nsets)]
    stream = torch.cuda.current_stream(dev).cuda_stream

    def slice_fwd(k):
        s = S[k % nsets]
        rc = lib.hdrnet_bilateral_slice_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), s["out"].data_ptr(),
                                            B, H, W, GH, GW, GD, C, stream)
        if rc:
            raise RuntimeError(lib.hdrnet_last_error().decode())

    slice_fwd(0)
    torch.cuda.synchronize()
    kern = lib.hdrnet_last_kernel().decode()
    for k in range(10):  # the reference's burn-in
        slice_fwd(k)
    med, mn = timeit(slice_fwd, 100, rounds=5)
    nbytes = 4 * B * (H * W * (1 + C) + GH * GW * GD * C)
    print(f"BilateralSlice fwd, reference micro-benchmark shape (batch {B}, {W}x{H}, grid {GW}x{GH}x{GD}x{C}); "
          f"{nsets} rotating sets\n{'slice fwd':28s} {kern:40s} {med:8.2f} us (min {mn:7.2f})  {nbytes / 1e6:7.1f} MB  "
          f"{nbytes / med / 1e3:7.1f} GB/s  {nbytes / med / 1e3 / 80:5.1f}% of 8 TB/s   {B * H * W / med:9.0f} MP/s")
    if args.json:
        json.dump(dict(workload="refbench", rows=[dict(op="slice fwd", kernel=kern, us=round(med, 2), us_min=round(mn, 2),
                                                       GBps=round(nbytes / med / 1e3, 1))]), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
