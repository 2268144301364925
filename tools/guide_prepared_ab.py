#!/usr/bin/env python3
"""Interleaved A/B of the fused-guide forwards with the exported parameter arrays and with parameters PREPARED once per
parameter set: same process, alternating rounds, rotating buffer sets.

    python tools/guide_prepared_ab.py [--workload 4k] [--rounds 7] [--steps 100]

Guide network (HDRNET_GUIDE_RELU_PRESCALED, hdrnet_guide_nn_prescale_f32): f32 -> guide network -> apply
(apply_fwd_seg<GUIDE_NN>), uint8 -> guide network -> apply -> uint8 (apply_fwd_io), guide network + apply + up-add of the
coarser pyramid level; the two forms give the same bits (checked).  Curves guide (hdrnet_curves_guide_prepare_f32 +
..._io_curves_prepared): f32 -> f32 and uint8 -> uint8; the guide agrees to 1e-6 (max difference of the outputs printed).
"""
import argparse
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
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--only", choices=("exported", "prepared"), default=None,
                    help="launch ONE form only, 20 times per case, no timing: for counter passes (rocprofv3 --pmc), whose "
                         "per-kernel means would otherwise mix the two forms of a kernel that switches at run time")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.hdrnet_enable_kernel_names(1)
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    npx = B * H * W
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // (4 * npx * 6)))
    gen = torch.Generator(device=dev).manual_seed(1)
    S = [dict(grid=torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen),
              inp=torch.rand((B, H, W, 3), device=dev, generator=gen),
              out=torch.empty((B, H, W, 3), device=dev),
              u8=torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8),
              o8=torch.empty((B, H, W, 3), device=dev, dtype=torch.uint8),
              coarse=torch.randn((B, H // 2, W // 2, 3), device=dev, generator=gen)) for _ in range(nsets)]
    conv1 = (torch.randn((16, 4), device=dev, generator=gen) * 0.8).contiguous()
    conv2 = (torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous()
    p1, p2 = torch.empty_like(conv1), torch.empty_like(conv2)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def chk(rc):
        if rc:
            raise RuntimeError(lib.hdrnet_last_error().decode())

    chk(lib.hdrnet_guide_nn_prescale_f32(conv1.data_ptr(), conv2.data_ptr(), 16, 3, 65536.0, p1.data_ptr(), p2.data_ptr(),
                                         stream))
    FAST, PRE = _lib.GUIDE_SIGMOID_FAST, _lib.GUIDE_RELU_PRESCALED

    def par(pre):
        return (p1.data_ptr(), p2.data_ptr(), FAST | PRE) if pre else (conv1.data_ptr(), conv2.data_ptr(), FAST)

    def f32(k, pre):
        s = S[k % nsets]
        c1, c2, fl = par(pre)
        chk(lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(s["grid"].data_ptr(), s["inp"].data_ptr(), c1, c2,
                                                            s["out"].data_ptr(), None, B, H, W, GH, GW, GD, 3, 3, 1, 16, fl,
                                                            stream))

    def u8(k, pre):
        s = S[k % nsets]
        c1, c2, fl = par(pre)
        chk(lib.hdrnet_bilateral_slice_apply_io_ex(s["grid"].data_ptr(), None, s["u8"].data_ptr(), s["o8"].data_ptr(), B, H, W,
                                                   GH, GW, GD, 3, 3, 1, 1, 255.0, 1, c1, c2, 16, None, fl, stream))

    def upadd(k, pre):
        s = S[k % nsets]
        c1, c2, fl = par(pre)
        chk(lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(s["grid"].data_ptr(), None, s["inp"].data_ptr(),
                                                          s["coarse"].data_ptr(), H // 2, W // 2, s["out"].data_ptr(), B, H, W,
                                                          GH, GW, GD, 3, 3, 1, c1, c2, 16, fl, stream))

    ccm = torch.cat([torch.eye(3, device=dev), torch.zeros((3, 1), device=dev)], 1) + 0.1 * torch.randn((3, 4), device=dev, generator=gen)
    shifts = (torch.linspace(0, 1, 17, device=dev)[:-1, None].repeat(1, 3)
              + 0.01 * torch.randn((16, 3), device=dev, generator=gen)).contiguous()
    slopes = (0.2 * torch.randn((16, 3), device=dev, generator=gen)).contiguous()
    mixv = torch.tensor([0.4, 0.35, 0.25, 0.0], device=dev)
    nprep = lib.hdrnet_curves_guide_prepared_bytes(3)
    prep = torch.empty((nprep // 4,), device=dev)
    import ctypes
    usable = ctypes.c_int(0)
    chk(lib.hdrnet_curves_guide_prepare_f32(shifts.data_ptr(), slopes.data_ptr(), 16, 3, prep.data_ptr(), nprep,
                                            ctypes.byref(usable), stream))
    print(f"curves tables prepared: usable = {usable.value}")
    assert usable.value == 1

    def curves(k, pre, u8io):
        s = S[k % nsets]
        chk(lib.hdrnet_bilateral_slice_apply_io_curves_prepared(
            s["grid"].data_ptr(), (s["u8"] if u8io else s["inp"]).data_ptr(), (s["o8"] if u8io else s["out"]).data_ptr(),
            B, H, W, GH, GW, GD, 3, 3, 1, 1 if u8io else 0, 255.0 if u8io else 1.0, 1 if u8io else 0, ccm.data_ptr(),
            shifts.data_ptr(), slopes.data_ptr(), mixv.data_ptr(), 16, prep.data_ptr() if pre else None, None, stream))

    print(f"{desc}; {nsets} rotating sets")
    for name, fn, key in (("f32 -> NN guide -> apply", f32, "out"), ("u8 -> NN guide -> apply -> u8", u8, "o8"),
                          ("NN guide + apply + up-add", upadd, "out"),
                          ("f32 -> curves guide -> apply", lambda k, pre: curves(k, pre, False), "out"),
                          ("u8 -> curves guide -> apply -> u8", lambda k, pre: curves(k, pre, True), "o8")):
        if args.only:
            for k in range(20):
                fn(k, args.only == "prepared")
            torch.cuda.synchronize()
            print(f"{name}: 20 launches, {args.only} form, kernel {lib.hdrnet_last_kernel().decode()}")
            continue
        fn(0, False)
        a = S[0][key].clone()
        fn(0, True)
        torch.cuda.synchronize()
        same = torch.equal(a, S[0][key])
        if not same:
            same = f"no, max |diff| = {float((a.float() - S[0][key].float()).abs().max()):.3g}"
        kern = lib.hdrnet_last_kernel().decode()
        t = {False: [], True: []}
        for k in range(600):  # pre-roll
            fn(k, False)
        for _ in range(args.rounds):
            for pre in (False, True):
                for k in range(5):
                    fn(k, pre)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for k in range(args.steps):
                    fn(k, pre)
                e1.record()
                torch.cuda.synchronize()
                t[pre].append(e0.elapsed_time(e1) * 1e3 / args.steps)
        m0, m1 = statistics.median(t[False]), statistics.median(t[True])
        print(f"{name:34s} {kern:34s} exported {m0:7.2f} us (min {min(t[False]):7.2f})   prepared {m1:7.2f} us "
              f"(min {min(t[True]):7.2f})   x{m1 / m0:.3f}   bit-identical: {same}")


if __name__ == "__main__":
    main()
