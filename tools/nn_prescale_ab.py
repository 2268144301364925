#!/usr/bin/env python3
"""Interleaved A/B of the fused guide-network forwards with the exported parameter layout and with the PRESCALED one
(HDRNET_GUIDE_RELU_PRESCALED, include/hdrnet_amd.h): same process, alternating rounds, rotating buffer sets.

    python tools/nn_prescale_ab.py [--workload 4k] [--rounds 7] [--steps 100]

Cases: f32 -> guide network -> apply (apply_fwd_seg<GUIDE_NN>), uint8 -> guide network -> apply -> uint8 (apply_fwd_io),
guide network + apply + up-add of the coarser pyramid level.  Checks first that the two forms give the same bits.
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

    print(f"{desc}; {nsets} rotating sets")
    for name, fn, key in (("f32 -> NN guide -> apply", f32, "out"), ("u8 -> NN guide -> apply -> u8", u8, "o8"),
                          ("NN guide + apply + up-add", upadd, "out")):
        fn(0, False)
        a = S[0][key].clone()
        fn(0, True)
        torch.cuda.synchronize()
        same = torch.equal(a, S[0][key])
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
        print(f"{name:32s} {kern:34s} exported {m0:7.2f} us (min {min(t[False]):7.2f})   prescaled {m1:7.2f} us "
              f"(min {min(t[True]):7.2f})   x{m1 / m0:.3f}   bit-identical: {same}")


if __name__ == "__main__":
    main()
