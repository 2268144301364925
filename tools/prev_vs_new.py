#!/usr/bin/env python3
"""Interleaved timing of the gradient (and forward) entry points of TWO builds of the library in one process:
the tree's libhdrnet_amd.so against a previous build given with --prev (e.g. built from the last commit in a
scratch worktree and copied to tools/exp/prev/libhdrnet_amd_prev.so -- *.so files are git-ignored but travel
to the GPU box).  Box-to-box spread is ~5 %, run-to-run drift 1-2 %: a kernel change smaller than that can
only be measured like this.

    python tools/prev_vs_new.py --prev tools/exp/prev/libhdrnet_amd_prev.so [--workload 4k] [--rounds 7]
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


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue  # an older build may lack newer entry points
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prev", required=True)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--cases", default="fwd,all,gg,g,v,slice_fwd,slice_bwd")
    ap.add_argument("--nsets", type=int, default=0, help="buffer sets to rotate over (default: enough to exceed the "
                    "Infinity Cache; 1 = cache-resident, for telling memory time from issue time)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = {"new": _lib.load(), "prev": bind(os.path.abspath(args.prev))}
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    Cin, Cout, C = 3, 3, 12
    npx = B * H * W
    nsets = args.nsets or max(3, -(-int(CACHE_BYTES * 1.5) // (4 * npx * 11)))
    gen = torch.Generator(device=dev).manual_seed(1)
    S = [dict(grid=torch.rand((B, GH, GW, GD, C), device=dev, generator=gen),
              guide=torch.rand((B, H, W), device=dev, generator=gen),
              inp=torch.rand((B, H, W, Cin), device=dev, generator=gen),
              dout=torch.randn((B, H, W, Cout), device=dev, generator=gen),
              out=torch.empty((B, H, W, Cout), device=dev),
              dgrid=torch.empty((B, GH, GW, GD, C), device=dev),
              dguide=torch.empty((B, H, W), device=dev),
              dinput=torch.empty((B, H, W, Cin), device=dev)) for _ in range(nsets)]
    sl = [dict(dout=torch.randn((B, H, W, C), device=dev, generator=gen),
               out=torch.empty((B, H, W, C), device=dev)) for _ in range(2)]
    conv1 = (torch.randn((16, Cin + 1), device=dev, generator=gen) * 0.8).contiguous()
    conv2 = (torch.randn((17,), device=dev, generator=gen) * 0.5).contiguous()
    ccm = (torch.eye(3, 4, device=dev) + 0.2 * torch.randn((3, 4), device=dev, generator=gen)).contiguous()
    shifts = (torch.linspace(0, 1, 17, device=dev)[:16, None].repeat(1, 3)
              + 0.01 * torch.randn((16, 3), device=dev, generator=gen)).contiguous()
    slopes = (0.3 * torch.randn((16, 3), device=dev, generator=gen)).contiguous()
    slopes[0] += 1.0
    mix = torch.tensor([0.4, 0.35, 0.25, 0.02], device=dev)
    u8 = [dict(inp=torch.randint(0, 256, (B, H, W, 3), device=dev, dtype=torch.uint8),
               out=torch.empty((B, H, W, 3), device=dev, dtype=torch.uint8)) for _ in range(nsets)]
    coarse = [torch.randn((B, H // 2, W // 2, 3), device=dev, generator=gen) for _ in range(nsets)]
    stream = torch.cuda.current_stream(dev).cuda_stream
    # The guide-network / curves-guide cases call each build the way its round's models did: a build with the round-5
    # prepare-once helpers gets the guide network's PRESCALED parameters (HDRNET_GUIDE_RELU_PRESCALED: the same bits) and
    # the curves guide's PREPARED tables (the same guide to 1e-6); older builds the exported arrays.
    prepared = {}
    for k, lib in libs.items():
        if hasattr(lib, "hdrnet_guide_nn_prescale_f32") and hasattr(lib, "hdrnet_curves_guide_prepare_f32"):
            p1, p2 = torch.empty_like(conv1), torch.empty_like(conv2)
            assert lib.hdrnet_guide_nn_prescale_f32(conv1.data_ptr(), conv2.data_ptr(), 16, 3, 65536.0, p1.data_ptr(),
                                                    p2.data_ptr(), stream) == 0
            nb = lib.hdrnet_curves_guide_prepared_bytes(3)
            cp = torch.empty((nb // 4,), device=dev)
            if lib.hdrnet_version() >= 241:
                usable = ctypes.c_int(0)
                assert lib.hdrnet_curves_guide_prepare_f32(shifts.data_ptr(), slopes.data_ptr(), 16, 3, cp.data_ptr(), nb,
                                                           ctypes.byref(usable), stream) == 0 and usable.value == 1
            else:  # a mid-round build: the set-up call did not report usability yet (the kernels read the ok word themselves)
                fn = lib.hdrnet_curves_guide_prepare_f32
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                               ctypes.c_void_p]
                assert fn(shifts.data_ptr(), slopes.data_ptr(), 16, 3, cp.data_ptr(), nb, stream) == 0
            prepared[k] = (p1, p2, cp)
    ws, ws2 = {}, {}
    for k, lib in libs.items():
        n = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, Cin, Cout, 1)
        ws[k] = torch.empty((max(n, 16),), dtype=torch.uint8, device=dev)
        n2 = lib.hdrnet_bilateral_slice_grad_workspace_bytes(B, H, W, GH, GW, GD, C)
        ws2[k] = torch.empty((max(n2, 16),), dtype=torch.uint8, device=dev)

    def make(case, which):
        lib = libs[which]
        prep = prepared.get(which)
        nn1, nn2 = (prep[0], prep[1]) if prep else (conv1, conv2)
        nnflags = _lib.GUIDE_SIGMOID_FAST | (_lib.GUIDE_RELU_PRESCALED if prep else 0)

        def chk(rc):
            if rc:
                raise RuntimeError(lib.hdrnet_last_error().decode())

        if case == "fwd":
            def fn(k):
                s = S[k % nsets]
                chk(lib.hdrnet_bilateral_slice_apply_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(),
                                                         s["out"].data_ptr(), B, H, W, GH, GW, GD, Cin, Cout, 1, stream))
            return fn
        if case == "slice_fwd":
            def fn(k):
                s, t = S[k % nsets], sl[k % 2]
                chk(lib.hdrnet_bilateral_slice_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), t["out"].data_ptr(),
                                                   B, H, W, GH, GW, GD, C, stream))
            return fn
        if case == "slice_bwd":
            def fn(k):
                s, t = S[k % nsets], sl[k % 2]
                chk(lib.hdrnet_bilateral_slice_grad_f32(s["grid"].data_ptr(), s["guide"].data_ptr(), t["dout"].data_ptr(),
                                                        s["dgrid"].data_ptr(), s["dguide"].data_ptr(), B, H, W, GH, GW, GD,
                                                        C, ws2[which].data_ptr(), ws2[which].numel(), stream))
            return fn
        if case == "nn":  # guide network fused into the forward
            def fn(k):
                s = S[k % nsets]
                # (a build with the ..._ex twin chooses its sigmoid by flag; older builds: fast when guide_out is NULL)
                if hasattr(lib, "hdrnet_bilateral_slice_apply_nnguide_f32_ex"):
                    chk(lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(
                        s["grid"].data_ptr(), s["inp"].data_ptr(), nn1.data_ptr(), nn2.data_ptr(), s["out"].data_ptr(),
                        None, B, H, W, GH, GW, GD, Cin, Cout, 1, 16, nnflags, stream))
                    return
                chk(lib.hdrnet_bilateral_slice_apply_nnguide_f32(
                    s["grid"].data_ptr(), s["inp"].data_ptr(), conv1.data_ptr(), conv2.data_ptr(), s["out"].data_ptr(),
                    None, B, H, W, GH, GW, GD, Cin, Cout, 1, 16, stream))
            return fn
        if case in ("u8", "u8nn"):  # u8 in -> (guide map | guide network) -> u8 out
            nn = case == "u8nn"

            def fn(k):
                s, t = S[k % nsets], u8[k % nsets]
                if hasattr(lib, "hdrnet_bilateral_slice_apply_io_ex"):
                    chk(lib.hdrnet_bilateral_slice_apply_io_ex(
                        s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), t["inp"].data_ptr(), t["out"].data_ptr(),
                        B, H, W, GH, GW, GD, 3, 3, 1, 1, 255.0, 1, nn1.data_ptr() if nn else None,
                        nn2.data_ptr() if nn else None, 16 if nn else 0, None, nnflags if nn else 0, stream))
                    return
                chk(lib.hdrnet_bilateral_slice_apply_io(
                    s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), t["inp"].data_ptr(), t["out"].data_ptr(),
                    B, H, W, GH, GW, GD, 3, 3, 1, 1, 255.0, 1, conv1.data_ptr() if nn else None,
                    conv2.data_ptr() if nn else None, 16 if nn else 0, None, stream))
            return fn
        if case in ("curves", "u8curves"):  # curves guide fused: f32 -> f32, u8 -> u8
            u = case == "u8curves"

            def fn(k):
                s, t = S[k % nsets], u8[k % nsets]
                if prep:
                    chk(lib.hdrnet_bilateral_slice_apply_io_curves_prepared(
                        s["grid"].data_ptr(), (t if u else s)["inp"].data_ptr(), (t if u else s)["out"].data_ptr(),
                        B, H, W, GH, GW, GD, 3, 3, 1, 1 if u else 0, 255.0 if u else 1.0, 1 if u else 0,
                        ccm.data_ptr(), shifts.data_ptr(), slopes.data_ptr(), mix.data_ptr(), 16, prep[2].data_ptr(), None, stream))
                    return
                chk(lib.hdrnet_bilateral_slice_apply_io_curves(
                    s["grid"].data_ptr(), (t if u else s)["inp"].data_ptr(), (t if u else s)["out"].data_ptr(),
                    B, H, W, GH, GW, GD, 3, 3, 1, 1 if u else 0, 255.0 if u else 1.0, 1 if u else 0,
                    ccm.data_ptr(), shifts.data_ptr(), slopes.data_ptr(), mix.data_ptr(), 16, None, stream))
            return fn
        if case in ("upadd", "nnupadd"):  # one pyramid level: (guide map | guide network) + apply + up-add of the coarser level
            nn = case == "nnupadd"

            def fn(k):
                s = S[k % nsets]
                if hasattr(lib, "hdrnet_bilateral_slice_apply_upadd_f32_ex"):
                    chk(lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(
                        s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), s["inp"].data_ptr(),
                        coarse[k % nsets].data_ptr(), H // 2, W // 2, s["out"].data_ptr(), B, H, W, GH, GW, GD, 3, 3, 1,
                        nn1.data_ptr() if nn else None, nn2.data_ptr() if nn else None, 16 if nn else 0, nnflags if nn else 0,
                        stream))
                    return
                chk(lib.hdrnet_bilateral_slice_apply_upadd_f32(
                    s["grid"].data_ptr(), None if nn else s["guide"].data_ptr(), s["inp"].data_ptr(), coarse[k % nsets].data_ptr(),
                    H // 2, W // 2, s["out"].data_ptr(), B, H, W, GH, GW, GD, 3, 3, 1, conv1.data_ptr() if nn else None,
                    conv2.data_ptr() if nn else None, 16 if nn else 0, stream))
            return fn
        dg, dgu, di = {"all": (1, 1, 1), "gg": (1, 1, 0), "g": (1, 0, 0), "v": (0, 1, 1)}[case]

        def fn(k):
            s = S[k % nsets]
            chk(lib.hdrnet_bilateral_slice_apply_grad_f32(
                s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
                s["dgrid"].data_ptr() if dg else None, s["dguide"].data_ptr() if dgu else None,
                s["dinput"].data_ptr() if di else None, B, H, W, GH, GW, GD, Cin, Cout, 1,
                ws[which].data_ptr(), ws[which].numel(), stream))
        return fn

    def time_launches(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(n):
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    print(f"{desc}; prev = {args.prev}")
    for case in args.cases.split(","):
        fns = {w: make(case, w) for w in ("prev", "new")}
        # same results?  (the new build against the previous one; gradients are deterministic in both)
        outs = {}
        for w in ("prev", "new"):
            for t in ("out", "dgrid", "dguide", "dinput"):
                S[0][t].fill_(0)
            fns[w](0)
            torch.cuda.synchronize()
            outs[w] = [S[0][t].clone() for t in ("out", "dgrid", "dguide", "dinput")]
        diff = max(float((a - b).abs().max()) for a, b in zip(outs["prev"], outs["new"]))
        res = {"prev": [], "new": []}
        for _ in range(args.rounds):
            for w in ("prev", "new"):
                time_launches(fns[w], 20)
                res[w].append(time_launches(fns[w], args.steps))
        mp, mn = statistics.median(res["prev"]), statistics.median(res["new"])
        print(f"{case:10s} prev {mp:7.2f} us (min {min(res['prev']):7.2f})   new {mn:7.2f} us (min {min(res['new']):7.2f})   "
              f"new / prev = {mn / mp:.3f}   max|new - prev| = {diff:.2e}")


if __name__ == "__main__":
    main()
