#!/usr/bin/env python3
"""In-process interleaved A/B timing of BilateralSliceApply forward kernel variants.

    python tools/ab_bench.py [--workload 4k] [--variants 0,1,2] [--rounds 5] [--steps 100]

Every round times each variant once (HIP events around `steps` back-to-back launches over
rotating buffer sets larger than the Infinity Cache), variants interleaved, and the table
reports median / min of the per-launch time.  Also times a device-to-device copy and a
3-stream elementwise op of the same byte volume as a bandwidth yardstick, and checks each
variant against the generic (bit-exact-to-reference) kernel before timing it.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, WORKLOADS, algorithmic_bytes, make_sets  # noqa: E402
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
    return e0.elapsed_time(e1) * 1e3 / steps  # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--yardstick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    abytes = algorithmic_bytes(1, H, W, GH, GW, GD)
    nsets = max(3, -(-int(CACHE_BYTES * 1.5) // abytes))
    sets = make_sets(dev, nsets, H, W, GH, GW, GD, 1234)
    stream = torch.cuda.current_stream(dev).cuda_stream
    variants = [int(v) for v in args.variants.split(",")]

    def launcher(flags):
        def fn(k):
            grid, guide, inp, out = sets[k % nsets]
            rc = lib.hdrnet_bilateral_slice_apply_f32_ex(
                grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), out.data_ptr(),
                1, H, W, GH, GW, GD, 3, 3, 1, flags, stream)
            if rc:
                raise RuntimeError(lib.hdrnet_last_error().decode())
        return fn

    # correctness of every variant against the generic kernel (same inputs)
    grid, guide, inp, out = sets[0]
    launcher(_lib.KERNEL_GENERIC)(0)
    ref = out.clone()
    names = {}
    for v in variants:
        out.zero_()
        launcher(_lib.KERNEL_FAST | (v << 8))(0)
        torch.cuda.synchronize()
        names[v] = lib.hdrnet_last_kernel().decode()
        err = (out - ref).abs().max().item()
        print(f"variant {v} [{names[v]}]: max|fast - generic| = {err:.3e}")
        assert err < 1e-5 or names[v].startswith("ABLATION"), err

    results = {v: [] for v in variants}
    yard = {"copy(out<-in, 2x100MB)": [], "elementwise(out=in*a+b, in 133MB out 100MB)": []}
    for r in range(args.rounds):
        for v in variants:
            fn = launcher(_lib.KERNEL_FAST | (v << 8))
            time_launches(fn, 10)
            results[v].append(time_launches(fn, args.steps))
        if args.yardstick:
            def cp(k):
                sets[k % nsets][3].copy_(sets[k % nsets][2])
            time_launches(cp, 5)
            yard["copy(out<-in, 2x100MB)"].append(time_launches(cp, args.steps))

            def ew(k):
                g, gu, i, o = sets[k % nsets]
                torch.addcmul(i, i, gu.unsqueeze(-1), out=o)
            time_launches(ew, 5)
            yard["elementwise(out=in*a+b, in 133MB out 100MB)"].append(time_launches(ew, args.steps))

    print(f"\n{desc}; {nsets} rotating sets; algorithmic {abytes / 1e6:.1f} MB/launch")
    for v in variants:
        t = results[v]
        med = statistics.median(t)
        print(f"variant {v:2d} {names[v]:28s} median {med:7.2f} us  min {min(t):7.2f} us  "
              f"-> {abytes / med / 1e3:7.1f} GB/s ({abytes / med / 1e3 / 8000 * 100:4.1f}% of 8 TB/s)  "
              f"{H * W / med:9.0f} MP/s   all: {[round(x, 1) for x in t]}")
    if args.yardstick:
        vol = {"copy(out<-in, 2x100MB)": 2 * 4 * H * W * 3,
               "elementwise(out=in*a+b, in 133MB out 100MB)": 4 * H * W * 7}
        for k, t in yard.items():
            med = statistics.median(t)
            print(f"yardstick {k}: median {med:7.2f} us -> {vol[k] / med / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
