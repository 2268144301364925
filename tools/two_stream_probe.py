#!/usr/bin/env python3
"""Diagnostic: independent frames issued round-robin on N HIP streams instead of one.

A launch of the 4K forward spends ~5 us filling the memory pipeline and ~6 us draining it (profiles/r03/
ab_variants_4k.txt, the timeline trace); on ONE stream consecutive launches are serialised by the queue's barrier
bit, so that cost is paid per frame.  Frames of a serving pipeline are independent: issued on two streams, the next
frame's fill runs under the previous frame's drain.  This is not what bench.py times (one stream, the op's plain
semantics); it is a number for deployments.

    python tools/two_stream_probe.py [--workload 4k] [--streams 1,2,3] [--launches 2000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import CACHE_BYTES, HBM_PEAK_GBPS, WORKLOADS, algorithmic_bytes, make_sets  # noqa: E402
from hdrnet_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k")
    ap.add_argument("--streams", default="1,2,3")
    ap.add_argument("--launches", type=int, default=2000)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, H, W, GH, GW, GD, desc = WORKLOADS[args.workload]
    abytes = algorithmic_bytes(B, H, W, GH, GW, GD)
    nsets = max(6, -(-int(CACHE_BYTES * 1.5) // abytes))  # >= 6: every stream works on its own buffers
    sets = make_sets(dev, nsets, B, H, W, GH, GW, GD, 1234)
    print(desc)
    for ns in [int(x) for x in args.streams.split(",")]:
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        handles = [s.cuda_stream for s in streams]

        def run(n):
            for k in range(n):
                grid, guide, inp, out = sets[k % nsets]
                rc = lib.hdrnet_bilateral_slice_apply_f32(grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), out.data_ptr(),
                                                          B, H, W, GH, GW, GD, 3, 3, 1, handles[k % ns])
                if rc:
                    raise RuntimeError(lib.hdrnet_last_error().decode())

        run(2000)
        torch.cuda.synchronize()
        res = []
        for _ in range(args.rounds):
            t0 = time.perf_counter()
            run(args.launches)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / args.launches * 1e6)
        us = min(res)
        print(f"{ns} stream(s): {us:6.2f} us per frame (rounds: {', '.join(f'{r:.2f}' for r in res)})  ->  "
              f"{abytes / us / 1e3:.0f} GB/s algorithmic = {abytes / us / 1e3 / HBM_PEAK_GBPS:.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
