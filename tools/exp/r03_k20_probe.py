#!/usr/bin/env python3
"""Why do the first 20 launches after a synchronize take 45 us in one host flow and 39 us in another?
Per-launch HIP events over the first 24 launches after a sync, for different host-side preambles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from hdrnet_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
B, H, W, GH, GW, GD, _ = bench.WORKLOADS["4k"]
dims = (B, H, W, GH, GW, GD)
sets = bench.make_sets(dev, 3, B, H, W, GH, GW, GD, seed=1)
stream = torch.cuda.current_stream(dev).cuda_stream

def step(n, k0):
    bench.run_steps(lib, sets, dims, stream, n, start=k0)

step(1, 0)
bench.preroll(step, lambda: torch.cuda.synchronize(dev))

def probe(label, pre_sync_sleep=0.0, warm=5, n=24, per_launch=True):
    step(warm, 0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(dev)
    if pre_sync_sleep:
        time.sleep(pre_sync_sleep)
    t0 = time.perf_counter()
    ev[0].record()
    if per_launch:
        for k in range(n):
            step(1, k)
            ev[k + 1].record()
    else:
        step(n, 0)
        ev[n].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    if per_launch:
        us = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
        print(f"{label}: host loop {1e6*(t1-t0):.0f} us; per-launch us: " + " ".join(f"{u:.0f}" for u in us) +
              f" | first20 avg {sum(us[:20])/20:.1f}")
    else:
        print(f"{label}: host loop {1e6*(t1-t0):.0f} us; {n} launches avg {ev[0].elapsed_time(ev[n])*1e3/n:.1f} us")

for rep in range(3):
    probe("per-launch events, no sleep")
    probe("block of 20, no sleep", n=20, per_launch=False)
    probe("block of 20, 200 us idle before", n=20, per_launch=False, pre_sync_sleep=200e-6)
    probe("block of 20, 2 ms idle before", n=20, per_launch=False, pre_sync_sleep=2e-3)
    probe("block of 20, warm 100", n=20, per_launch=False, warm=100)
    probe("block of 200", n=200, per_launch=False)
