#!/usr/bin/env python3
"""Old (round 2) vs new bench.py run_steps / timed on the same buffers in one process, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as new
import _bench_r02_tmp as old
from hdrnet_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
B, H, W, GH, GW, GD, _ = new.WORKLOADS["4k"]
sets = new.make_sets(dev, 3, B, H, W, GH, GW, GD, seed=1)
stream = torch.cuda.current_stream(dev).cuda_stream
dims_new = (B, H, W, GH, GW, GD)
dims_old = (H, W, GH, GW, GD)

def step(n, k0):
    new.run_steps(lib, sets, dims_new, stream, n, start=k0)

step(1, 0)
new.preroll(step, lambda: torch.cuda.synchronize(dev))
for rep in range(4):
    old.run_steps(lib, sets, dims_old, stream, 5)
    w, g = old.timed(lib, sets, dims_old, stream, 20, False, dev)
    print(f"old run_steps/timed: events {g/20*1e6:.1f} us wall {w/20*1e6:.1f} us")
    step(5, 0)
    w, g = new.timed(step, 20, False, dev)
    print(f"new run_steps/timed: events {g/20*1e6:.1f} us wall {w/20*1e6:.1f} us")
    # new timed with old run_steps
    old.run_steps(lib, sets, dims_old, stream, 5)
    w, g = new.timed(lambda n, k0: old.run_steps(lib, sets, dims_old, stream, n, start=k0), 20, False, dev)
    print(f"new timed + old run_steps: events {g/20*1e6:.1f} us wall {w/20*1e6:.1f} us")
