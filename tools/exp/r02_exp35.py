"""Per-100-launch timeline of the product forward from a cold start (when does a slow box slow down?)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import WORKLOADS, make_sets
from hdrnet_amd import _lib
lib = _lib.load()
H, W, GH, GW, GD, desc = WORKLOADS['4k']
dev = torch.device('cuda:0')
S = make_sets(dev, 3, H, W, GH, GW, GD, 1)
st = torch.cuda.current_stream(dev).cuda_stream
torch.cuda.synchronize()
time.sleep(1.0)   # idle
N, step = 12000, 100
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N // step + 1)]
ev[0].record()
for k in range(N):
    g, gu, i, o = S[k % 3]
    lib.hdrnet_bilateral_slice_apply_f32(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(), 1, H, W, GH, GW, GD, 3, 3, 1, st)
    if (k + 1) % step == 0: ev[(k + 1) // step].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) / step * 1e3 for i in range(N // step)]
cum = 0.0
out = []
for i, x in enumerate(t):
    cum += x * step / 1e3
    out.append(f"{cum:6.0f}ms:{x:5.1f}")
print("us per launch over windows of 100 launches (cumulative ms : us):")
for r in range(0, len(out), 10): print("  " + "  ".join(out[r:r + 10]))
print(subprocess.run("rocm-smi --showclocks --showpower 2>&1 | grep -i 'sclk\\|Power (W)' | tr -s ' ' | cut -c1-80", shell=True, capture_output=True, text=True).stdout)
