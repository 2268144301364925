"""Per-10-launch timeline of a forward flavour right after 4 ms of a DIFFERENT kernel (is the slow mode a ramp?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import WORKLOADS, make_sets
from hdrnet_amd import _lib
lib = _lib.load_tools()
H, W, GH, GW, GD, desc = WORKLOADS['4k']
dev = torch.device('cuda:0')
S = make_sets(dev, 3, H, W, GH, GW, GD, 1)
st = torch.cuda.current_stream(dev).cuda_stream
def launch(v, k):
    g, gu, i, o = S[k % 3]
    rc = lib.hdrnet_bilateral_slice_apply_f32_ex(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(), 1, H, W, GH, GW, GD, 3, 3, 1, _lib.KERNEL_AUTO | (v << 8), st)
    assert rc == 0
def timeline(v, n=200, step=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n // step + 1)]
    ev[0].record()
    for k in range(n):
        launch(v, k)
        if (k + 1) % step == 0: ev[(k + 1) // step].record()
    torch.cuda.synchronize()
    return [round(ev[i].elapsed_time(ev[i + 1]) / step * 1e3, 1) for i in range(n // step)]
for v in (39, 23, 106): launch(v, 0)
torch.cuda.synchronize()
for k in range(1500): launch(39, k)   # pre-roll
torch.cuda.synchronize()
for rep in range(3):
    for other in (19, 106, 28):
        for v in (39, 31, 23):
            for k in range(100): launch(other, k)
            print(f"after 100 x variant {other:3d}: variant {v}: {timeline(v)}", flush=True)
