# Probe: what the closing synchronize of a 20-launch region costs, with and without hipDeviceScheduleSpin.
import ctypes, os, sys, time
spin = os.environ.get("SPIN") == "1"
if spin:
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))  # hipDeviceScheduleSpin
    print("hipSetDeviceFlags(spin) rc", rc, file=sys.stderr)
sys.path.insert(0, os.getcwd())
import torch
from bench import WORKLOADS, make_sets
from hdrnet_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, H, W, GH, GW, GD, desc = WORKLOADS["4k"]
S = make_sets(dev, 3, B, H, W, GH, GW, GD, 1234)
stream = torch.cuda.current_stream(dev).cuda_stream
outs = [torch.empty((B, H, W, 3), device=dev) for _ in range(3)]
def launch(k):
    g, gu, i = S[k % 3][:3]
    lib.hdrnet_bilateral_slice_apply_f32(g.data_ptr(), gu.data_ptr(), i.data_ptr(), outs[k % 3].data_ptr(), B, H, W, GH, GW, GD, 3, 3, 1, stream)
for k in range(6000): launch(k)
torch.cuda.synchronize()
res = []
for rep in range(30):
    for k in range(5): launch(k)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    for k in range(20): launch(k)
    th = time.perf_counter()
    ev1.record()
    while not ev1.query(): pass
    tq = time.perf_counter()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    res.append(((t1 - t0) * 1e6 / 20, ev0.elapsed_time(ev1) * 1e3 / 20, (th - t0) * 1e6, (tq - t0) * 1e6, (t1 - tq) * 1e6))
import statistics
for name, idx in (("wall us/launch", 0), ("events us/launch", 1), ("host launch loop us", 2), ("until event fired us", 3), ("closing synchronize us", 4)):
    v = [r[idx] for r in res]
    print(f"spin={int(spin)} {name:26s} median {statistics.median(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f}")
