"""Interleaved timing of gradient-pass variants across several builds of the TOOLS library (one process, one box).
    python tools/exp/split_ab.py name=path[:variant] ...   e.g. f32=lib.so:0 form1=a.so:2 form2=b.so:2"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
from bench import WORKLOADS, CACHE_BYTES

def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    return lib

specs = []
wl = "4k"
for a in sys.argv[1:]:
    if a.startswith("--workload="):
        wl = a.split("=")[1]; continue
    name, rest = a.split("=")
    path, var = rest.rsplit(":", 1)
    specs.append((name, bind(os.path.abspath(path)), int(var)))
dev = torch.device("cuda:0")
B, H, W, GH, GW, GD, desc = WORKLOADS[wl]
npx = B * H * W
nsets = max(3, -(-int(CACHE_BYTES * 1.5) // (4 * npx * 11)))
gen = torch.Generator(device=dev).manual_seed(1)
S = [dict(grid=torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen), guide=torch.rand((B, H, W), device=dev, generator=gen),
          inp=torch.rand((B, H, W, 3), device=dev, generator=gen), dout=torch.randn((B, H, W, 3), device=dev, generator=gen),
          dgrid=torch.empty((B, GH, GW, GD, 12), device=dev), dguide=torch.empty((B, H, W), device=dev),
          dinput=torch.empty((B, H, W, 3), device=dev)) for _ in range(nsets)]
stream = torch.cuda.current_stream(dev).cuda_stream
wsb = max(lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, 3, 3, 1) for _, lib, _ in specs)
ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)

def make(lib, var, case):
    dg, dgu, di = {"all": (1, 1, 1), "gg": (1, 1, 0), "g": (1, 0, 0)}[case]
    def fn(k):
        s = S[k % nsets]
        rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
            s["grid"].data_ptr(), s["guide"].data_ptr(), s["inp"].data_ptr(), s["dout"].data_ptr(),
            s["dgrid"].data_ptr() if dg else None, s["dguide"].data_ptr() if dgu else None, s["dinput"].data_ptr() if di else None,
            B, H, W, GH, GW, GD, 3, 3, 1, ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (var << 8), stream)
        assert rc == 0, lib.hdrnet_last_error().decode()
    return fn

def t(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for k in range(n): fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

print(desc)
for case in ("all", "gg", "g"):
    fns = {name: make(lib, var, case) for name, lib, var in specs}
    ref = None
    for name, f in fns.items():
        f(0); torch.cuda.synchronize()
        d = S[0]["dgrid"].double().clone()
        if ref is None: ref = d
        else:
            print(f"  {case} {name}: dgrid vs {specs[0][0]}: max|d|/scale = {float((d - ref).abs().max() / ref.abs().max()):.2e}, "
                  f"sum(d - ref)/sum|ref| = {float((d - ref).sum() / ref.abs().sum()):+.2e}")
    t(next(iter(fns.values())), 200)
    res = {n: [] for n in fns}
    for _ in range(9):
        for n, f in fns.items():
            t(f, 15); res[n].append(t(f, 50))
    for n, v in res.items():
        print(f"case {case:4s} {n:10s} median {statistics.median(v):8.2f} us  min {min(v):8.2f}")
