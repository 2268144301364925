"""bf16-split contraction (tools variant 2) against the exact-f32 pass of the same library, for a list of
libraries: which build computes what, on which guide range, which gradient subset.
    python tools/exp/split_check2.py LIB [LIB ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    return lib


for path in sys.argv[1:]:
    lib = bind(path)
    for (B, H, W, lo, hi, dscale) in [(1, 1080, 1920, 0.0, 1.0, 1.0), (1, 1080, 1920, -0.02, 1.02, 1.0), (1, 270, 480, -0.02, 1.02, 1.0),
                                      (1, 2160, 3840, 0.0, 1.0, 1.0), (1, 1080, 1920, 0.0, 1.0, 1e-7), (1, 1080, 1920, 0.0, 1.0, 1e4)]:
        gen = torch.Generator(device=dev).manual_seed(3)
        GH = GW = 16; GD = 8
        grid = torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen)
        guide = torch.rand((B, H, W), device=dev, generator=gen) * (hi - lo) + lo
        inp = torch.rand((B, H, W, 3), device=dev, generator=gen)
        dout = torch.randn((B, H, W, 3), device=dev, generator=gen) * dscale  # 1e-7: dout of a mean loss over 2 M pixels
        wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, 3, 3, 1)
        ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
        for subset in ((1, 1, 1), (1, 1, 0), (1, 0, 0)):
            outs = {}
            for v in (0, 2, 10):
                dg, dgu, di = torch.empty_like(grid), torch.empty_like(guide), torch.empty_like(inp)
                rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
                    grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(), dg.data_ptr(),
                    dgu.data_ptr() if subset[1] else None, di.data_ptr() if subset[2] else None, B, H, W, GH, GW, GD, 3, 3, 1,
                    ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (v << 8), stream)
                if rc != 0:
                    continue  # an older library without this variant
                torch.cuda.synchronize()
                outs[v] = dg
            for v in (2, 10):
                if v not in outs:
                    continue
                a, b = outs[0].double(), outs[v].double()
                d = (a - b).abs()
                bad = (d > 1e-4 * a.abs().max()).nonzero()
                print(f"{os.path.basename(path)} {H}x{W} guide [{lo}, {hi}] dout x {dscale:g} subset {subset} variant {v}: rel-to-scale "
                      f"{float(d.max() / a.abs().max()):.2e}; cells off by > 1e-4 scale: {len(bad)}"
                      + (f", first {bad[:4].tolist()}" if len(bad) else ""))
