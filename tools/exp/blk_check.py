"""Block-form gradient pass (tools variant 11) against the dense pass (variant 0) of the same library and, for dgrid,
against a float64 torch reference on the small shapes: random / smooth / constant / two-plane / edge guides, GD 4..16.
    python tools/exp/blk_check.py [LIB]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
lib = _lib.load_tools()
lib.hdrnet_enable_kernel_names(1)


def guides(B, H, W, GD, gen):
    yy = torch.linspace(0, 1, H, device=dev)[:, None]
    xx = torch.linspace(0, 1, W, device=dev)[None, :]
    smooth = (0.5 + 0.4 * torch.sin(5.0 * xx + 3.0 * yy) * torch.cos(2.0 * yy - xx))[None].expand(B, H, W)
    yield "random", torch.rand((B, H, W), device=dev, generator=gen)
    yield "smooth+2%noise", (smooth + 0.02 * torch.randn((B, H, W), device=dev, generator=gen)).clamp(0, 1).contiguous()
    yield "smooth", smooth.contiguous()
    yield "ramp_x", xx.expand(H, W)[None].expand(B, H, W).contiguous()
    yield "ramp_y_wide", (yy * 1.2 - 0.1).expand(H, W)[None].expand(B, H, W).contiguous()
    for c in (0.0, 0.03, 0.5, 0.97, 1.0, (GD - 0.5) / GD, 0.5 / GD, 1.5 / GD):
        yield f"const {c:.4f}", torch.full((B, H, W), c, device=dev)
    yield "two-plane noise", (0.5 + 0.5 / GD * (torch.rand((B, H, W), device=dev, generator=gen) - 0.5)).contiguous()


bad = 0
for (B, H, W, GH, GW, GD) in [(1, 270, 480, 16, 16, 8), (2, 135, 250, 8, 8, 4), (1, 540, 960, 16, 16, 16), (1, 1080, 1920, 16, 16, 8),
                              (1, 300, 500, 32, 32, 12), (1, 2160, 3840, 16, 16, 8)]:
    gen = torch.Generator(device=dev).manual_seed(5)
    grid = torch.rand((B, GH, GW, GD, 12), device=dev, generator=gen)
    inp = torch.rand((B, H, W, 3), device=dev, generator=gen)
    dout = torch.randn((B, H, W, 3), device=dev, generator=gen)
    wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, 3, 3, 1)
    ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device=dev)
    for gname, guide in guides(B, H, W, GD, gen):
        for subset in ((1, 1, 1), (1, 1, 0), (1, 0, 0)):
            outs = {}
            for v in (0, 11):
                dg = torch.full_like(grid, float("nan")); dgu = torch.full_like(guide, float("nan")); di = torch.full_like(inp, float("nan"))
                rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
                    grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(), dg.data_ptr(),
                    dgu.data_ptr() if subset[1] else None, di.data_ptr() if subset[2] else None, B, H, W, GH, GW, GD, 3, 3, 1,
                    ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (v << 8), stream)
                assert rc == 0, lib.hdrnet_last_error().decode()
                torch.cuda.synchronize()
                outs[v] = (dg, dgu, di, lib.hdrnet_last_kernel().decode())
            a, b = outs[0][0].double(), outs[11][0].double()
            scale = float(a.abs().max())
            e = float((a - b).abs().max()) / scale
            same_px = all(torch.equal(outs[0][k], outs[11][k]) for k in (1, 2) if subset[k])
            flag = "" if (e < 2e-6 and same_px and torch.isfinite(b).all()) else "   <<<<<< BAD"
            bad += bool(flag)
            if flag or subset == (1, 0, 0):
                print(f"{H}x{W} grid {GH}x{GW}x{GD} {gname:18s} {subset} {outs[11][3]:26s} |blk - dense| = {e:.2e} x scale, "
                      f"dguide/dinput identical: {same_px}{flag}")
print("BAD cases:", bad)
