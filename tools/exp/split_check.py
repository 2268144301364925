import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
lib = _lib.load_tools(); dev = torch.device("cuda:0")
stream = torch.cuda.current_stream(dev).cuda_stream
gen = torch.Generator(device=dev).manual_seed(3)
for (B,H,W,GH,GW,GD) in [(1,1080,1920,16,16,8),(2,300,500,16,16,8)]:
    grid = torch.rand((B,GH,GW,GD,12),device=dev,generator=gen); guide = torch.rand((B,H,W),device=dev,generator=gen)
    inp = torch.rand((B,H,W,3),device=dev,generator=gen); dout = torch.randn((B,H,W,3),device=dev,generator=gen)
    wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B,H,W,GH,GW,GD,3,3,1)
    ws = torch.empty((max(wsb,16),),dtype=torch.uint8,device=dev)
    outs = {}
    for v in (0, 2):
        dg, dgu, di = torch.empty_like(grid), torch.empty_like(guide), torch.empty_like(inp)
        rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
            dg.data_ptr(), dgu.data_ptr(), di.data_ptr(), B,H,W,GH,GW,GD,3,3,1, ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (v << 8), stream)
        assert rc == 0, lib.hdrnet_last_error().decode()
        torch.cuda.synchronize(); outs[v] = (dg, dgu, di)
    a, b = outs[0][0].double(), outs[2][0].double()
    print(f"{B}x{H}x{W}: dgrid split vs f32: max|d| = {float((a-b).abs().max()):.3e}, scale {float(a.abs().max()):.3g}, rel-to-scale {float((a-b).abs().max()/a.abs().max()):.2e}, mean signed rel {float(((b-a)/a.abs().clamp_min(1e-6)).mean()):.2e}; dguide equal {torch.equal(outs[0][1], outs[2][1])}")
