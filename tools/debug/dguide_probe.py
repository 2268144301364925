import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
sys.path.insert(0, "tests")
from test_gpu_parity import APPLY_SHAPES, rand_case, T, N
from hdrnet_amd import hdrnet_ops as ops
dev = torch.device("cuda:0")
shape = APPLY_SHAPES[3]
B, H, W, GH, GW, GD, Cin, Cout, ho, lo, hi = shape
rng = np.random.default_rng(abs(hash(shape)) % (2 ** 31) + 1)
grid, guide, inp, dout = rand_case(rng, B, H, W, GH, GW, GD, Cin, Cout, ho, lo, hi)
res = {}
for which in ("generic", "auto"):
    tg = T(grid, dev).requires_grad_(True); tgu = T(guide, dev).requires_grad_(True); ti = T(inp, dev).requires_grad_(True)
    with ops.kernel_override(which):
        ops.bilateral_slice_apply(tg, tgu, ti, has_offset=ho).backward(T(dout, dev))
    res[which] = N(tgu.grad)
d = np.abs(res["auto"] - res["generic"])
idx = np.unravel_index(np.argsort(d.ravel())[-5:], d.shape)
for b, y, x in zip(*idx):
    g = guide[b, y, x]; gzf = np.float32(g) * np.float32(GD)
    print((b, y, x), "diff", d[b, y, x], "fast", res["auto"][b, y, x], "generic", res["generic"][b, y, x],
          "guide", repr(g), "gzf", repr(gzf), "frac(gzf-0.5)", float(gzf - 0.5 - np.floor(gzf - 0.5)))
