"""Generates the oracle side of tests/test_gpu_fullsize.py::test_optimize_both_reference: the reference's
test_optimize_both (hdrnet/test/ops_test.py:280-322: 10 000 gradient-descent steps, lr 1e-1, on grid AND
guide logits) run on the CPU oracle for several np.random seeds.  The reference does not seed its data,
and whether its `SSE < 1e-4` holds depends on the draw:

    seed 1234 -> 4.2299519e-04   seed 0 -> 1.587e-04   seed 1 -> 2.355e-03
    seed 2    -> 1.6063e-05      seed 3 -> 3.7599e-06  seed 4 -> 4.686e-04

    python tests/golden/optimize_both_oracle.py
"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import oracle
P = oracle.port()
def run(seed):
    np.random.seed(seed)
    w, gw, gd = 32, 8, 2
    logits = (np.random.rand(1, 1, w).astype(np.float32) * 2.0 - 1.0)
    grid = np.random.rand(1, 1, gw, gd, 1).astype(np.float32)
    target = np.sin(np.linspace(0, 2*np.pi, w)).astype(np.float32)[None, None, :, None]
    lr = np.float32(0.1)
    for step in range(10000):
        guide = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
        out = P.bilateral_slice(grid, guide)
        dout = (2.0 * (out - target)).astype(np.float32)
        dgrid, dguide = P.bilateral_slice_grad(grid, guide, dout)
        dlog = (dguide * guide * (1 - guide)).astype(np.float32)
        grid = (grid - lr * dgrid).astype(np.float32)
        logits = (logits - lr * dlog).astype(np.float32)
    guide = (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)
    out = P.bilateral_slice(grid, guide)
    return float(((out - target) ** 2).sum())
for seed in (1234, 0, 1, 2, 3, 4):
    print(seed, run(seed))
