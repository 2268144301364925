"""Generates tests/golden/*.npz from the reference's OWN CPU implementation
(oracle/_ref/libhdrnet_ref.so = /root/reference/hdrnet/ops/bilateral_slice_apply.cc +
bilateral_slice.cc compiled unchanged, see oracle/Makefile).

Run in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

The fixtures pin the oracle port (tests/test_oracle_pinning.py) and the HIP path
(tests/test_gpu_parity.py) wherever /root/reference is absent (the GPU box).
Inputs follow the reference's own tests: np.random.seed(1234); np.random.rand(...)
(hdrnet/hdrnet_ops_test.py:101-107, :277-285) and its test extents (:91-100, :185-210,
:366-408; hdrnet/test/ops_test.py:61-86, :345-365).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (B, H, W, Cin, GH, GW, GD, Cout, has_offset, guide_lo, guide_hi)
APPLY_CASES = {
    # hdrnet_ops_test.py:91-100 create_forward_test defaults
    "apply_forward_default": (3, 30, 25, 3, 16, 12, 8, 3, True, 0.0, 1.0),
    # hdrnet_ops_test.py:366-378 / :396-408 grid + input gradient extents
    "apply_grad_grid": (3, 8, 5, 3, 6, 3, 7, 4, True, 0.0, 1.0),
    # hdrnet_ops_test.py:380-393 guide gradient extents
    "apply_grad_guide": (1, 6, 18, 1, 3, 9, 7, 1, True, 0.0, 1.0),
    # hdrnet/test/ops_test.py:345-365: has_offset False -> 4 outputs from a 12-ch grid
    "apply_no_offset": (2, 12, 9, 3, 5, 4, 6, 4, False, 0.0, 1.0),
    # not covered by the reference's tests: guide outside [0, 1], HDRNet's Cj=4 / Cout=3
    "apply_guide_out_of_range": (2, 21, 34, 3, 4, 6, 8, 3, True, -0.4, 1.4),
    # image smaller than the grid
    "apply_tiny_image": (1, 3, 2, 3, 8, 8, 4, 3, True, 0.0, 1.0),
}

# name: (B, H, W, GH, GW, GD, C, guide_lo, guide_hi)
SLICE_CASES = {
    "slice_forward_default": (3, 30, 25, 16, 12, 8, 12, 0.0, 1.0),   # hdrnet_ops_test.py:91-100
    "slice_grad_grid": (3, 8, 5, 6, 3, 7, 16, 0.0, 1.0),             # :185-195 (gc = 4 * 4)
    "slice_grad_guide": (1, 6, 18, 3, 9, 7, 2, 0.0, 1.0),            # :200-210
    "slice_jax_shape_small": (2, 64, 48, 16, 12, 8, 2, 0.0, 1.0),    # hdrnet_ops_jax_tf2_test.py:28-34 (h,w / 10)
    "slice_guide_out_of_range": (2, 17, 23, 5, 4, 3, 5, -0.5, 1.5),
}


def main():
    R = oracle.ref()
    out = {}
    for name, (B, H, W, Cin, GH, GW, GD, Cout, ho, lo, hi) in APPLY_CASES.items():
        np.random.seed(1234)
        Cj = Cin + int(ho)
        grid = np.random.rand(B, GH, GW, GD, Cout * Cj).astype(np.float32)
        guide = (np.random.rand(B, H, W) * (hi - lo) + lo).astype(np.float32)
        inp = np.random.rand(B, H, W, Cin).astype(np.float32)
        dout = np.random.randn(B, H, W, Cout).astype(np.float32)
        o = R.bilateral_slice_apply(grid, guide, inp, ho)
        dgrid, dguide, dinput = R.bilateral_slice_apply_grad(grid, guide, inp, dout, ho)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), grid=grid, guide=guide, input=inp,
                            dout=dout, out=o, dgrid=dgrid, dguide=dguide, dinput=dinput,
                            has_offset=np.array(ho))
        out[name] = float(np.abs(o).max())
    for name, (B, H, W, GH, GW, GD, C, lo, hi) in SLICE_CASES.items():
        np.random.seed(1234)
        grid = np.random.rand(B, GH, GW, GD, C).astype(np.float32)
        guide = (np.random.rand(B, H, W) * (hi - lo) + lo).astype(np.float32)
        dout = np.random.randn(B, H, W, C).astype(np.float32)
        o = R.bilateral_slice(grid, guide)
        dgrid, dguide = R.bilateral_slice_grad(grid, guide, dout)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), grid=grid, guide=guide, dout=dout,
                            out=o, dgrid=dgrid, dguide=dguide)
        out[name] = float(np.abs(o).max())
    # Known-answer test of hdrnet/test/ops_test.py:61-86 (inputs are analytic; store the
    # reference's actual outputs, which sit ~2e-4 below `val` because of the smoothed tent).
    d = 3
    grid = np.zeros((3, 3, 4, d, 1), np.float32)
    grid[:, :, :, 1, :] = 1.0
    grid[:, :, :, 2, :] = 2.0
    outs = []
    for val in range(d):
        guide = (((val + 0.5) / (1.0 * d)) * np.ones((3, 5, 9))).astype(np.float32)
        outs.append(R.bilateral_slice(grid, guide))
    np.savez_compressed(os.path.join(HERE, "slice_interpolate_kat.npz"), grid=grid, outs=np.stack(outs))
    for k, v in out.items():
        print(f"{k}: max|out| = {v:.4f}")
    print("bytes:", sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
