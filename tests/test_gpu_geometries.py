"""Oracle parity of the fast paths at the launch geometries no small-shape test reaches (VERDICT r02,
"What's missing" 3 / "Next round" 1).  The rows-per-task plan of the gradient pass, the segment plan of
the forward and the load / store flavour are all chosen PER LAUNCH from the frame geometry, so each
geometry BASELINE.json names gets its own direct check against the C oracle (OpenMP):

* BilateralSlice forward + both gradients at 1080p and 4K, C = 12 and C = 1
  (hdrnet/ops/bilateral_slice.cc:25-168);
* BilateralSliceApply backward, all three VJPs, at 4000x3000 / grid 32x32x8 (config #5's per-GPU shape)
  and at B = 4 x 1080p (config #4's per-GPU shape) (hdrnet/ops/bilateral_slice_apply.cc:84-259);
* BilateralSliceApply forward at 1080p for has_offset = False and for Cin = 1;
* the (Cin, Cout) = (4, 4) shape with and without offset, forward and backward (the fast-path tables
  used to disagree on it).

Tolerances (tests/conftest.py): forward rtol = atol = 1e-5; gradients rtol 1e-4 with FLAT atols for the
per-pixel VJPs -- dinput 1e-5, dguide 2e-5 (the reference's own float32 arithmetic
is 1.1e-5 from the float64 value of its formula on this data, tools/dguide_noise_floor.py) -- and
atol = 1e-5 x max|want| for dgrid (a cell is a sum of tens of thousands of terms of random sign).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import check_dgrid, check_pixel_grad  # noqa: E402

FWD_TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from hdrnet_amd import hdrnet_ops
    return hdrnet_ops


@pytest.fixture(scope="module")
def mt_port(port):
    port.set_threads(os.cpu_count() or 1)
    yield port
    port.set_threads(1)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def guide_map(rng, shape, lo=-0.02, hi=1.02):
    return (rng.random(shape, dtype=np.float32) * (hi - lo) + lo).astype(np.float32)


FRAMES = {"1080p": (1080, 1920), "4K": (2160, 3840)}


# ---- BilateralSlice at full frame size -------------------------------------------------------------
@pytest.mark.parametrize("C", [12, 1])
@pytest.mark.parametrize("frame", list(FRAMES))
def test_slice_forward_vs_oracle_at_frame_size(dev, ops, mt_port, frame, C):
    H, W = FRAMES[frame]
    rng = np.random.default_rng(H + 31 * C)
    grid = rng.random((1, 16, 16, 8, C), dtype=np.float32)
    guide = guide_map(rng, (1, H, W))
    want = mt_port.bilateral_slice(grid, guide)
    got = N(ops.bilateral_slice(T(grid, dev), T(guide, dev)))
    assert ops.last_kernel() == "slice_fwd_rows", ops.last_kernel()
    err = np.abs(got - want)
    print(f"slice fwd {frame} C={C}: max|err| = {err.max():.3e}, worst / (1e-6 bar) = "
          f"{(err / (1e-6 + 1e-6 * np.abs(want))).max():.2f}")
    np.testing.assert_allclose(got, want, **FWD_TOL)


@pytest.mark.parametrize("C", [12, 1])
@pytest.mark.parametrize("frame", list(FRAMES))
def test_slice_backward_vs_oracle_at_frame_size(dev, ops, mt_port, frame, C):
    H, W = FRAMES[frame]
    rng = np.random.default_rng(H + 17 * C)
    grid = rng.random((1, 16, 16, 8, C), dtype=np.float32)
    guide = guide_map(rng, (1, H, W))
    dout = rng.standard_normal((1, H, W, C)).astype(np.float32)
    wg, wgu = mt_port.bilateral_slice_grad(grid, guide, dout)
    tg, tgu = (T(a, dev).requires_grad_(True) for a in (grid, guide))
    ops.bilateral_slice(tg, tgu).backward(T(dout, dev))
    kern = ops.last_kernel()
    # C % 4 == 0: all in one fused pass; C = 1: the row kernel for dguide + the MFMA pass for dgrid
    assert kern == ("slice_bwd_fused/mfma" if C % 4 == 0 else "slice_vjp_rows/vec4+grid_grad_mfma"), kern
    check_dgrid(N(tg.grad), wg, f"slice bwd {frame} C={C}")
    check_pixel_grad(N(tgu.grad), wgu, f"slice bwd {frame} C={C}", "dguide")


# ---- BilateralSliceApply backward at config #5's and config #4's per-GPU shapes ---------------------
@pytest.mark.parametrize("name,B,H,W,GH,GW", [("hdrp 4000x3000 / 32x32x8 (config #5)", 1, 3000, 4000, 32, 32),
                                              ("4 x 1080p (config #4 per GPU)", 4, 1080, 1920, 16, 16)])
def test_apply_backward_vs_oracle_at_config_shape(dev, ops, mt_port, name, B, H, W, GH, GW):
    rng = np.random.default_rng(B * 100 + H)
    grid = rng.random((B, GH, GW, 8, 12), dtype=np.float32)
    guide = guide_map(rng, (B, H, W))
    inp = rng.random((B, H, W, 3), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    wg, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
    ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
    assert ops.last_kernel() == "apply_bwd_fused/mfma", ops.last_kernel()
    check_dgrid(N(tg.grad), wg, name)
    check_pixel_grad(N(tgu.grad), wgu, name, "dguide")
    check_pixel_grad(N(ti.grad), wi, name, "dinput")
    # the training case (the input needs no gradient) plans its rows per task on its own
    tg2, tgu2 = (T(a, dev).requires_grad_(True) for a in (grid, guide))
    ops.bilateral_slice_apply(tg2, tgu2, T(inp, dev), has_offset=True).backward(T(dout, dev))
    assert ops.last_kernel() == "apply_bwd_fused/mfma", ops.last_kernel()
    check_dgrid(N(tg2.grad), wg, name + " (no dinput)")
    check_pixel_grad(N(tgu2.grad), wgu, name + " (no dinput)", "dguide")


# ---- forward at 1080p: the other channel configurations ---------------------------------------------
@pytest.mark.parametrize("Cin,Cout,off", [(3, 3, False), (1, 1, True), (1, 1, False), (1, 3, True), (3, 4, True)])
def test_forward_1080p_other_shapes_vs_oracle(dev, ops, mt_port, Cin, Cout, off):
    H, W = FRAMES["1080p"]
    rng = np.random.default_rng(Cin * 10 + Cout + (5 if off else 0))
    grid = rng.random((1, 16, 16, 8, Cout * (Cin + (1 if off else 0))), dtype=np.float32)
    guide = guide_map(rng, (1, H, W))
    inp = rng.random((1, H, W, Cin), dtype=np.float32)
    want = mt_port.bilateral_slice_apply(grid, guide, inp, off)
    got = N(ops.bilateral_slice_apply(T(grid, dev), T(guide, dev), T(inp, dev), has_offset=off))
    assert ops.last_kernel() == "apply_fwd_seg/vec4", ops.last_kernel()
    err = np.abs(got - want)
    print(f"fwd 1080p ({Cin},{Cout},{off}): max|err| = {err.max():.3e}")
    np.testing.assert_allclose(got, want, **FWD_TOL)


# ---- (4, 4): with and without offset, both directions ------------------------------------------------
@pytest.mark.parametrize("off", [False, True])
def test_four_by_four_both_directions(dev, ops, mt_port, off):
    """hdrnet/test/ops_test.py:345-365 uses has_offset = False -> Cout = 4 on 4-channel inputs.  Forward
    and per-pixel VJPs have a fast specialisation for both settings; dgrid rides the MFMA pass -- fused with the
    per-pixel VJPs where C = Cout * Cj <= 16 (no offset: 16), as two 16-column channel windows beside the per-pixel
    kernel where not (offset: 20; the generic gather until round 6)."""
    B, H, W, GH, GW, GD = 2, 270, 480, 8, 8, 8
    Cj = 4 + (1 if off else 0)
    rng = np.random.default_rng(44 + Cj)
    grid = rng.random((B, GH, GW, GD, 4 * Cj), dtype=np.float32)
    guide = guide_map(rng, (B, H, W))
    inp = rng.random((B, H, W, 4), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, 4)).astype(np.float32)
    want = mt_port.bilateral_slice_apply(grid, guide, inp, off)
    wg, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, off)
    tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
    out = ops.bilateral_slice_apply(tg, tgu, ti, has_offset=off)
    assert ops.last_kernel() == "apply_fwd_seg/vec4", ops.last_kernel()
    np.testing.assert_allclose(N(out), want, **FWD_TOL)
    out.backward(T(dout, dev))
    kern = ops.last_kernel()
    assert kern == ("apply_vjp_seg/vec4+grid_grad_mfma" if off else "apply_bwd_fused/mfma"), kern
    check_dgrid(N(tg.grad), wg, f"(4,4,{off})")
    check_pixel_grad(N(tgu.grad), wgu, f"(4,4,{off})", "dguide")
    check_pixel_grad(N(ti.grad), wi, f"(4,4,{off})", "dinput")


# ---- one frame as row bands (hdrnet_bilateral_slice_apply_rows_f32; SURVEY.md section 8e, optional) ---
@pytest.mark.parametrize("H,W,GH,GW,cuts", [(270, 480, 16, 16, (0, 135, 270)),
                                            (271, 480, 7, 9, (0, 1, 100, 100, 271)),
                                            (2160, 3840, 16, 16, tuple(270 * i for i in range(9)))])
def test_row_bands_equal_whole_frame(dev, ops, mt_port, H, W, GH, GW, cuts):
    """Bands of any partition (1-row and empty ones included), concatenated: bit-identical to the
    whole-frame reference on the generic kernel, bit-identical to the whole-frame launch on the fast
    kernel (same arithmetic per pixel; gyf = (y0 + y + .5) * GH / H_total), and the fast one within the
    forward tolerance of the oracle.  The last case is a 4K frame over 8 GPUs."""
    rng = np.random.default_rng(H + W)
    grid = rng.random((1, GH, GW, 8, 12), dtype=np.float32)
    guide = guide_map(rng, (1, H, W))
    inp = rng.random((1, H, W, 3), dtype=np.float32)
    want = mt_port.bilateral_slice_apply(grid, guide, inp, True)
    tg, tgu, ti = T(grid, dev), T(guide, dev), T(inp, dev)
    for family, kern in (("generic", "apply_fwd_generic"), ("fast", "apply_fwd_seg/vec4")):
        with ops.kernel_override(family):
            whole = N(ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True))
            bands = []
            for a, b in zip(cuts, cuts[1:]):
                bands.append(N(ops.bilateral_slice_apply_rows(tg, tgu[:, a:b], ti[:, a:b], H, a, True)))
                if b > a:
                    assert ops.last_kernel() == kern, ops.last_kernel()
        got = np.concatenate(bands, axis=1)
        assert np.array_equal(got, whole), family
        if family == "generic":
            assert np.array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, **FWD_TOL)
    with pytest.raises(ValueError):
        ops.bilateral_slice_apply_rows(tg, tgu[:, :10], ti[:, :10], H, H - 5, True)
