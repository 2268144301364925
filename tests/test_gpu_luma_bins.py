"""Oracle parity over the reference's own hyper-parameter range of the grid: `luma_bins` (the grid depth GD,
hdrnet/bin/train.py:235; scripts/ll/train_std.sh:3,13 take it from the command line) x `spatial_bin`
(GH = GW = net_input_size / spatial_bin = 256 / {8, 16, 32} = 32, 16, 8; hdrnet/models.py:63-70,131).

Until round 6 no GPU test used GD > 8 and every gradient at GD > 8 silently ran the generic gather kernel
("What's missing" 1).  The gradient pass now contracts 9 .. 16 planes as two 8-plane tiles per
task (grid_grad_mfma.hip, NH = 2) and skips, per 64-pixel chunk, a tile no tap of the chunk falls into, so the
cases here are chosen to reach every branch of that:

* a U[0, 1) guide (every chunk straddles both halves),
* an image-like guide (a smooth ramp + a little noise: most chunks live in one half, some straddle 7 | 8),
* a guide confined to the lower / the upper half (one tile never contracted: its partial tiles must be zeros),
* GD = 9, 12 (a ragged upper half) and 16, odd frame sizes, every subset of the gradients.

Each case: BilateralSliceApply forward + all three VJPs (hdrnet/ops/bilateral_slice_apply.cc:24-259) and
BilateralSlice forward + both VJPs (bilateral_slice.cc:25-168) against the C oracle, kernel names asserted.
Tolerances: tests/conftest.py (forward rtol = atol = 1e-5; gradients rtol 1e-4, dinput 1e-5 flat, dguide 2e-5
flat x GD / 8 -- the z derivative of the tent is GD x the tap difference, so the reference's own float32 noise
doubles with the depth --, dgrid 1e-5 x max|want|).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import DGUIDE_ATOL, DINPUT_ATOL, GRAD_RTOL, check_dgrid  # noqa: E402

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


def make_guide(rng, kind, B, H, W):
    u = rng.random((B, H, W), dtype=np.float32)
    if kind == "uniform":          # every 64-pixel chunk has taps in both plane halves
        return (u * 1.04 - 0.02).astype(np.float32)
    if kind == "smooth":           # image-like: a diagonal luminance ramp + 2 % noise
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        ramp = 0.5 + 0.5 * np.sin(2.0 * np.pi * (xx / W * 1.5 + yy / H))
        return (0.96 * ramp[None] + 0.04 * u).astype(np.float32)
    if kind == "low":              # planes 0 .. GD/2 - 1 only
        return (u * 0.40).astype(np.float32)
    if kind == "high":             # the upper half only, incl. the forced-1 half cell at the top
        return (0.60 + u * 0.45).astype(np.float32)
    raise ValueError(kind)


def check_pix(got, want, name, what, GD):
    atol = (DGUIDE_ATOL * max(1.0, GD / 8.0)) if what == "dguide" else DINPUT_ATOL
    err = np.abs(got - want)
    print(f"{name} {what}: max|err| = {err.max():.3e}, worst / bar = "
          f"{(err / (atol + GRAD_RTOL * np.abs(want))).max():.2f}")
    np.testing.assert_allclose(got, want, rtol=GRAD_RTOL, atol=atol, err_msg=f"{name} {what}")


def apply_case(dev, ops, P, name, B, H, W, GH, GW, GD, kind, seed, Cin=3, Cout=3, off=True,
               fwd_kernel="apply_fwd_seg/vec4", bwd_kernel="apply_bwd_fused/mfma"):
    rng = np.random.default_rng(seed)
    Cj = Cin + (1 if off else 0)
    grid = rng.random((B, GH, GW, GD, Cout * Cj), dtype=np.float32)
    guide = make_guide(rng, kind, B, H, W)
    inp = rng.random((B, H, W, Cin), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    want = P.bilateral_slice_apply(grid, guide, inp, off)
    wg, wgu, wi = P.bilateral_slice_apply_grad(grid, guide, inp, dout, off)
    tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
    out = ops.bilateral_slice_apply(tg, tgu, ti, has_offset=off)
    if fwd_kernel:
        assert ops.last_kernel() == fwd_kernel, ops.last_kernel()
    err = np.abs(N(out) - want)
    print(f"{name} fwd: max|err| = {err.max():.3e}, worst / (1e-6 bar) = {(err / (1e-6 + 1e-6 * np.abs(want))).max():.2f}")
    np.testing.assert_allclose(N(out), want, **FWD_TOL)
    out.backward(T(dout, dev))
    if bwd_kernel:
        assert ops.last_kernel() == bwd_kernel, ops.last_kernel()
    check_dgrid(N(tg.grad), wg, name)
    check_pix(N(tgu.grad), wgu, name, "dguide", GD)
    check_pix(N(ti.grad), wi, name, "dinput", GD)
    return (grid, guide, inp, dout), (wg, wgu, wi)


def slice_case(dev, ops, P, name, B, H, W, GH, GW, GD, C, kind, seed, bwd_kernel="slice_bwd_fused/mfma"):
    rng = np.random.default_rng(seed)
    grid = rng.random((B, GH, GW, GD, C), dtype=np.float32)
    guide = make_guide(rng, kind, B, H, W)
    dout = rng.standard_normal((B, H, W, C)).astype(np.float32)
    want = P.bilateral_slice(grid, guide)
    wg, wgu = P.bilateral_slice_grad(grid, guide, dout)
    tg, tgu = (T(a, dev).requires_grad_(True) for a in (grid, guide))
    out = ops.bilateral_slice(tg, tgu)
    assert ops.last_kernel() == "slice_fwd_rows", ops.last_kernel()
    np.testing.assert_allclose(N(out), want, **FWD_TOL)
    out.backward(T(dout, dev))
    if bwd_kernel:
        assert ops.last_kernel() == bwd_kernel, ops.last_kernel()
    check_dgrid(N(tg.grad), wg, name)
    check_pix(N(tgu.grad), wgu, name, "dguide", GD)


# ---- luma_bins x spatial_bin at 1080p -------------------------------
@pytest.mark.parametrize("spatial_bin", [8, 16, 32])
@pytest.mark.parametrize("luma_bins", [4, 16])
def test_apply_1080p_luma_bins_x_spatial_bin(dev, ops, mt_port, luma_bins, spatial_bin):
    G = 256 // spatial_bin
    apply_case(dev, ops, mt_port, f"apply 1080p lb={luma_bins} sb={spatial_bin}", 1, 1080, 1920, G, G, luma_bins,
               "uniform", 100 * luma_bins + spatial_bin)


@pytest.mark.parametrize("spatial_bin", [8, 16, 32])
@pytest.mark.parametrize("luma_bins", [4, 16])
def test_slice_1080p_luma_bins_x_spatial_bin(dev, ops, mt_port, luma_bins, spatial_bin):
    G = 256 // spatial_bin
    slice_case(dev, ops, mt_port, f"slice 1080p lb={luma_bins} sb={spatial_bin}", 1, 1080, 1920, G, G, luma_bins, 12,
               "uniform", 7 * luma_bins + spatial_bin)


# ---- GD = 16 on a 4K frame: the fast pass is the one selected -----------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "smooth"])
def test_apply_4k_luma_bins_16(dev, ops, mt_port, kind):
    apply_case(dev, ops, mt_port, f"apply 4K lb=16 {kind} guide", 1, 2160, 3840, 16, 16, 16, kind, 1600 + len(kind))


# ---- the plane-half skip: guides that leave a half empty, straddle rarely, or always ------------------------
@pytest.mark.parametrize("kind", ["smooth", "low", "high"])
@pytest.mark.parametrize("GD", [9, 12, 16])
def test_apply_plane_halves(dev, ops, mt_port, GD, kind):
    apply_case(dev, ops, mt_port, f"apply 540x960 GD={GD} {kind}", 2, 540, 960, 16, 16, GD, kind, GD * 31 + len(kind))


@pytest.mark.parametrize("kind", ["smooth", "high"])
@pytest.mark.parametrize("GD,C", [(16, 12), (12, 4), (16, 16), (10, 8)])
def test_slice_plane_halves(dev, ops, mt_port, GD, C, kind):
    slice_case(dev, ops, mt_port, f"slice 540x960 GD={GD} C={C} {kind}", 2, 540, 960, 16, 16, GD, C, kind,
               GD * 13 + C + len(kind))


# ---- other channel shapes and ragged frames at GD > 8 ------------------------------------------------------
@pytest.mark.parametrize("case", [
    # B, H, W, GH, GW, GD, Cin, Cout, off, forward kernel, backward kernel (None: whatever the dispatcher picks)
    (2, 271, 483, 7, 9, 16, 3, 3, True, None, None),              # ragged width: the row kernels + the MFMA pass
    (1, 270, 480, 16, 16, 16, 4, 4, False, "apply_fwd_seg/vec4", "apply_bwd_fused/mfma"),   # C = 16
    (1, 270, 480, 8, 8, 13, 1, 1, True, "apply_fwd_seg/vec4", None),                       # C = 2: dgrid alone on the pass
    (3, 135, 240, 16, 12, 16, 1, 3, True, "apply_fwd_seg/vec4", None),                     # C = 6
    (1, 270, 480, 16, 16, 16, 3, 3, False, "apply_fwd_seg/vec4", None),                    # C = 9
    (1, 37, 53, 16, 16, 16, 3, 3, True, None, None),
    # C = 20 (round 6): dgrid as two 16-column channel windows of the MFMA pass beside the per-pixel VJP kernel -- the
    # last listed shape that took the generic gather
    (1, 270, 480, 16, 16, 8, 4, 4, True, "apply_fwd_seg/vec4", "apply_vjp_seg/vec4+grid_grad_mfma"),
    (2, 270, 480, 8, 16, 16, 4, 4, True, "apply_fwd_seg/vec4", "apply_vjp_seg/vec4+grid_grad_mfma"),
    (1, 540, 960, 16, 16, 4, 4, 4, True, "apply_fwd_seg/vec4", "apply_vjp_seg/vec4+grid_grad_mfma"),
])
def test_apply_other_shapes_deep_grid(dev, ops, mt_port, case):
    B, H, W, GH, GW, GD, Cin, Cout, off, fk, bk = case
    apply_case(dev, ops, mt_port, f"apply {case[:9]}", B, H, W, GH, GW, GD, "uniform", sum(case[:8]), Cin, Cout, off, fk, bk)
    if bk is None:
        assert "generic" not in ops.last_kernel() or H * W < 4096, ops.last_kernel()


def test_gradient_subsets_deep_grid(dev, ops, mt_port):
    """dgrid alone / dgrid + dguide / dgrid + dinput / all three at GD = 16: four kernel instantiations with
    their own register budgets and row plans; each against the oracle."""
    B, H, W, GH, GW, GD = 2, 540, 960, 16, 16, 16
    rng = np.random.default_rng(616)
    grid = rng.random((B, GH, GW, GD, 12), dtype=np.float32)
    guide = make_guide(rng, "smooth", B, H, W)
    inp = rng.random((B, H, W, 3), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    wg, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    ref_dgrid = None
    for need_gu, need_in, kern in ((False, False, "grid_grad_mfma"), (True, False, "apply_bwd_fused/mfma"),
                                   (False, True, "apply_bwd_fused/mfma"), (True, True, "apply_bwd_fused/mfma")):
        tg = T(grid, dev).requires_grad_(True)
        tgu = T(guide, dev).requires_grad_(need_gu)
        ti = T(inp, dev).requires_grad_(need_in)
        ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
        assert ops.last_kernel() == kern, ops.last_kernel()
        name = f"subset dguide={need_gu} dinput={need_in}"
        check_dgrid(N(tg.grad), wg, name)
        if need_gu:
            check_pix(N(tgu.grad), wgu, name, "dguide", GD)
        if need_in:
            check_pix(N(ti.grad), wi, name, "dinput", GD)
        # the contraction is the same fmaf chain in every instantiation with the same row plan; across plans the
        # summation order differs: rounding only
        if ref_dgrid is None:
            ref_dgrid = N(tg.grad)
        else:
            np.testing.assert_allclose(N(tg.grad), ref_dgrid, rtol=1e-4, atol=1e-5 * float(np.abs(wg).max()))


def test_deep_grid_backward_is_bit_repeatable(dev, ops):
    """No atomics at NH = 2 either: two launches on the same data are bit-identical."""
    rng = np.random.default_rng(99)
    B, H, W, GD = 1, 1080, 1920, 16
    grid = rng.random((B, 16, 16, GD, 12), dtype=np.float32)
    guide, inp = make_guide(rng, "uniform", B, H, W), rng.random((B, H, W, 3), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    res = []
    for _ in range(2):
        tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
        ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
        res.append((N(tg.grad), N(tgu.grad), N(ti.grad)))
    for a, b in zip(*res):
        assert np.array_equal(a, b)


_WARN_SCRIPT = """
import numpy as np, torch
from hdrnet_amd import hdrnet_ops as ops
rng = np.random.default_rng(17)
dev = torch.device("cuda:0")
B, H, W, GD = 1, 270, 480, 17
tg = torch.from_numpy(rng.random((B, 4, 4, GD, 12), dtype=np.float32)).to(dev).requires_grad_(True)
tgu = torch.from_numpy(rng.random((B, H, W), dtype=np.float32)).to(dev)
ti = torch.from_numpy(rng.random((B, H, W, 3), dtype=np.float32)).to(dev)
for _ in range(2):
    ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).sum().backward()
torch.cuda.synchronize()
print("KERNEL", ops.last_kernel())
"""


def test_generic_fallback_warns_once_on_a_large_frame():
    """GD = 17 has no fast gradient pass: HDRNET_KERNEL_AUTO takes the generic gather, and says so on stderr once
    per process when the call is frame-sized (capi.hip: warn_generic_grid_grad).  In a process of its own: the
    warning is once per process and other tests of this run may have drawn it already."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _WARN_SCRIPT], capture_output=True, text=True, cwd=root,
                         env=dict(os.environ, HDRNET_AMD_KERNEL_NAMES="1"), timeout=600)
    assert res.returncode == 0, res.stderr
    assert "generic" in res.stdout, res.stdout
    assert res.stderr.count("takes the generic grid-gradient kernel") == 1, res.stderr
