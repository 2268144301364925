"""Direct oracle parity of the HIP path at every BASELINE.json config size, forward AND backward, plus
the reference tests that round 1 only ran shortened or at reduced size:

* config #2 (1920x1080, grid 16x16x8x12) and config #5 (4000x3000, grid 32x32x8x12; also through the
  uint16 / 32767 wire format of hdrnet/data_pipeline.py:267-274) forward vs the C oracle (OpenMP);
* 1080p and 4K backward, all three VJPs, vs the C oracle's gather-form gradients
  (hdrnet/ops/bilateral_slice_apply.cc:84-259);
* hdrnet/test/ops_test.py:189-322 -- test_grid_optimize / test_guide_optimize / test_optimize_both
  with the reference's data, optimiser (plain gradient descent on the summed squared error),
  learning rates, step counts and thresholds, on the HIP ops;
* hdrnet/hdrnet_ops_jax_tf2_test.py:26-48 -- JAX twin == op at the reference's real size
  (batch 4, 640x480, grid 16x12x8x2), assertAllClose defaults (rtol = atol = 1e-6).

The oracle is only the checker here (tests/ may import oracle/).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from conftest import check_dgrid, check_pixel_grad  # noqa: E402

FWD_RTOL = FWD_ATOL = 1e-5   # required; the 1e-6 bar is reported


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


def frame(rng, H, W, GH, GW, GD, lo=-0.02, hi=1.02):
    grid = rng.random((1, GH, GW, GD, 12), dtype=np.float32)
    guide = (rng.random((1, H, W), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    inp = rng.random((1, H, W, 3), dtype=np.float32)
    return grid, guide, inp


CONFIGS = {
    "1080p (config #2)": (1080, 1920, 16, 16, 8),
    "4K (config #3 hot path)": (2160, 3840, 16, 16, 8),
    "hdrp 4000x3000 (config #5)": (3000, 4000, 32, 32, 8),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_forward_vs_oracle_at_config_size(dev, ops, mt_port, name):
    H, W, GH, GW, GD = CONFIGS[name]
    rng = np.random.default_rng(H * 7 + W)
    grid, guide, inp = frame(rng, H, W, GH, GW, GD)
    want = mt_port.bilateral_slice_apply(grid, guide, inp, True)
    got = N(ops.bilateral_slice_apply(T(grid, dev), T(guide, dev), T(inp, dev), has_offset=True))
    assert ops.last_kernel() == "apply_fwd_seg/vec4"
    err = np.abs(got - want)
    worst = float((err / (1e-6 + 1e-6 * np.abs(want))).max())
    print(f"{name}: max|err| = {err.max():.3e}, worst / (1e-6 bar) = {worst:.2f}")
    np.testing.assert_allclose(got, want, rtol=FWD_RTOL, atol=FWD_ATOL)
    assert worst < 4.0, worst


def test_hdrp_uint16_wire_format_vs_oracle(dev, ops, mt_port):
    """Config #5 as stated: uint16 linear in [0, 32767] -> float / 32767 (data_pipeline.py:267-274),
    4000x3000, grid 32x32x8x12, through the fused wire-format entry point."""
    H, W, GH, GW, GD = CONFIGS["hdrp 4000x3000 (config #5)"]
    rng = np.random.default_rng(55)
    grid, guide, _ = frame(rng, H, W, GH, GW, GD)
    raw = rng.integers(0, 32768, size=(1, H, W, 3), dtype=np.uint16)
    inp = (raw.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    want = mt_port.bilateral_slice_apply(grid, guide, inp, True)
    traw = torch.from_numpy(raw.astype(np.int32)).to(dev).to(torch.uint16)
    got = N(ops.bilateral_slice_apply_io(T(grid, dev), traw, guide=T(guide, dev), input_white_level=32767.0,
                                         has_offset=True))
    assert ops.last_kernel().startswith("apply_fwd_io/u16->f32"), ops.last_kernel()
    np.testing.assert_allclose(got, want, rtol=FWD_RTOL, atol=FWD_ATOL)


@pytest.mark.parametrize("name", ["1080p (config #2)", "4K (config #3 hot path)"])
def test_backward_vs_oracle_at_config_size(dev, ops, mt_port, name):
    """All three VJPs of a full frame against the oracle's gather-form gradients.  Tolerances
    (tests/conftest.py): rtol 1e-4; dgrid atol = 1e-5 x max|want| (cells are sums of ~30 000 terms of
    random sign), dinput FLAT atol 1e-5, dguide FLAT atol 2e-5 (the reference's own f32 noise is 1.1e-5)."""
    H, W, GH, GW, GD = CONFIGS[name]
    rng = np.random.default_rng(H + 3 * W)
    grid, guide, inp = frame(rng, H, W, GH, GW, GD)
    dout = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    wg, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
    ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
    assert ops.last_kernel() == "apply_bwd_fused/mfma", ops.last_kernel()
    check_dgrid(N(tg.grad), wg, name)
    check_pixel_grad(N(tgu.grad), wgu, name, "dguide")
    check_pixel_grad(N(ti.grad), wi, name, "dinput")


def test_per_pixel_vjps_without_dgrid_vs_oracle_at_1080p(dev, ops, mt_port):
    """dguide + dinput with the grid frozen (no dgrid requested): the stand-alone per-pixel kernel on the
    forward's core, a full 1080p frame against the oracle."""
    H, W, GH, GW, GD = CONFIGS["1080p (config #2)"]
    rng = np.random.default_rng(77)
    grid, guide, inp = frame(rng, H, W, GH, GW, GD)
    dout = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    _, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    tg = T(grid, dev)
    tgu, ti = (T(a, dev).requires_grad_(True) for a in (guide, inp))
    ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
    assert ops.last_kernel() == "apply_vjp_seg/vec4", ops.last_kernel()
    check_pixel_grad(N(tgu.grad), wgu, "vjp_seg 1080p", "dguide")
    check_pixel_grad(N(ti.grad), wi, "vjp_seg 1080p", "dinput")


@pytest.mark.parametrize("B,H,W,GH,GW", [(3, 540, 960, 8, 8), (4, 270, 480, 16, 16), (2, 600, 450, 5, 9)])
def test_batched_backward_vs_oracle_every_gradient_subset(dev, ops, mt_port, B, H, W, GH, GW):
    """The rows-per-task plan of the fused backward depends on the batch, the frame, the grid AND on which
    gradients are requested (each kernel variant has its own resident-workgroup count): every subset that
    includes dgrid, batched, against the oracle; the subsets must also agree with each other to rounding."""
    GD = 8
    rng = np.random.default_rng(B * 1000 + H)
    grid = rng.random((B, GH, GW, GD, 12), dtype=np.float32)
    guide = rng.random((B, H, W), dtype=np.float32)
    inp = rng.random((B, H, W, 3), dtype=np.float32)
    dout = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    wg, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    want = {"dgrid": wg, "dguide": wgu, "dinput": wi}
    for need_guide, need_input in ((True, True), (True, False), (False, True), (False, False)):
        tg = T(grid, dev).requires_grad_(True)
        tgu = T(guide, dev).requires_grad_(need_guide)
        ti = T(inp, dev).requires_grad_(need_input)
        ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
        kern = ops.last_kernel()
        assert kern == ("apply_bwd_fused/mfma" if (need_guide or need_input) else "grid_grad_mfma"), kern
        got = {"dgrid": tg.grad, "dguide": tgu.grad, "dinput": ti.grad}
        tag = f"B={B} {H}x{W} (dguide={need_guide}, dinput={need_input})"
        check_dgrid(N(got["dgrid"]), want["dgrid"], tag)
        for nm in ("dguide", "dinput"):
            if got[nm] is not None:
                check_pixel_grad(N(got[nm]), want[nm], tag, nm)


# ---- hdrnet/test/ops_test.py:178-322, the reference's data / optimiser / step counts / thresholds ----
def _fit(ops, grid, guide_logits, target, lr, steps, opt_grid, opt_guide):
    params = [p for p, on in ((grid, opt_grid), (guide_logits, opt_guide)) if on]
    opt = torch.optim.SGD(params, lr=lr)  # tf.train.GradientDescentOptimizer(lr) on sum of squares
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        guide = torch.sigmoid(guide_logits) if opt_guide else guide_logits
        out = ops.bilateral_slice(grid, guide)
        loss = ((target - out) ** 2).sum()
        loss.backward()
        opt.step()
    with torch.no_grad():
        guide = torch.sigmoid(guide_logits) if opt_guide else guide_logits
        out = ops.bilateral_slice(grid, guide)
    return float(((out - target) ** 2).sum())


def _sine_target(dev, w):
    t = np.sin(np.linspace(0, 2 * np.pi, w)).astype(np.float32)
    return T(t[np.newaxis, np.newaxis, :, np.newaxis], dev)


def test_grid_optimize_reference(dev, ops):
    """ops_test.py:189-230: 10 000 steps of gradient descent (lr 1e-2) on the grid, SSE < 0.0085."""
    np.random.seed(1234)
    w, gw, gd = 32, 16, 8
    guide = T(np.linspace(0, 1, w).astype(np.float32)[np.newaxis, np.newaxis, :], dev)
    grid = T(np.random.rand(1, 1, gw, gd, 1).astype(np.float32), dev).requires_grad_(True)
    sse = _fit(ops, grid, guide, _sine_target(dev, w), 1e-2, 10000, True, False)
    assert sse < 0.0085, sse


def test_guide_optimize_reference(dev, ops):
    """ops_test.py:232-278: 6 000 steps (lr 1e-3) on the guide logits through a sigmoid, SSE < 1e-4."""
    w, gw, gd = 32, 8, 2
    guide0 = np.linspace(0.5 / gd, 1 - 0.5 / gd, w).astype(np.float32)[np.newaxis, np.newaxis, :]
    grid = np.linspace(-1, 1, gd).astype(np.float32)[np.newaxis, np.newaxis, np.newaxis, :, np.newaxis]
    grid = np.tile(grid, [1, 1, gw, 1, 1])
    logits = T(guide0, dev).requires_grad_(True)
    sse = _fit(ops, T(grid, dev), logits, _sine_target(dev, w), 1e-3, 6000, False, True)
    assert sse < 1e-4, sse


# The reference draws its data with an UNSEEDED np.random, and whether its threshold holds depends on
# the draw: the CPU oracle (reference semantics) itself ends at these SSEs
# (tests/golden/optimize_both_oracle.py), two of six above 1e-4.
OPTIMIZE_BOTH_ORACLE_SSE = {1234: 4.2299519e-04, 2: 1.6063153e-05, 3: 3.7598593e-06}


@pytest.mark.parametrize("seed", sorted(OPTIMIZE_BOTH_ORACLE_SSE))
def test_optimize_both_reference(dev, ops, seed):
    """ops_test.py:280-322: 10 000 steps (lr 1e-1) on grid AND guide logits.  The HIP ops must land
    where the reference's own CPU op lands from the same draw (10 000 chained f32 steps: 2 %), and meet
    the reference's `SSE < 1e-4` for the draws for which the reference itself meets it."""
    np.random.seed(seed)
    w, gw, gd = 32, 8, 2
    logits = T((np.random.rand(1, 1, w).astype(np.float32) * 2.0 - 1.0), dev).requires_grad_(True)
    grid = T(np.random.rand(1, 1, gw, gd, 1).astype(np.float32), dev).requires_grad_(True)
    sse = _fit(ops, grid, logits, _sine_target(dev, w), 1e-1, 10000, True, True)
    want = OPTIMIZE_BOTH_ORACLE_SSE[seed]
    assert abs(sse - want) <= 0.02 * want + 1e-7, (sse, want)
    if want < 1e-4:
        assert sse < 1e-4, sse


def test_jax_twin_equals_op_at_reference_size(dev, ops):
    """hdrnet_ops_jax_tf2_test.py:26-48 at its real size: batch 4, guide 640x480, grid 16x12x8, 2
    channels; assertAllClose defaults rtol = atol = 1e-6."""
    from oracle import jax_np
    rng = np.random.default_rng(1234)
    grid = rng.random((4, 16, 12, 8, 2), dtype=np.float32)
    guide = rng.random((4, 640, 480), dtype=np.float32)
    want = jax_np.batched(jax_np.bilateral_slice, grid, guide)
    got = N(ops.bilateral_slice(T(grid, dev), T(guide, dev)))
    assert got.shape == (4, 640, 480, 2)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["1080p (config #2)", "4K (config #3 hot path)"])
def test_dguide_noise_hip_vs_float64_against_the_reference_s_own(dev, ops, mt_port, name):
    """tests/conftest.py holds dguide to a flat 2e-5 (4e-5 until round 5) because the reference's OWN float32 arithmetic is
    ~1e-5 away from the float64 value of its formulas -- but is the HIP path merely ordered differently, or noisier?
    Measured here instead of argued: max|HIP - f64| and max|oracle - f64| of dguide (and dinput) on a full frame of the
    suite's data, for both HIP paths that produce dguide (the fused all-gradients pass and the per-pixel VJP kernel).
    Required: HIP's distance to the exact value <= 1.5 x the reference's own (round 4: 2 x, met at 1.91; round 5's
    fused pass contracts the grid's z difference and sums the two tap derivatives without cancellation: 0.5-0.7 x)."""
    from oracle.f64_vjps import f64_vjps
    H, W, GH, GW, GD = CONFIGS[name]
    rng = np.random.default_rng(H + 3 * W)
    grid, guide, inp = frame(rng, H, W, GH, GW, GD)
    dout = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    _, wgu, wi = mt_port.bilateral_slice_apply_grad(grid, guide, inp, dout, True)
    dg64, di64 = f64_vjps(grid, guide, inp, dout)
    ref_g, ref_i = np.abs(wgu - dg64).max(), np.abs(wi - di64).max()
    tg, tgu, ti = (T(a, dev).requires_grad_(True) for a in (grid, guide, inp))
    ops.bilateral_slice_apply(tg, tgu, ti, has_offset=True).backward(T(dout, dev))
    k_all = ops.last_kernel()
    tgu2, ti2 = (T(a, dev).requires_grad_(True) for a in (guide, inp))
    ops.bilateral_slice_apply(T(grid, dev), tgu2, ti2, has_offset=True).backward(T(dout, dev))
    k_px = ops.last_kernel()
    for label, gu, gi in ((k_all, tgu.grad, ti.grad), (k_px, tgu2.grad, ti2.grad)):
        hip_g, hip_i = np.abs(N(gu) - dg64).max(), np.abs(N(gi) - di64).max()
        print(f"{name} [{label}]: dguide |HIP - f64| = {hip_g:.3e}, |reference f32 - f64| = {ref_g:.3e}, ratio {hip_g / ref_g:.2f}; "
              f"dinput {hip_i:.3e} vs {ref_i:.3e}, ratio {hip_i / ref_i:.2f}; max|dguide| = {np.abs(dg64).max():.3g}")
        assert hip_g <= 1.5 * ref_g, (label, hip_g, ref_g)
        assert hip_i <= 1.5 * ref_i + 1e-7, (label, hip_i, ref_i)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_adjointness_and_linearity_at_config_size(dev, ops, name):
    """Size-independent properties of the op, checked on the HIP path alone at every BASELINE config size (no oracle in
    the loop): for fixed guide the output is LINEAR in the grid and AFFINE in the input
    (bilateral_slice_apply.cc:72-80), so with the VJPs of one backward call
        <dout, out(grid)>                      == <dgrid, grid>                       (adjoint of grid -> out)
        <dout, out(input)> - <dout, out(0)>    == <dinput, input>                     (adjoint of input -> out)
        out(a grid1 + b grid2)                 == a out(grid1) + b out(grid2)         (linearity)
    Inner products accumulated in float64; the two sides agree to the float32 rounding of ~1e7 summed terms."""
    H, W, GH, GW, GD = CONFIGS[name]
    gen = torch.Generator(device=dev).manual_seed(H + W)
    grid = torch.rand((1, GH, GW, GD, 12), device=dev, generator=gen).requires_grad_(True)
    grid2 = torch.rand((1, GH, GW, GD, 12), device=dev, generator=gen)
    guide = torch.rand((1, H, W), device=dev, generator=gen) * 1.04 - 0.02
    inp = torch.rand((1, H, W, 3), device=dev, generator=gen).requires_grad_(True)
    dout = torch.randn((1, H, W, 3), device=dev, generator=gen)
    out = ops.bilateral_slice_apply(grid, guide, inp, has_offset=True)
    out.backward(dout)
    dot = lambda a, b: float((a.double() * b.double()).sum())  # noqa: E731
    lhs = dot(dout, out.detach())
    scale = float((dout.double().abs() * out.detach().double().abs()).sum())
    assert abs(lhs - dot(grid.grad, grid.detach())) <= 2e-6 * scale, (lhs, dot(grid.grad, grid.detach()), scale)
    with torch.no_grad():
        out0 = ops.bilateral_slice_apply(grid.detach(), guide, torch.zeros_like(inp), has_offset=True)
        assert abs((lhs - dot(dout, out0)) - dot(inp.grad, inp.detach())) <= 2e-6 * scale
        a, b = 0.75, -1.5
        o2 = ops.bilateral_slice_apply(grid2, guide, inp.detach(), has_offset=True)
        oc = ops.bilateral_slice_apply(a * grid.detach() + b * grid2, guide, inp.detach(), has_offset=True)
        torch.testing.assert_close(oc, a * out.detach() + b * o2, rtol=1e-5, atol=2e-5)
