"""Seeded fuzz of the dispatcher: random frames, grids, channel shapes, gradient subsets and pointer alignments through
the C-ABI, the kernels HDRNET_KERNEL_AUTO picks against the generic kernels of the same library (which the parity suite
pins bit-exactly to the reference's CPU op: tests/test_gpu_parity.py::test_apply_forward_random).  No oracle in the
loop, so two hundred cases run in seconds; what it guards is everything chosen PER CALL -- the segment plan of the
forward, the row plan and the plane halves / channel windows of the gradient pass, the vec4 / scalar flavours by width
and by pointer alignment, the fused / un-fused gradient split by which outputs are wanted
(hdrnet/ops/bilateral_slice_apply.cc:24-259, bilateral_slice.cc:25-168).

Tolerances are the suite's (tests/conftest.py): forward 1e-5, dinput flat 1e-5, dguide flat 2e-5 scaled by GD / 8 (the
derivative of the z tent carries a factor GD), dgrid 1e-5 x max|want|; all with rtol 1e-4 on the gradients."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SHAPES = [(3, 3, True), (3, 3, False), (3, 4, True), (1, 1, True), (1, 1, False), (1, 3, True), (4, 4, True),
          (4, 4, False), (2, 5, True), (3, 1, True)]  # the last two have no fast specialisation


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from hdrnet_amd import hdrnet_ops
    return hdrnet_ops


def draw_case(rng):
    Cin, Cout, off = SHAPES[rng.integers(len(SHAPES))]
    B = int(rng.integers(1, 4))
    H = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 135, 270]))
    W = int(rng.choice([1, 3, 4, 5, 64, 127, 128, 250, 256, 480, 483, 960, 1000, 1024, 1028]))
    GH, GW = int(rng.choice([1, 2, 3, 7, 8, 16, 32])), int(rng.choice([1, 2, 5, 8, 16, 31, 32]))
    GD = int(rng.choice([1, 2, 4, 7, 8, 9, 12, 16, 17]))
    lo, hi = [(0.0, 1.0), (-0.2, 1.2), (0.45, 0.55), (0.0, 0.06)][rng.integers(4)]
    return B, H, W, GH, GW, GD, Cin, Cout, off, lo, hi


def misaligned(t, rng):
    """The same values at a storage offset of 1 .. 3 floats: contiguous, but not 16-byte aligned."""
    k = int(rng.integers(1, 4))
    buf = torch.empty(t.numel() + k, dtype=t.dtype, device=t.device)
    v = buf[k:].view(t.shape)
    v.copy_(t)
    return v


@pytest.mark.parametrize("seed", range(8))
def test_apply_fuzz_auto_against_generic(dev, ops, seed):
    rng = np.random.default_rng(1000 + seed)
    gen = torch.Generator(device=dev).manual_seed(1000 + seed)
    kernels = set()
    for case in range(25):
        B, H, W, GH, GW, GD, Cin, Cout, off, lo, hi = draw_case(rng)
        Cj = Cin + (1 if off else 0)
        grid = torch.rand((B, GH, GW, GD, Cout * Cj), device=dev, generator=gen)
        guide = torch.rand((B, H, W), device=dev, generator=gen) * (hi - lo) + lo
        inp = torch.rand((B, H, W, Cin), device=dev, generator=gen)
        dout = torch.randn((B, H, W, Cout), device=dev, generator=gen)
        if rng.random() < 0.25:
            guide, inp, dout = misaligned(guide, rng), misaligned(inp, rng), misaligned(dout, rng)
        if rng.random() < 0.15:
            grid = misaligned(grid, rng)
        need = [bool(rng.integers(2)) for _ in range(3)]
        if not any(need):
            need[int(rng.integers(3))] = True
        tag = f"seed {seed} case {case}: {(B, H, W, GH, GW, GD, Cin, Cout, off)} guide [{lo}, {hi}] grads {need}"
        res = {}
        for which in ("generic", "auto"):
            tg, tgu, ti = (t.detach().requires_grad_(n) for t, n in zip((grid, guide, inp), need))
            with ops.kernel_override(which):
                out = ops.bilateral_slice_apply(tg, tgu, ti, has_offset=off)
                kf = ops.last_kernel()
                out.backward(dout)
                kb = ops.last_kernel()
            res[which] = (out.detach(), tg.grad, tgu.grad, ti.grad, kf, kb)
        kernels.update((res["auto"][4], res["auto"][5]))
        a, g = res["auto"], res["generic"]
        torch.testing.assert_close(a[0], g[0], rtol=1e-5, atol=1e-5, msg=lambda m: f"{tag} fwd [{a[4]}]: {m}")
        if need[0]:
            scale = max(1.0, float(g[1].abs().max()))
            torch.testing.assert_close(a[1], g[1], rtol=1e-4, atol=1e-5 * scale, msg=lambda m: f"{tag} dgrid [{a[5]}]: {m}")
        if need[1]:
            torch.testing.assert_close(a[2], g[2], rtol=1e-4, atol=2e-5 * max(1.0, GD / 8.0),
                                       msg=lambda m: f"{tag} dguide [{a[5]}]: {m}")
        if need[2]:
            torch.testing.assert_close(a[3], g[3], rtol=1e-4, atol=1e-5, msg=lambda m: f"{tag} dinput [{a[5]}]: {m}")
    print(f"seed {seed}: kernels exercised: {sorted(kernels)}")


@pytest.mark.parametrize("seed", range(4))
def test_slice_fuzz_auto_against_generic(dev, ops, seed):
    rng = np.random.default_rng(2000 + seed)
    gen = torch.Generator(device=dev).manual_seed(2000 + seed)
    kernels = set()
    for case in range(25):
        B, H, W, GH, GW, GD, _, _, _, lo, hi = draw_case(rng)
        C = int(rng.choice([1, 2, 3, 4, 8, 12, 16, 20]))
        grid = torch.rand((B, GH, GW, GD, C), device=dev, generator=gen)
        guide = torch.rand((B, H, W), device=dev, generator=gen) * (hi - lo) + lo
        dout = torch.randn((B, H, W, C), device=dev, generator=gen)
        if rng.random() < 0.25:
            guide, dout = misaligned(guide, rng), misaligned(dout, rng)
        need = [bool(rng.integers(2)) for _ in range(2)]
        if not any(need):
            need[int(rng.integers(2))] = True
        tag = f"seed {seed} case {case}: {(B, H, W, GH, GW, GD, C)} guide [{lo}, {hi}] grads {need}"
        res = {}
        for which in ("generic", "auto"):
            tg, tgu = (t.detach().requires_grad_(n) for t, n in zip((grid, guide), need))
            with ops.kernel_override(which):
                out = ops.bilateral_slice(tg, tgu)
                kf = ops.last_kernel()
                out.backward(dout)
                kb = ops.last_kernel()
            res[which] = (out.detach(), tg.grad, tgu.grad, kf, kb)
        kernels.update((res["auto"][3], res["auto"][4]))
        a, g = res["auto"], res["generic"]
        torch.testing.assert_close(a[0], g[0], rtol=1e-5, atol=1e-5, msg=lambda m: f"{tag} fwd [{a[3]}]: {m}")
        if need[0]:
            scale = max(1.0, float(g[1].abs().max()))
            torch.testing.assert_close(a[1], g[1], rtol=1e-4, atol=1e-5 * scale, msg=lambda m: f"{tag} dgrid [{a[4]}]: {m}")
        if need[1]:
            torch.testing.assert_close(a[2], g[2], rtol=1e-4, atol=2e-5 * max(1.0, GD / 8.0) * max(1.0, C / 12.0),
                                       msg=lambda m: f"{tag} dguide [{a[4]}]: {m}")
    print(f"seed {seed}: kernels exercised: {sorted(kernels)}")
