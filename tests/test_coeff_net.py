"""The low-resolution coefficient network on the HIP kernels of csrc/coeff_net.hip (SURVEY.md section 8f row 1:
the caller of the hot path, hdrnet/models.py:62-142) against the torch restatement of the same graph
(hdrnet_amd/models.py: _Coefficients), evaluated in float64 on the CPU.

CPU part: the exported parameter layout (conv [Cout][kh][kw][Cin], fc [in][out], batch norm folded) reproduces the
module; workspace sizes and argument validation of the C-ABI entry points (no GPU call).
GPU part (-m gpu): the kernels themselves, over the hyper-parameters of hdrnet/bin/train.py:227-236.
"""
import copy
import ctypes

import pytest
import torch
import torch.nn.functional as F

from hdrnet_amd import models


def randomize(module, seed=0):
    """Move every parameter and batch-norm statistic off its initial value (zero biases, unit variances)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() == 1 and "bn.weight" not in name:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.3 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return module


def evaluate_exported(w, low):
    """The network evaluated from the EXPORTED arrays with plain torch ops -- the layout contract of
    include/hdrnet_amd.h (hdrnet_coeff_net) spelled out."""
    keep = iter(w._keep)

    def take(has_bias=True):
        wt = next(keep)
        return wt, (next(keep) if has_bias else None)

    def conv(x, wt, b, stride, relu):
        k = wt.shape[1]
        x = models.tf_same_pad(x, k, stride)
        y = F.conv2d(x, wt.permute(0, 3, 1, 2), b, stride=stride)  # [Cout][kh][kw][Cin] -> OIHW
        return F.relu(y) if relu else y

    x = low.permute(0, 3, 1, 2)
    for _ in range(w.n_splat):
        wt, b = take()
        x = conv(x, wt, b, 2, True)
    g = x
    for _ in range(2):
        wt, b = take()
        g = conv(g, wt, b, 2, True)
    g = g.permute(0, 2, 3, 1).reshape(g.shape[0], -1)
    for i in range(3):
        wt, b = take()
        g = g @ wt + b  # [in][out]
        if i < 2:
            g = F.relu(g)
    wt, b = take()
    loc = conv(x, wt, b, 1, True)
    wt, b = take(has_bias=False)
    loc = conv(loc, wt, None, 1, False)
    fusion = F.relu(loc + g[:, :, None, None])
    wt, b = take()
    pred = conv(fusion, wt, b, 1, False)
    B, _, GH, GW = pred.shape
    gd = w.params["luma_bins"]
    pred = pred.reshape(B, w.n_in, w.n_out, gd, GH, GW)
    return pred.permute(0, 4, 5, 3, 2, 1).contiguous()


@pytest.mark.parametrize("bn", [False, True])
def test_exported_layout_reproduces_the_module(bn):
    torch.manual_seed(3)
    m = randomize(models.HDRNetCurves(dict(batch_norm=bn, net_input_size=64, spatial_bin=8)).eval()).double()
    net = m.coefficients
    low = torch.rand(2, 64, 64, 3, dtype=torch.float64)
    with torch.no_grad():
        want = net(low)
        w = copy.deepcopy(net).float().exported()
        # the exported arrays are float32; evaluate them in float64
        w._keep = [t.double() for t in w._keep]
        got = evaluate_exported(w, low)
    assert got.shape == want.shape == (2, 8, 8, 8, 3, 4)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), float((got - want).abs().max())


def test_export_cache_follows_the_parameters():
    m = models.HDRNetCurves(dict(net_input_size=64, spatial_bin=8)).eval()
    net = m.coefficients
    a = net.exported()
    assert net.exported() is a
    with torch.no_grad():
        net.pred.conv.bias.add_(1.0)
    b = net.exported()
    assert b is not a
    assert torch.equal(b._keep[-1], net.pred.conv.bias)
    assert b._keep[-1].data_ptr() != net.pred.conv.bias.data_ptr()  # a snapshot, not an alias


def test_guide_fold_caches_follow_the_parameters():
    """Inference reuses the folded guide parameters (no re-fold launches per frame) until a parameter changes."""
    m = randomize(models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).eval(), seed=3)
    a = m.guide.folded()
    assert m.guide.folded() is a
    with torch.no_grad():
        m.guide.bn.running_mean.add_(0.5)
    b = m.guide.folded()
    assert b is not a and not torch.equal(a[0], b[0])
    state = {k: v.clone() for k, v in m.state_dict().items()}
    state["guide.w2"] = state["guide.w2"] * 2
    m.load_state_dict(state)
    c = m.guide.folded()
    assert torch.equal(c[1][:-1], 2 * b[1][:-1])
    # with autograd the fold stays attached to the parameters (and is not cached)
    d = m.guide.folded(detach=False)
    assert d[0].requires_grad and m.guide.folded() is c
    mc = models.HDRNetCurves().eval()
    e = mc.guide.exported()
    assert mc.guide.exported() is e
    with torch.no_grad():
        mc.guide.slopes.mul_(1.5)
    f = mc.guide.exported()
    assert f is not e and torch.equal(f[2], mc.guide.slopes.t())


def test_workspace_and_validation_without_gpu():
    from hdrnet_amd import _lib
    lib = _lib.load()
    w = models.HDRNetPointwiseNNGuide().eval().coefficients.exported()
    # 128^2*8 + 64^2*16 + 32^2*32 + 16^2*64 (splat) + 2 * 16^2*64 (local) + 8^2*64 + 4^2*64 (global convs)
    # + fc partial sums 64*256 (fc1) + 16*128 (fc2), floats per image
    per_image = 131072 + 65536 + 32768 + 16384 + 2 * 16384 + 4096 + 1024 + 16384 + 2048
    assert lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(w.net), 1) == 4 * per_image
    assert lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(w.net), 3) == 12 * per_image
    assert lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(w.net), 0) == 0
    assert lib.hdrnet_coefficients_workspace_bytes(None, 1) == 0
    assert lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(w.net), 65536) == 0  # the batch is a launch-grid extent
    assert lib.hdrnet_coefficients_f32(None, ctypes.byref(w.net), None, 65536, None, 0, None) == 1
    assert b"batch out of range" in lib.hdrnet_last_error()
    # luma_bins = 6: channel groups of 6, 12, ... are not whole power-of-two float4 groups -> unsupported
    w6 = models.HDRNetPointwiseNNGuide(dict(luma_bins=6)).eval().coefficients.exported()
    assert not w6.supported(1)
    assert lib.hdrnet_coefficients_f32(None, ctypes.byref(w6.net), None, 1, None, 0, None) == 1
    assert b"unsupported hyper-parameters" in lib.hdrnet_last_error()
    assert lib.hdrnet_coefficients_f32(None, None, None, 1, None, 0, None) == 1
    assert lib.hdrnet_coefficients_f32(None, ctypes.byref(w.net), None, 1, None, 0, None) == 1
    assert b"null buffer" in lib.hdrnet_last_error()
    assert lib.hdrnet_coefficients_f32(None, ctypes.byref(w.net), None, 0, None, 0, None) == 0  # empty batch: no-op
    # a missing parameter is an argument error, not a crash
    net = _lib.CoeffNet.from_buffer_copy(w.net)
    net.fc_w[1] = None
    assert lib.hdrnet_coefficients_f32(None, ctypes.byref(net), None, 1, None, 0, None) == 1
    assert b"null parameter" in lib.hdrnet_last_error()


def test_cpu_module_never_takes_the_native_path():
    m = models.HDRNetCurves(dict(net_input_size=64, spatial_bin=8)).eval()
    low = torch.rand(1, 64, 64, 3)
    with torch.no_grad():
        assert not m.coefficients._use_native(low)
        assert m.coefficients(low).shape == (1, 8, 8, 8, 3, 4)


# ---------------------------------------------------------------------------------------------------- GPU

CASES = {
    "default": (models.HDRNetPointwiseNNGuide, dict(), 1),
    "batch3": (models.HDRNetPointwiseNNGuide, dict(), 3),
    "batch_norm": (models.HDRNetPointwiseNNGuide, dict(batch_norm=True), 2),
    "curves": (models.HDRNetCurves, dict(batch_norm=True), 1),
    "pyramid": (models.HDRNetGaussianPyrNN, dict(), 2),
    "grid32": (models.HDRNetPointwiseNNGuide, dict(spatial_bin=32), 1),      # config #5's 32 x 32 grid: 3 splat layers
    "input512": (models.HDRNetPointwiseNNGuide, dict(net_input_size=512), 1),  # 5 splat layers, up to 128 channels
    "bins4": (models.HDRNetPointwiseNNGuide, dict(luma_bins=4), 1),           # 4, 8, 16, 32 channels
    "cm2": (models.HDRNetPointwiseNNGuide, dict(channel_multiplier=2), 1),    # 16 ... 128 channels: four staged chunks
    "grid8": (models.HDRNetPointwiseNNGuide, dict(spatial_bin=8, net_input_size=128), 1),  # global path 4 -> 2 cells
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_native_coefficients_vs_float64(case):
    cls, params, B = CASES[case]
    torch.manual_seed(11)
    m = randomize(cls(params).eval(), seed=5)
    N = m.params["net_input_size"]
    low = torch.rand(B, N, N, 3)
    with torch.no_grad():
        ref = copy.deepcopy(m.coefficients).double()(low.double())
        cpu32 = m.coefficients(low)
        md = m.to("cuda:0")
        net = md.coefficients
        assert net._use_native(low.cuda())
        got = net(low.cuda())
        net.native = False
        stock = net(low.cuda())
        net.native = True
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    err = float((got.cpu().double() - ref).abs().max())
    err_stock = float((stock.cpu().double() - ref).abs().max())
    err_cpu = float((cpu32.double() - ref).abs().max())
    print(f"{case}: |coeffs| <= {scale:.3g}; native {err:.3g}, stock GPU ops {err_stock:.3g}, torch CPU f32 {err_cpu:.3g} from float64")
    assert err <= 2e-6 * scale + 2.0 * max(err_stock, err_cpu), (err, err_stock, err_cpu, scale)
    assert err <= 1e-5 * scale


@pytest.mark.gpu
def test_native_levels_are_the_reference_slices():
    torch.manual_seed(2)
    m = randomize(models.HDRNetGaussianPyrNN().eval(), seed=9).to("cuda:0")
    low = torch.rand(2, 256, 256, 3, device="cuda:0")
    with torch.no_grad():
        coeffs = m.coefficients(low)  # [B, GH, GW, gd, 9, 4]
        lv = m.coefficients.levels(low)
    assert coeffs.shape == (2, 16, 16, 8, 9, 4) and len(lv) == 3
    for il in range(3):
        want = coeffs[:, :, :, :, 3 * il:3 * il + 3, :].reshape(2, 16, 16, 8, 12)
        assert lv[il].is_contiguous() and torch.equal(lv[il], want)


@pytest.mark.gpu
@pytest.mark.parametrize("cls", [models.HDRNetCurves, models.HDRNetPointwiseNNGuide, models.HDRNetGaussianPyrNN])
def test_model_inference_native_vs_stock_coefficients(cls):
    torch.manual_seed(4)
    m = randomize(cls(dict(batch_norm=True)).eval(), seed=1).to("cuda:0")
    low = torch.rand(1, 256, 256, 3, device="cuda:0")
    full = torch.rand(1, 272, 480, 3, device="cuda:0")
    with torch.no_grad():
        got = m(low, full)
        m.coefficients.native = False
        want = m(low, full)
        m.coefficients.native = True
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), float((got - want).abs().max())


@pytest.mark.gpu
def test_native_coefficients_under_hipgraph_and_determinism():
    from hdrnet_amd.runtime import GraphedInference
    torch.manual_seed(6)
    m = randomize(models.HDRNetPointwiseNNGuide().eval(), seed=2).to("cuda:0")
    low = torch.rand(1, 256, 256, 3, device="cuda:0")
    full = torch.rand(1, 272, 480, 3, device="cuda:0")
    with torch.no_grad():
        a = m.coefficients(low)
        b = m.coefficients(low)
        assert torch.equal(a, b)  # fixed summation order, no atomics
        eager = m(low, full)
        g = GraphedInference(m, [low, full])
        low2 = torch.rand_like(low)
        out = g(low2, full).clone()
        want = m(low2, full)
        assert torch.equal(out, want)
        assert not torch.equal(out, eager)
        # the graph holds the parameters of capture time: a replay after a change RAISES (it used to
        # serve the old folded weights silently), check_parameters=False keeps the old behaviour, recapture() picks it up
        unchecked = GraphedInference(m, [low, full], check_parameters=False)
        m.coefficients.pred.conv.bias.add_(0.25)
        m.guide.b2.add_(0.5)
        with pytest.raises(RuntimeError, match="recapture"):
            g(low2, full)
        stale = unchecked(low2, full).clone()
        assert torch.equal(stale, want)
        g.recapture()
        fresh = g(low2, full).clone()
        assert torch.equal(fresh, m(low2, full)) and not torch.equal(fresh, want)


@pytest.mark.gpu
def test_training_mode_and_autograd_stay_on_the_stock_ops():
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to("cuda:0")
    low = torch.rand(2, 256, 256, 3, device="cuda:0")
    m.train()
    assert not m.coefficients._use_native(low)
    m.eval()
    assert not m.coefficients._use_native(low)  # grad mode on, parameters require grad
    with torch.no_grad():
        assert m.coefficients._use_native(low)


TRAIN_CASES = {
    "pyramid_b2": (dict(_cls="pyramid"), 2),   # 288 prediction channels: four and a half 64-channel chunks
    "default_b4": (dict(), 4),
    "default_b1": (dict(), 1),
    "grid32_b2": (dict(spatial_bin=32), 2),
    "bins4_b3": (dict(luma_bins=4), 3),
    "small_b8": (dict(spatial_bin=8, net_input_size=128), 8),
    "cm2_b2": (dict(channel_multiplier=2), 2),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(TRAIN_CASES))
def test_native_training_gradients_vs_float64(case):
    """Forward + backward of the coefficient network (no batch norm) on the HIP kernels against torch autograd over the
    same module: every parameter gradient, judged against a float64 evaluation on the CPU."""
    params, B = TRAIN_CASES[case]
    params = dict(params)
    cls = models.HDRNetGaussianPyrNN if params.pop("_cls", "") == "pyramid" else models.HDRNetPointwiseNNGuide
    torch.manual_seed(21)
    m = randomize(cls(dict(batch_norm=False, **params)), seed=7)
    N = m.params["net_input_size"]
    low = torch.rand(B, N, N, 3)
    ref = copy.deepcopy(m.coefficients).double()
    out64 = ref(low.double())
    wts = torch.randn(out64.shape, dtype=torch.float64)
    (out64 * wts).sum().backward()
    net = m.coefficients.to("cuda:0")
    lowd, wd = low.cuda(), wts.float().cuda()
    assert net._use_native_training(lowd)
    out = net(lowd)
    assert out.grad_fn is not None and "CoefficientsTrain" in type(out.grad_fn).__name__
    (out * wd).sum().backward()
    native = {n: p.grad.clone() for n, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    net.native_training = False
    out_t = net(lowd)
    (out_t * wd).sum().backward()
    net.native_training = True
    scale_o = float(out64.abs().max())
    assert float((out.detach().cpu().double() - out64).abs().max()) <= 1e-5 * scale_o
    worst = 0.0
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        g64 = q.grad
        scale = float(g64.abs().max()) + 1e-30
        e_nat = float((native[name].cpu().double() - g64).abs().max()) / scale
        e_tor = float((p.grad.cpu().double() - g64).abs().max()) / scale
        worst = max(worst, e_nat)
        assert native[name].stride() == p.stride(), name
        assert e_nat <= 2e-5 + 2.0 * e_tor, (name, e_nat, e_tor)
    print(f"{case}: worst relative gradient error of the HIP path {worst:.2e}")


@pytest.mark.gpu
def test_native_training_is_deterministic_and_falls_back_with_batch_norm():
    torch.manual_seed(3)
    m = randomize(models.HDRNetPointwiseNNGuide(dict(batch_norm=False)), seed=1).to("cuda:0")
    low = torch.rand(4, 256, 256, 3, device="cuda:0")
    grads = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        m.coefficients(low).square().sum().backward()
        grads.append([p.grad.clone() for p in m.coefficients.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    mb = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to("cuda:0")
    assert not mb.coefficients._use_native_training(low)
    low.requires_grad_(True)
    assert not m.coefficients._use_native_training(low)  # the input's own gradient: torch ops


@pytest.mark.gpu
def test_native_training_writes_into_a_released_gradient_bucket():
    """With the gradients in a flat bucket (dist.GradBucket) and released (.grad = None), the backward kernels write every
    coefficient-network gradient straight into its segment: autograd adopts the alias (.grad's memory IS the segment),
    gather() has only the other parameters to copy, and the flat buffer equals the gradients of a run without a bucket.
    A parameter used TWICE in one backward is handed its segment once; the second gradient is added by autograd."""
    from hdrnet_amd import dist as hd
    torch.manual_seed(4)
    m = randomize(models.HDRNetPointwiseNNGuide(dict(batch_norm=False)), seed=2).to("cuda:0").train()
    low = torch.rand(2, 256, 256, 3, device="cuda:0")
    low2 = torch.rand(2, 256, 256, 3, device="cuda:0")
    params = list(m.coefficients.parameters())

    def plain(inputs):
        for p in params:
            p.grad = None
        sum(m.coefficients(x).square().sum() for x in inputs).backward()
        return [p.grad.clone() for p in params]

    want1, want2 = plain([low]), plain([low, low2])
    bucket = hd.GradBucket(params, align=4)
    for inputs, want in (([low], want1), ([low, low2], want2)):
        bucket.flat.fill_(float("nan"))
        bucket.release()
        sum(m.coefficients(x).square().sum() for x in inputs).backward()
        adopted = sum(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, bucket.views))
        if len(inputs) == 1:
            assert adopted == len(params), (adopted, len(params))  # nothing left for gather() to copy
        bucket.gather()
        assert bucket.attached()
        for p, v, w in zip(params, bucket.views, want):
            assert p.grad is v
            scale = float(w.abs().max()) + 1e-30
            assert float((v - w).abs().max()) <= 2e-6 * scale, float((v - w).abs().max()) / scale
    # bound gradients (no release): accumulation must not alias
    before = [v.clone() for v in bucket.views]
    m.coefficients(low).square().sum().backward()
    for v, b, w in zip(bucket.views, before, want1):
        scale = float((b + w).abs().max()) + 1e-30
        assert float((v - (b + w)).abs().max()) <= 4e-6 * scale
