"""hdrnet_amd/models.py: the closed-form pieces on CPU, the hot-path
composition and the training step on the GPU (BASELINE.json configs #3 / #4 at test size)."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from hdrnet_amd import models


def test_tf_same_padding_matches_tensorflow_rule():
    # TF SAME, k=3, s=2 on an even extent pads 0 before / 1 after; on an odd extent 1 / 1.
    x = torch.arange(36.0).reshape(1, 1, 6, 6)
    p = models.tf_same_pad(x, 3, 2)
    assert p.shape == (1, 1, 7, 7) and torch.equal(p[0, 0, :6, :6], x[0, 0]) and p[0, 0, 6].abs().sum() == 0
    x = torch.ones(1, 1, 5, 5)
    assert models.tf_same_pad(x, 3, 2).shape == (1, 1, 7, 7)
    assert models.tf_same_pad(x, 3, 1).shape == (1, 1, 7, 7)
    assert models.tf_same_pad(x, 1, 1).shape == (1, 1, 5, 5)
    # output extent = ceil(n / s)
    conv = models._Conv(1, 1, 3, stride=2)
    for n in (256, 255, 17, 16):
        assert conv(torch.zeros(1, 1, n, n)).shape[-1] == math.ceil(n / 2)


def test_coefficient_unroll_order():
    """conv channel (j * n_out + i) * gd + z  ->  coeffs[b, gy, gx, z, i, j]  (models.py:134-138)."""
    p = models.default_params()
    net = models._Coefficients(p, n_out=3, n_in=4)
    gd = p["luma_bins"]
    with torch.no_grad():
        net.pred.conv.weight.zero_()
        net.pred.conv.bias.copy_(torch.arange(gd * 12, dtype=torch.float32))
    with torch.no_grad():
        out = net(torch.rand(2, 256, 256, 3))
    assert out.shape == (2, 16, 16, 8, 3, 4)
    for i in range(3):
        for j in range(4):
            for z in range(gd):
                assert float(out[1, 5, 7, z, i, j]) == (j * 3 + i) * gd + z
    # numpy restatement of tf.stack(tf.split(.., 12, 3), 4) then tf.stack(tf.split(.., 4, 4), 5)
    t = np.arange(gd * 12, dtype=np.float32)[None, None, None, :]
    a = np.stack(np.split(t, 12, axis=3), axis=4)
    b = np.stack(np.split(a, 4, axis=4), axis=5)
    assert np.array_equal(b[0, 0, 0], out[0, 0, 0].numpy())


def test_parameter_count_and_shapes():
    m = models.HDRNetPointwiseNNGuide()
    n = sum(p.numel() for p in m.parameters() if p.requires_grad)
    assert 4.5e5 < n < 5.2e5, n  # ~482 k parameters at cm=1, gd=8
    assert models.HDRNetGaussianPyrNN().coefficients.pred.conv.out_channels == 8 * 9 * 4


def test_guides_match_their_formulas():
    torch.manual_seed(0)
    im = torch.rand(2, 5, 7, 3)
    g = models._CurvesGuide()
    out = g(im)
    # at initialisation: ccm ~ I, curve = relu(x - 0) (slope 1 on the first knot only), mixing = mean
    assert torch.allclose(out, im.mean(-1).clamp(0, 1), atol=2e-3)
    pw = models._PointwiseNNGuide(16).eval()
    with torch.no_grad():
        pw.bn.running_mean.normal_()
        pw.bn.running_var.uniform_(0.5, 2.0)
        pw.bn.bias.normal_()
    h = im.numpy() @ pw.w1.detach().numpy()
    h = (h - pw.bn.running_mean.numpy()) / np.sqrt(pw.bn.running_var.numpy() + 1e-3) + pw.bn.bias.detach().numpy()
    ref = 1.0 / (1.0 + np.exp(-(np.maximum(h, 0) @ pw.w2.detach().numpy() + float(pw.b2.detach()))))
    assert np.allclose(pw(im).detach().numpy(), ref, atol=1e-5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hdrnet_amd import dist as hd
    from hdrnet_amd import models as M
    hd.init(backend="gloo")
    torch.manual_seed(0)  # same weights everywhere
    net = M._Coefficients(M.default_params(), 3, 4)
    torch.manual_seed(100 + rank)  # different data per rank
    x = torch.rand(2, 256, 256, 3)
    net(x).square().mean().backward()
    local = [p.grad.clone() for p in net.parameters() if p.grad is not None]
    n = hd.allreduce_gradients_flat(net.parameters(), world)
    # reference: gather every rank's local grads and average
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, local)
    ok = True
    for k, p in enumerate([p for p in net.parameters() if p.grad is not None]):
        want = sum(g[k] for g in gathered) / world
        ok = ok and torch.allclose(p.grad, want, rtol=1e-5, atol=1e-7)
    if rank == 0:
        q.put((ok, n))
    hd.barrier()
    torch.distributed.destroy_process_group()


def _toy_net():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def test_grad_bucket_views_survive_backward_and_step():
    """dist.GradBucket: every .grad is a view of ONE flat buffer, autograd accumulates into the views in place,
    zero_() is one memset, and runtime.TrainStep on it takes exactly the steps of the plain torch loop."""
    from hdrnet_amd import dist as hd
    from hdrnet_amd.runtime import TrainStep
    net, ref = _toy_net(), _toy_net()
    net[0].bias.requires_grad_(False)  # a frozen parameter stays outside the bucket (the models' BN scale)
    ref[0].bias.requires_grad_(False)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    ropt = torch.optim.SGD([p for p in ref.parameters() if p.requires_grad], lr=0.1)
    step = TrainStep(net, lambda out, tgt: (out - tgt).square().mean(), opt)
    b = step.bucket
    assert b.flat.numel() == sum(p.numel() for p in net.parameters() if p.requires_grad) and net[0].bias.grad is None
    assert b.attached() and not step.distributed
    torch.manual_seed(1)
    for _ in range(3):
        x, y = torch.rand(5, 6), torch.rand(5, 3)
        loss = step([x], [y])
        ropt.zero_grad(set_to_none=True)
        rl = (ref(x) - y).square().mean()
        rl.backward()
        ropt.step()
        assert torch.equal(loss, rl) and b.attached()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.equal(p, q)
            if p.requires_grad:
                assert torch.equal(p.grad, q.grad)
    assert b.allreduce() == b.flat.numel()  # no process group: a no-op
    b.zero_()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters() if p.requires_grad)
    opt.zero_grad(set_to_none=True)  # what the bucket's user must NOT do: the views are gone
    assert not b.attached()
    # a channels_last convolution weight (what the coefficient network's weights are on the GPU): the view takes the
    # parameter's strides -- autograd's layout contract, and what the fused Adam requires
    conv = torch.nn.Conv2d(8, 16, 3).to(memory_format=torch.channels_last)
    cb = hd.GradBucket(conv.parameters())
    assert conv.weight.grad.stride() == conv.weight.stride() != conv.weight.contiguous().stride()
    conv(torch.rand(2, 8, 6, 6)).sum().backward()
    assert cb.attached() and float(cb.flat.abs().sum()) > 0


def _train_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hdrnet_amd import dist as hd
    from hdrnet_amd.runtime import TrainStep
    hd.init(backend="gloo")
    net = _toy_net()  # same weights on every rank
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    step = TrainStep(net, lambda out, tgt: (out - tgt).square().mean(), opt)
    assert step.distributed
    torch.manual_seed(100 + rank)  # different data per rank
    data = [(torch.rand(5, 6), torch.rand(5, 3)) for _ in range(3)]
    for x, y in data:
        step([x], [y])
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, data)
    if rank == 0:
        # reference: one process, the ranks' losses averaged = the mean gradient
        ref = _toy_net()
        ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
        for k in range(3):
            ropt.zero_grad(set_to_none=True)
            (sum((ref(gathered[r][k][0]) - gathered[r][k][1]).square().mean() for r in range(world)) / world).backward()
            ropt.step()
        ok = all(torch.allclose(p, r, rtol=1e-5, atol=1e-7) for p, r in zip(net.parameters(), ref.parameters()))
        q.put((ok, step.bucket.attached(), step.bucket.flat.numel()))
    hd.barrier()
    torch.distributed.destroy_process_group()


def test_train_step_distributed_branch_gloo():
    """runtime.TrainStep at world size 2 (gloo): forward + backward into the flat bucket, ONE in-place all-reduce
    (averaged), optimizer step -- the parameters follow the single-process run on the union of the ranks' data.
    GraphedTrainStep shares this code path after its graph replay (runtime.py)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ok, attached, n = q.get()
    assert ok and attached and n == 6 * 16 + 16 + 16 * 3 + 3


def test_flat_bucket_gradient_allreduce_gloo():
    """The training step's one collective: a single flat fp32 bucket, 2 ranks, gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ok, n = q.get()
    assert ok and 4.0e5 < n < 5.2e5


# ---- GPU: the model drives the HIP hot path ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cls", ["HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN"])
def test_model_inference_composes_with_oracle(cls, port):
    """coeffs / guide from the torch graph, slice-apply from the HIP kernel == the same coeffs /
    guide pushed through the CPU oracle (pins the hot path inside each model graph)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = getattr(models, cls)().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 270, 480, 3, device=dev)
    with torch.no_grad():
        out = m(low, full)
        coeffs = m.coefficients(low)
    assert out.shape == (1, 270, 480, 3) and torch.isfinite(out).all()
    if cls != "HDRNetGaussianPyrNN":
        with torch.no_grad():
            guide = m.guide(full)
        g5 = coeffs.reshape(1, 16, 16, 8, 12).cpu().numpy()
        want = port.bilateral_slice_apply(g5, guide.cpu().numpy(), full.cpu().numpy(), True)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_config3_full_inference_4k():
    """BASELINE.json configs[2]: HDRNetPointwiseNNGuide, 3840x2160, batch 1, one MI355X."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = models.HDRNetPointwiseNNGuide().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 2160, 3840, 3, device=dev)
    with torch.no_grad():
        out = m(low, full)
    from hdrnet_amd import hdrnet_ops
    assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4+nnguide"  # guide net fused in eval mode
    assert out.shape == (1, 2160, 3840, 3) and torch.isfinite(out).all()
    m.fuse_guide = False
    with torch.no_grad():
        ref = m(low, full)
    assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4"
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_config4_training_step_reduces_loss():
    """BASELINE.json configs[3] at test size: fwd + bwd through the HIP VJPs + Adam; the L2 loss
    (hdrnet/metrics.py:21-24) to a fixed target falls."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    low = torch.rand(4, 256, 256, 3, device=dev)
    full = F.interpolate(low.permute(0, 3, 1, 2), size=(270, 480), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    target = (full * 0.8 + 0.1).clamp(0, 1)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = (m(low, full) - target).square().mean()
        loss.backward()
        from hdrnet_amd import dist as hd
        hd.allreduce_gradients_flat(m.parameters())  # no-op at world size 1
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], losses[::5]


@pytest.mark.gpu
def test_graphed_inference_replays_exactly():
    """hipGraph capture of the whole inference (torch ops + the HIP kernels through ctypes)."""
    from hdrnet_amd.runtime import GraphedInference
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = models.HDRNetPointwiseNNGuide().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 540, 960, 3, device=dev)
    g = GraphedInference(m, [low, full])
    for seed in (5, 6):
        torch.manual_seed(seed)
        low2, full2 = torch.rand_like(low), torch.rand_like(full)
        with torch.no_grad():
            want = m(low2, full2)
        got = g(low2, full2).clone()
        torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError):
        g(low, torch.rand(1, 540, 964, 3, device=dev))


@pytest.mark.gpu
def test_frame_pipeline_matches_sequential():
    """Independent frames round-robin over 2 / 3 streams (runtime.FramePipeline): the same bits as one stream, for
    the bare op and for graph-captured whole inferences with per-lane static buffers."""
    from hdrnet_amd import hdrnet_ops as ops
    from hdrnet_amd.runtime import FramePipeline, GraphedInference
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    grid = torch.rand(1, 16, 16, 8, 12, device=dev)
    frames = [(torch.rand(1, 270, 480, device=dev), torch.rand(1, 270, 480, 3, device=dev)) for _ in range(7)]
    want = [ops.bilateral_slice_apply(grid, g, i, has_offset=True) for g, i in frames]
    for depth in (1, 2, 3):
        pipe = FramePipeline(lambda: (lambda g, i: ops.bilateral_slice_apply(grid, g, i, has_offset=True)), depth=depth)
        got = []
        tickets = []
        for k, (g, i) in enumerate(frames):
            tickets.append(pipe.submit(g, i))
            if len(tickets) == depth:  # collect the oldest frame before its lane is reused
                got.append(pipe.result(tickets.pop(0)).clone())
        while tickets:
            got.append(pipe.result(tickets.pop(0)).clone())
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    m = models.HDRNetPointwiseNNGuide().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 270, 480, 3, device=dev)
    pipe = FramePipeline(lambda: GraphedInference(m, [low, full]), depth=2)
    ins = [(torch.rand_like(low), torch.rand_like(full)) for _ in range(5)]
    # against ONE captured graph on one stream.  (Two captures of the same module are not bit-identical -- the stock
    # convolutions may pick other algorithms / workspaces per capture -- so this half is held to rounding; a race between
    # lanes would show as whole wrong frames.  The bare op above is held to the bit.)
    one = GraphedInference(m, [low, full])
    ref = [one(a, b).clone() for a, b in ins]
    outs = []
    prev = None
    for a, b in ins:
        t = pipe.submit(a, b)
        if prev is not None:
            outs.append(pipe.result(prev).clone())
        prev = t
    outs.append(pipe.result(prev).clone())
    for a, b in zip(outs, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        torch.testing.assert_close(outs[-1], m(*ins[-1]), rtol=1e-5, atol=1e-5)  # and the eager module, to rounding
    with pytest.raises(ValueError):
        FramePipeline(lambda: None, depth=0)


@pytest.mark.gpu
def test_frame_pipeline_inputs_dropped_after_submit():
    """The streaming pattern FramePipeline is built for: the caller lets go of a frame's inputs right after
    submit() and allocates the next frame -- the caching allocator would hand the SAME blocks out again on the
    caller's stream while the lane's stream still reads them unless submit() records the lane's use of them
    (ADVICE r03).  Big frames (the kernel runs for tens of microseconds), every freed block immediately
    re-allocated and overwritten with garbage on the caller's stream."""
    from hdrnet_amd import hdrnet_ops as ops
    from hdrnet_amd.runtime import FramePipeline
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    grid = torch.rand(1, 16, 16, 8, 12, device=dev, generator=gen)
    H, W, n = 1080, 1920, 12
    seeds = list(range(100, 100 + n))

    def frame(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        return torch.rand(1, H, W, device=dev, generator=g), torch.rand(1, H, W, 3, device=dev, generator=g)

    want = []
    for sd in seeds:
        gu, inp = frame(sd)
        want.append(ops.bilateral_slice_apply(grid, gu, inp, has_offset=True).cpu())
        del gu, inp
    torch.cuda.synchronize()
    for depth in (2, 3):
        pipe = FramePipeline(lambda: (lambda g, i: ops.bilateral_slice_apply(grid, g, i, has_offset=True)), depth=depth)
        got, tickets = [], []
        for sd in seeds:
            gu, inp = frame(sd)
            tickets.append(pipe.submit(gu, inp))
            ptrs = (gu.data_ptr(), inp.data_ptr())
            del gu, inp  # back to the allocator while the lane may still be reading them
            junk = [torch.full((1, H, W), float("nan"), device=dev), torch.full((1, H, W, 3), float("nan"), device=dev)]
            # with record_stream the allocator must NOT have reused the blocks the lane is still reading
            del junk, ptrs
            if len(tickets) == depth:
                got.append(pipe.result(tickets.pop(0)).cpu())
        while tickets:
            got.append(pipe.result(tickets.pop(0)).cpu())
        for k, (a, b) in enumerate(zip(got, want)):
            assert torch.equal(a, b), f"depth {depth}: frame {k} corrupted"


# ---- training side of the fused guide network (extended to training) ----
def _torch_guide(inp, conv1, conv2):
    """fp32 torch statement of the folded guide network (hdrnet/models.py:203-210)."""
    h = inp @ conv1[:, :-1].t() + conv1[:, -1]
    return torch.sigmoid(torch.relu(h) @ conv2[:-1] + conv2[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 37, 52, 3, 16), (1, 64, 128, 3, 8), (1, 21, 36, 1, 4)])
def test_input_moments(shape):
    from hdrnet_amd import hdrnet_ops
    B, H, W, Cin, _ = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    x = torch.rand(B, H, W + 1, Cin, device=dev)[:, :, :W].contiguous()  # ragged pixel count
    sums, mom = hdrnet_ops.input_moments(x)
    assert hdrnet_ops.last_kernel() == "input_moments"
    flat = x.reshape(-1, Cin).double()
    torch.testing.assert_close(sums.double(), flat.sum(0), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(mom.double(), flat.t() @ flat, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 36, 52, 3, 16), (1, 64, 128, 3, 8), (1, 20, 36, 1, 4)])
def test_fused_guide_apply_gradients_match_composition(shape):
    """d(out)/d(grid, input, conv1, conv2) of the ONE fused differentiable op == autograd through
    [torch guide network] -> [bilateral_slice_apply] (whose VJPs the parity tests pin to the oracle)."""
    from hdrnet_amd import hdrnet_ops
    B, H, W, Cin, n = shape
    Cout = Cin
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    grid = torch.rand(B, 8, 8, 8, Cout * (Cin + 1), device=dev)
    inp = torch.rand(B, H, W, Cin, device=dev)
    conv1 = torch.randn(n, Cin + 1, device=dev) * 0.8
    conv2 = torch.randn(n + 1, device=dev) * 0.5
    dout = torch.randn(B, H, W, Cout, device=dev)

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (grid, inp, conv1, conv2)]
        g, x, c1, c2 = leaves
        if fused:
            out = hdrnet_ops.bilateral_slice_apply_nnguide(g, x, c1, c2, has_offset=True)
        else:
            out = hdrnet_ops.bilateral_slice_apply(g, _torch_guide(x, c1, c2), x, has_offset=True)
        out.backward(dout)
        return out.detach(), [t.grad for t in leaves]

    out_f, grads_f = run(True)
    assert hdrnet_ops.last_kernel() == "guide_nn_grad"
    out_r, grads_r = run(False)
    torch.testing.assert_close(out_f, out_r, rtol=2e-5, atol=2e-5)
    for name, a, b in zip(("dgrid", "dinput", "dconv1", "dconv2"), grads_f, grads_r):
        scale = b.abs().max().item()
        err = (a - b).abs().max().item()
        assert err <= 2e-4 * scale + 1e-5, (name, err, scale)

    # parameters only (the training case: the image needs no gradient): dinput is skipped
    g = grid.clone().requires_grad_(True)
    c1 = conv1.clone().requires_grad_(True)
    out = hdrnet_ops.bilateral_slice_apply_nnguide(g, inp, c1, conv2, has_offset=True)
    out.backward(dout)
    torch.testing.assert_close(c1.grad, grads_f[2], rtol=1e-5, atol=1e-6)
    # and the run is deterministic
    _, again = run(True)
    for a, b in zip(grads_f, again):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_training_fused_guide_matches_unfused_module():
    """HDRNetPointwiseNNGuide.train(): the fused path (batch statistics from the input's moments,
    fused forward, guide-network VJP kernel) == the composed torch graph: loss, every parameter
    gradient, and the batch-norm running statistics."""
    from hdrnet_amd import hdrnet_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    ref = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    ref.load_state_dict(m.state_dict())
    ref.fuse_guide = False
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 136, 240, 3, device=dev)
    target = torch.rand(2, 136, 240, 3, device=dev)
    loss = (m(low, full) - target).square().mean()
    assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4+nnguide"
    loss.backward()
    loss_ref = (ref(low, full) - target).square().mean()
    loss_ref.backward()
    torch.testing.assert_close(loss, loss_ref, rtol=1e-5, atol=1e-7)
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        scale = q.grad.abs().max().item()
        err = (p.grad - q.grad).abs().max().item()
        assert err <= 1e-3 * scale + 1e-7, (name, err, scale)
    torch.testing.assert_close(m.guide.bn.running_mean, ref.guide.bn.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(m.guide.bn.running_var, ref.guide.bn.running_var, rtol=1e-4, atol=1e-6)
    assert int(m.guide.bn.num_batches_tracked) == int(ref.guide.bn.num_batches_tracked) == 1


@pytest.mark.gpu
def test_graphed_train_step_matches_eager():
    """hipGraph capture of a whole training step (fwd + loss + bwd through the HIP VJPs + optimizer
    update): the parameter updates of three replays equal those of three eager steps.  Small
    plain-SGD steps keep the comparison well-conditioned: the coefficient network's MIOpen /
    rocBLAS backward kernels are not run-to-run deterministic, and at a training-size learning rate
    two EAGER runs of this problem already differ by 10 % in loss after five steps."""
    from hdrnet_amd.runtime import GraphedTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 136, 240, 3, device=dev)
    target = torch.rand(2, 136, 240, 3, device=dev)

    def loss_fn(out, tgt):
        return (out - tgt).square().mean()

    m0 = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    state = {k: v.clone() for k, v in m0.state_dict().items()}

    def make():
        m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
        m.load_state_dict(state)
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-5)
        return m, opt

    me, oe = make()
    for _ in range(2 + 3):  # GraphedTrainStep warms up with 2 eager steps before capturing
        oe.zero_grad(set_to_none=True)
        le = loss_fn(me(low, full), target)
        le.backward()
        oe.step()
    mg, og = make()
    gstep = GraphedTrainStep(mg, loss_fn, og, [low, full], [target], warmup=2)
    for _ in range(3):
        lg = gstep([low, full], [target])
    torch.testing.assert_close(lg, le, rtol=1e-3, atol=1e-6)
    for (name, p), (_, q) in zip(mg.named_parameters(), me.named_parameters()):
        if not p.requires_grad:
            continue
        p0 = state[name]
        dg, de = p.detach() - p0, q.detach() - p0
        scale = de.abs().max().item()
        assert scale > 0, name
        assert (dg - de).abs().max().item() <= 5e-2 * scale, (name, (dg - de).abs().max().item(), scale)


@pytest.mark.gpu
def test_graphed_train_step_split_form_matches_one_graph():
    """GraphedTrainStep's multi-rank structure -- graph = forward + backward into the flat gradient bucket, then the
    (here: no-op) all-reduce and the optimizer step outside the graph -- against the one-graph form, same data."""
    from hdrnet_amd.runtime import GraphedTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 136, 240, 3, device=dev)
    target = torch.rand(2, 136, 240, 3, device=dev)
    m0 = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
    state = {k: v.clone() for k, v in m0.state_dict().items()}
    res = []
    for split in (False, True):
        m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev).train()
        m.load_state_dict(state)
        opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-5)
        g = GraphedTrainStep(m, lambda out, tgt: (out - tgt).square().mean(), opt, [low, full], [target], warmup=2,
                             flat_bucket=split)
        assert g.split == split and g.bucket.attached()
        for _ in range(3):
            loss = g([low, full], [target])
        assert g.bucket.attached()
        res.append((loss.clone(), {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-3, atol=1e-6)
    for n in res[0][1]:
        d0, d1 = res[0][1][n] - state[n], res[1][1][n] - state[n]
        scale = d0.abs().max().item()
        assert scale > 0 and (d0 - d1).abs().max().item() <= 5e-2 * scale, n


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 17, 23, 34, 46, 3), (1, 5, 7, 11, 13, 3), (2, 8, 8, 16, 16, 1), (1, 1, 4, 3, 9, 3),
                                   (1, 12, 10, 6, 5, 3), (1, 6, 6, 6, 6, 5)])
def test_upsample_add_and_its_vjp_vs_torch_float64(shape):
    """hdrnet_ops.upsample_add = resize(coarse, align_corners) + fine (hdrnet/models.py:283-287) and its two gradients
    against torch's interpolate in float64: up-sampling by two and by odd ratios, one source row, DOWN-sampling, equal
    sizes, a channel count without a specialisation; the gather-form transpose is bit-reproducible."""
    from hdrnet_amd import hdrnet_ops
    B, ih, iw, oh, ow, C = shape
    torch.manual_seed(sum(shape))
    coarse = torch.randn(B, ih, iw, C, device="cuda:0", requires_grad=True)
    fine = torch.randn(B, oh, ow, C, device="cuda:0", requires_grad=True)
    w = torch.randn(B, oh, ow, C, device="cuda:0")
    out = hdrnet_ops.upsample_add(coarse, fine)
    (out * w).sum().backward()
    c64 = coarse.detach().double().requires_grad_(True)
    f64 = fine.detach().double().requires_grad_(True)
    ref = F.interpolate(c64.permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear", align_corners=True).permute(0, 2, 3, 1) + f64
    (ref * w.double()).sum().backward()
    assert torch.allclose(out.double(), ref, rtol=0, atol=1e-5)
    assert torch.equal(fine.grad, w)
    scale = float(c64.grad.abs().max()) + 1e-30
    assert float((coarse.grad.double() - c64.grad).abs().max()) <= 2e-6 * scale + 1e-6
    g1 = coarse.grad.clone()
    coarse.grad = None
    (hdrnet_ops.upsample_add(coarse, fine.detach()) * w).sum().backward()
    assert torch.equal(coarse.grad, g1)
    with pytest.raises(ValueError):
        hdrnet_ops.upsample_add(coarse, fine[..., :1] if C > 1 else fine.repeat(1, 1, 1, 2))


@pytest.mark.gpu
def test_graphed_train_step_double_buffered_feed_equals_the_serial_feed():
    """GraphedTrainStep(feeds=2): batches staged by prefetch() on the copy stream into alternating buffer sets, each with
    its own captured graph, give bit-for-bit the losses of the serial feed over the same sequence of batches; misuse
    raises."""
    from hdrnet_amd import metrics, optim
    from hdrnet_amd.runtime import GraphedTrainStep
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(9)
    batches = [(torch.rand(2, 256, 256, 3, device=dev, generator=gen), torch.rand(2, 136, 240, 3, device=dev, generator=gen),
                torch.rand(2, 136, 240, 3, device=dev, generator=gen)) for _ in range(3)]
    losses = []
    for feeds in (1, 2):
        torch.manual_seed(12)
        m = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).train()
        opt = optim.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        low, full, tgt = batches[0]
        g = GraphedTrainStep(m, lambda out, t: metrics.l2_loss(t, out), opt, [low, full], [tgt], warmup=2, feeds=feeds)
        got = []
        if feeds == 1:
            with pytest.raises(RuntimeError, match="feeds=2"):
                g.prefetch([low, full], [tgt])
            for k in range(5):
                low, full, tgt = batches[k % 3]
                got.append(float(g([low, full], [tgt]).detach()))
        else:
            with pytest.raises(RuntimeError, match="no staged batch"):
                g.step()
            g.prefetch([low, full], [tgt])
            for k in range(5):
                if k < 4:
                    low, full, tgt = batches[(k + 1) % 3]
                    g.prefetch([low, full], [tgt])
                if k == 0:
                    with pytest.raises(RuntimeError, match="every buffer set"):
                        g.prefetch([low, full], [tgt])
                    with pytest.raises(RuntimeError, match="pending"):
                        g([low, full], [tgt])
                got.append(float(g.step().detach()))
        assert g.bucket.attached()
        losses.append(got)
    assert losses[0] == losses[1], losses


@pytest.mark.gpu
def test_pyramid_model_fused_matches_composed():
    """HDRNetGaussianPyrNN inference: the fused path (resize kernel, per-level guide net + slice-apply
    + up-add in one launch) == the composition of the un-fused ops (hdrnet/models.py:213-289)."""
    from hdrnet_amd import hdrnet_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    m = models.HDRNetGaussianPyrNN().to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 272, 480, 3, device=dev)
    with torch.no_grad():
        out = m(low, full)
        assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4+nnguide+upadd"
        m.fuse_guide = False
        ref = m(low, full)
        assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4"
    torch.testing.assert_close(out, ref, rtol=5e-5, atol=5e-5)


@pytest.mark.gpu
def test_pyramid_training_fused_matches_composed():
    """HDRNetGaussianPyrNN.train(): per-level fused differentiable guide + slice-apply == the graph
    composed from torch ops (loss and every parameter gradient)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    m = models.HDRNetGaussianPyrNN(dict(batch_norm=True)).to(dev).train()
    ref = models.HDRNetGaussianPyrNN(dict(batch_norm=True)).to(dev).train()
    ref.load_state_dict(m.state_dict())
    ref.fuse_guide = False
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 144, 256, 3, device=dev)
    target = torch.rand(2, 144, 256, 3, device=dev)
    loss = (m(low, full) - target).square().mean()
    loss.backward()
    loss_ref = (ref(low, full) - target).square().mean()
    loss_ref.backward()
    torch.testing.assert_close(loss, loss_ref, rtol=2e-5, atol=1e-7)
    checked = 0
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        scale = q.grad.abs().max().item()
        err = (p.grad - q.grad).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked > 20


@pytest.mark.gpu
def test_curves_model_fused_matches_composed():
    """HDRNetCurves inference (the reference's default model): curves guide evaluated inside the
    slice-apply kernel == guide module + slice-apply op."""
    from hdrnet_amd import hdrnet_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    m = models.HDRNetCurves().to(dev).eval()
    with torch.no_grad():  # move the guide away from its (trivial) initial state
        m.guide.ccm.add_(torch.randn(3, 3, device=dev) * 0.2)
        m.guide.slopes.add_(torch.randn(3, 16, device=dev) * 0.2)
        m.guide.mix_b.add_(0.03)
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 270, 480, 3, device=dev)
    with torch.no_grad():
        out = m(low, full)
        assert hdrnet_ops.last_kernel() == "apply_fwd_io/f32->f32+curvesguide/cells"  # the prepared tables (round 5)
        m.fuse_guide = False
        ref = m(low, full)
        assert hdrnet_ops.last_kernel() == "apply_fwd_seg/vec4"
    torch.testing.assert_close(out, ref, rtol=3e-5, atol=3e-5)
    # the inference forward used the curves' PREPARED lookup tables (round 5), cached per parameter state: a changed knot
    # set must rebuild them (and give the same result as without them: prepare_curves = False)
    assert m.prepare_curves and getattr(m.guide, "_prepared_cache", None) is not None
    with torch.no_grad():
        m.guide.shifts.add_(torch.randn(3, 16, device=dev) * 0.004)  # (knots stay more than a cell apart)
        del m.fuse_guide
        out2 = m(low, full)
        assert hdrnet_ops.last_kernel() == "apply_fwd_io/f32->f32+curvesguide/cells"
        m.prepare_curves = False
        out3 = m(low, full)
        m.fuse_guide = False
        ref2 = m(low, full)
    assert (out2 - out).abs().max() > 1e-5
    torch.testing.assert_close(out2, ref2, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(out2, out3, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 36, 52), (1, 31, 64)])
def test_fused_curves_apply_gradients_match_composition(shape):
    """All six gradients of the ONE fused differentiable curves-guide + slice-apply op == autograd
    through [torch curves guide] -> [bilateral_slice_apply] (ragged pixel count, guide values that
    hit the clip on both sides)."""
    from hdrnet_amd import hdrnet_ops
    B, H, W = shape
    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    grid = torch.rand(B, 8, 8, 8, 12, device=dev)
    inp = torch.rand(B, H, W, 3, device=dev)
    ccm = torch.cat([torch.eye(3, device=dev), torch.zeros(3, 1, device=dev)], 1) + 0.3 * torch.randn(3, 4, device=dev)
    shifts = torch.linspace(0, 1, 17, device=dev)[:-1, None].repeat(1, 3) + 0.013 * torch.randn(16, 3, device=dev)
    slopes = 0.4 * torch.randn(16, 3, device=dev)
    slopes[0] += 1.0
    mix = torch.tensor([0.9, 0.8, 0.7, -0.4], device=dev)  # pre-clip range ~[-0.4, 2]: clips on both sides
    dout = torch.randn(B, H, W, 3, device=dev)

    def torch_guide(x, c, sh, sl, mx):
        t = (x.unsqueeze(-2) * c[:, :3]).sum(-1) + c[:, 3]
        cv = (sl.t() * torch.relu(t.unsqueeze(-1) - sh.t())).sum(-1)
        return ((cv * mx[:3]).sum(-1) + mx[3]).clamp(0.0, 1.0)

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (grid, inp, ccm, shifts, slopes, mix)]
        g, x, c, sh, sl, mx = leaves
        if fused:
            out = hdrnet_ops.bilateral_slice_apply_curves(g, x, c, sh, sl, mx, has_offset=True)
        else:
            out = hdrnet_ops.bilateral_slice_apply(g, torch_guide(x, c, sh, sl, mx), x, has_offset=True)
        out.backward(dout)
        return out.detach(), [t.grad for t in leaves]

    out_f, grads_f = run(True)
    assert hdrnet_ops.last_kernel() == "curves_guide_grad"
    out_r, grads_r = run(False)
    frac_clipped = float(((torch_guide(inp, ccm, shifts, slopes, mix) <= 0) |
                          (torch_guide(inp, ccm, shifts, slopes, mix) >= 1)).float().mean())
    assert 0.02 < frac_clipped < 0.9, frac_clipped
    torch.testing.assert_close(out_f, out_r, rtol=3e-5, atol=3e-5)
    for name, a, b in zip(("dgrid", "dinput", "dccm", "dshifts", "dslopes", "dmix"), grads_f, grads_r):
        scale = b.abs().max().item()
        err = (a - b).abs().max().item()
        assert err <= 3e-4 * scale + 1e-5, (name, err, scale)
    _, again = run(True)
    for a, b in zip(grads_f, again):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_training_fused_curves_matches_unfused_module():
    """HDRNetCurves.train() (the reference's default model): fused forward + curves VJP kernel == the
    composed torch graph: loss and every parameter gradient."""
    dev = torch.device("cuda:0")
    torch.manual_seed(14)
    m = models.HDRNetCurves(dict(batch_norm=True)).to(dev).train()
    with torch.no_grad():
        m.guide.ccm.add_(torch.randn(3, 3, device=dev) * 0.2)
        m.guide.slopes.add_(torch.randn(3, 16, device=dev) * 0.2)
    ref = models.HDRNetCurves(dict(batch_norm=True)).to(dev).train()
    ref.load_state_dict(m.state_dict())
    ref.fuse_guide = False
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 136, 240, 3, device=dev)
    target = torch.rand(2, 136, 240, 3, device=dev)
    loss = (m(low, full) - target).square().mean()
    loss.backward()
    loss_ref = (ref(low, full) - target).square().mean()
    loss_ref.backward()
    torch.testing.assert_close(loss, loss_ref, rtol=2e-5, atol=1e-7)
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        scale = q.grad.abs().max().item()
        err = (p.grad - q.grad).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (name, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("cls,train", [("HDRNetPointwiseNNGuide", True), ("HDRNetPointwiseNNGuide", False),
                                       ("HDRNetGaussianPyrNN", True), ("HDRNetGaussianPyrNN", False),
                                       ("HDRNetCurves", True)])
def test_input_gradient_fused_equals_composed(cls, train):
    """d loss / d fullres_input must be COMPLETE whichever path the module takes: the fused
    training paths treat the batch-norm statistics as constants and build the pyramid outside autograd,
    so when the full-resolution input itself requires a gradient the modules must route to a path that
    carries every term.  fuse_guide = True vs False: same loss, same input gradient, same parameter
    gradients."""
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    m = getattr(models, cls)(dict(batch_norm=True)).to(dev)
    ref = getattr(models, cls)(dict(batch_norm=True)).to(dev)
    ref.load_state_dict(m.state_dict())
    ref.fuse_guide = False
    m.train(train)
    ref.train(train)
    low = torch.rand(2, 256, 256, 3, device=dev)
    full0 = torch.rand(2, 48, 64, 3, device=dev)
    target = torch.rand(2, 48, 64, 3, device=dev)
    grads = []
    for mod in (m, ref):
        full = full0.clone().requires_grad_(True)
        loss = (mod(low, full) - target).square().mean()
        loss.backward()
        assert full.grad is not None
        grads.append((loss.detach(), full.grad.clone()))
    torch.testing.assert_close(grads[0][0], grads[1][0], rtol=2e-5, atol=1e-7)
    scale = grads[1][1].abs().max().item()
    err = (grads[0][1] - grads[1][1]).abs().max().item()
    assert err <= 2e-3 * scale + 1e-9, (cls, train, err, scale)
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if p.requires_grad and q.grad is not None:
            s = q.grad.abs().max().item()
            assert (p.grad - q.grad).abs().max().item() <= 2e-3 * s + 1e-7, name


# ---- TensorFlow variable names <-> torch modules (hdrnet_amd/tf_import.py) -----------------------------
@pytest.mark.parametrize("cls", ["HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN"])
def test_tf_variable_mapping_round_trips(cls):
    """export_tf_variables -> load_tf_variables into a fresh model reproduces the forward bit for bit, the
    exported names / layouts are the reference's (hdrnet/layers.py:25-93 scopes, conv [kh, kw, cin, cout],
    fc [cin, cout]) and a missing or mis-shaped variable is an error.  (That the mapping matches a REAL
    TensorFlow dump is what test_tf_fixture_parity checks, when fixtures exist.)"""
    from hdrnet_amd import tf_import
    torch.manual_seed(3)
    a = getattr(models, cls)(dict(batch_norm=True)).eval()
    with torch.no_grad():  # non-trivial statistics / curve parameters
        for m in a.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.05)
    v = tf_import.export_tf_variables(a)
    assert v["inference/coefficients/splat/conv1/weights:0"].shape == (3, 3, 3, 8)
    assert v["inference/coefficients/global/fc1/weights:0"].shape == (4 * 4 * 64, 256)
    assert "inference/coefficients/splat/conv1/biases:0" in v            # first splat layer: no batch norm
    assert "inference/coefficients/splat/conv2/BatchNorm/moving_variance:0" in v
    assert "inference/coefficients/local/conv2/biases:0" not in v        # use_bias=False (models.py:118)
    if cls == "HDRNetCurves":
        assert v["inference/guide/shifts:0"].shape == (1, 1, 3, 16) and v["inference/guide/slopes:0"].shape == (1, 1, 1, 3, 16)
        assert v["inference/guide/channel_mixing/weights:0"].shape == (1, 1, 3, 1)
    elif cls == "HDRNetPointwiseNNGuide":
        assert v["inference/guide/conv1/weights:0"].shape == (1, 1, 3, 16) and v["inference/guide/conv2/biases:0"].shape == (1,)
    else:
        assert v["inference/guide/level_2/conv2/weights:0"].shape == (1, 1, 16, 1)
        assert v["inference/coefficients/prediction/conv1/weights:0"].shape == (1, 1, 64, 8 * 9 * 4)
    torch.manual_seed(99)
    b = getattr(models, cls)(dict(batch_norm=True)).eval()
    tf_import.load_tf_variables(b, v)
    lo = torch.rand(1, 256, 256, 3)
    with torch.no_grad():
        assert torch.equal(a.coefficients(lo), b.coefficients(lo))
        hi = torch.rand(1, 16, 16, 3)
        ga = [g(hi) for g in a.guide] if cls == "HDRNetGaussianPyrNN" else [a.guide(hi)]
        gb = [g(hi) for g in b.guide] if cls == "HDRNetGaussianPyrNN" else [b.guide(hi)]
        assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    bad = dict(v)
    del bad["inference/coefficients/global/fc2/weights:0"]
    with pytest.raises(KeyError):
        tf_import.load_tf_variables(b, bad)
    bad = dict(v)
    bad["inference/coefficients/global/fc2/weights:0"] = bad["inference/coefficients/global/fc2/weights:0"].T.copy()
    with pytest.raises(ValueError):
        tf_import.load_tf_variables(b, bad)
    # a variable under inference/ that the mapping has no place for (a batch norm trained with scale=True) is refused;
    # optimizer slots, the step counter and a fixture's own arrays are not model state
    extra = dict(v)
    extra["inference/coefficients/splat/conv2/BatchNorm/gamma:0"] = np.ones(16, np.float32)
    with pytest.raises(ValueError, match="no place for"):
        tf_import.load_tf_variables(b, extra)
    tf_import.load_tf_variables(b, extra, strict=False)
    ok = dict(v)
    ok["inference/coefficients/splat/conv1/weights/Adam:0"] = np.zeros((3, 3, 3, 8), np.float32)
    ok["global_step:0"] = np.zeros((), np.int64)
    ok["lowres_input"] = np.zeros((1, 256, 256, 3), np.float32)
    tf_import.load_tf_variables(b, ok)


TF_FIXTURES = os.path.join(ROOT, "tests", "golden", "tf")


@pytest.mark.parametrize("cls", ["HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN"])
def test_tf_fixture_parity(cls):
    """Graph-level parity of the model modules against TensorFlow itself:
    consumes tests/golden/tf/<Model>.npz written by tools/export_tf_fixtures.py on a machine that has
    TensorFlow + the reference.  NO SUCH FIXTURE HAS BEEN PRODUCED YET (no TensorFlow in this image, no
    network) -- until one is committed this test skips.  The fixtures that ARE committed come from the reference's
    graph code executed on oracle/tf1_shim: tests/test_tf_shim_fixtures.py."""
    path = os.path.join(TF_FIXTURES, cls + ".npz")
    if not os.path.exists(path):
        pytest.skip("no TensorFlow fixture (run tools/export_tf_fixtures.py where TensorFlow and the reference exist)")
    from hdrnet_amd import tf_import
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    m = getattr(models, cls)(dict(batch_norm=True)).eval()  # the parameters tools/export_tf_fixtures.py builds with
    m.fuse_guide = False
    tf_import.load_tf_variables(m, {k[len("var/"):]: a for k, a in fx.items() if k.startswith("var/")})
    lo, hi = torch.from_numpy(fx["lowres_input"]), torch.from_numpy(fx["fullres_input"])
    tol = dict(rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        np.testing.assert_allclose(m.coefficients(lo).numpy(), fx["bilateral_coefficients"], **tol)
        if cls == "HDRNetGaussianPyrNN":
            lvls = [hi]
            for _ in range(2):
                h, w = lvls[-1].shape[1] // 2, lvls[-1].shape[2] // 2
                lvls.append(m._resize(lvls[-1], h, w))
            for l, lvl in enumerate(lvls):
                np.testing.assert_allclose(lvl.numpy(), fx["multiscale_%d" % l], **tol)   # TF's legacy resize
                np.testing.assert_allclose(m.guide[l](lvl).numpy(), fx["guide_%d" % l], **tol)
        else:
            np.testing.assert_allclose(m.guide(hi).numpy(), fx["guide"], **tol)
    if "output" in fx and torch.cuda.is_available():
        with torch.no_grad():
            got = m.cuda()(lo.cuda(), hi.cuda()).cpu().numpy()
        np.testing.assert_allclose(got, fx["output"], **tol)


def test_split_levels_equals_the_three_slices():
    """models._SplitLevels: the pyramid's per-level grids (hdrnet/models.py:280) in one copy / one stack == slicing +
    reshape under autograd, values and gradients, including a level whose output is not used."""
    torch.manual_seed(0)
    c = torch.randn(2, 4, 5, 3, 9, 4, dtype=torch.float64, requires_grad=True)
    ws = [torch.randn(2, 4, 5, 3, 12, dtype=torch.float64) for _ in range(3)]
    for used in ((0, 1, 2), (0,), (2, 1)):
        outs = models._SplitLevels.apply(c, 3)
        ref = [c[:, :, :, :, l * 3:(l + 1) * 3, :].reshape(2, 4, 5, 3, 12) for l in range(3)]
        assert all(o.is_contiguous() and torch.equal(o, r) for o, r in zip(outs, ref))
        g1, = torch.autograd.grad(sum((outs[l] * ws[l]).sum() for l in used), c)
        g2, = torch.autograd.grad(sum((ref[l] * ws[l]).sum() for l in used), c)
        assert torch.equal(g1, g2)


def test_metrics_match_the_reference_formulas():
    """hdrnet/metrics.py:8-20: l2_loss = mean(square(target - prediction)), psnr = mean over the batch of
    -10 / ln 10 * log(mean(square))."""
    import math
    from hdrnet_amd import metrics
    torch.manual_seed(0)
    t, p = torch.rand(3, 8, 9, 3, dtype=torch.float64), torch.rand(3, 8, 9, 3, dtype=torch.float64, requires_grad=True)
    want = (t - p).square().mean()
    got = metrics.l2_loss(t, p)
    assert torch.allclose(got, want, rtol=1e-12)
    g1, = torch.autograd.grad(got, p)
    g2, = torch.autograd.grad(want, p)
    assert torch.allclose(g1, g2, rtol=1e-12)
    sq = (t - p.detach()).square().reshape(3, -1).mean(1)
    assert torch.allclose(metrics.psnr(t, p.detach()), (-10.0 / math.log(10.0) * sq.log()).mean(), rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,n", [(3, 16), (3, 8), (1, 4), (3, 16)])
def test_guide_fold_batch_kernel_equals_the_torch_fold(cin, n):
    """hdrnet_guide_fold_batch_f32 and its VJP against the differentiable float64 torch math it replaces
    (_PointwiseNNGuide._folded_batch_torch): values, every parameter gradient, the running statistics."""
    import copy
    torch.manual_seed(cin * 100 + n)
    dev = "cuda:0"
    g = models._PointwiseNNGuide(n, nchans=cin).to(dev).train()
    with torch.no_grad():
        g.bn.bias.normal_(0, 0.3)
        g.b2.fill_(0.2)
        g.bn.running_mean.normal_(0, 0.1)
        g.bn.running_var.uniform_(0.5, 1.5)
    ref = copy.deepcopy(g)
    x = torch.rand(5000, cin, device=dev)
    sums, mom = x.sum(0), x.t() @ x
    if cin == 3 and n == 8:  # a degenerate feature: zero weights -> zero variance (the clamp's flat side)
        with torch.no_grad():
            g.w1[:, 2] = 0
            ref.w1[:, 2] = 0
    c1, c2 = g.folded_batch(sums, mom, x.shape[0])
    r1, r2 = ref._folded_batch_torch(sums, mom, x.shape[0])
    assert torch.allclose(c1, r1, rtol=1e-6, atol=1e-7) and torch.allclose(c2, r2)
    assert torch.allclose(g.bn.running_mean, ref.bn.running_mean, rtol=1e-6, atol=1e-8)
    assert torch.allclose(g.bn.running_var, ref.bn.running_var, rtol=1e-6, atol=1e-8)
    assert int(g.bn.num_batches_tracked) == int(ref.bn.num_batches_tracked) == 1
    w1v, w2v = torch.randn_like(c1), torch.randn_like(c2)
    ((c1 * w1v).sum() + (c2 * w2v).sum()).backward()
    ((r1 * w1v).sum() + (r2 * w2v).sum()).backward()
    for (name, p), (_, q) in zip(g.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None, name
            continue
        scale = float(q.grad.abs().max()) + 1e-12
        assert float((p.grad - q.grad).abs().max()) <= 2e-6 * scale, (name, p.grad, q.grad)
    # the eval-mode fold cache notices the running statistics the kernel moved
    g.eval()
    a = g.folded()
    g.train()
    g.folded_batch(sums, mom, x.shape[0])
    g.eval()
    assert g.folded() is not a


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 270, 480, 3), (1, 7, 5, 3), (3, 33, 41, 1)])
def test_l2_loss_kernels_equal_the_reference_formula(shape):
    """metrics.l2_loss on the GPU (csrc/metrics.hip) against mean(square(target - prediction)) in float64, values and
    gradient, including lengths that are not a multiple of 4 and an upstream gradient other than 1."""
    from hdrnet_amd import metrics
    torch.manual_seed(5)
    t = torch.rand(shape, device="cuda:0")
    p = torch.rand(shape, device="cuda:0", requires_grad=True)
    loss = metrics.l2_loss(t, p)
    assert "L2Loss" in type(loss.grad_fn).__name__
    (loss * 3.5).backward()
    p64 = p.detach().double().requires_grad_(True)
    want = (t.double() - p64).square().mean()
    (want * 3.5).backward()
    assert abs(float(loss) - float(want)) <= 1e-6 * float(want)
    assert torch.allclose(p.grad.double(), p64.grad, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
def test_l2_loss_fused_unit_gradient_paths():
    """The forward's unit gradient: grad_output == 1 (the scale kernel returns at once), a second backward through a
    retained graph (recomputed by hdrnet_l2_loss_grad_f32, not scaled twice), no gradient wanted (the plain forward),
    and the C entry points' argument checks."""
    from hdrnet_amd import _lib, metrics
    torch.manual_seed(6)
    shape = (2, 37, 53, 3)
    t = torch.rand(shape, device="cuda:0")
    p = torch.rand(shape, device="cuda:0", requires_grad=True)
    want = (2.0 / p.numel()) * (p.detach().double() - t.double())
    loss = metrics.l2_loss(t, p)
    g1, = torch.autograd.grad(loss, p, retain_graph=True)                      # root: grad_output = 1
    assert torch.allclose(g1.double(), want, rtol=1e-6, atol=1e-14)
    g2, = torch.autograd.grad(loss, p, torch.tensor(-2.5, device="cuda:0"), retain_graph=True)
    assert torch.allclose(g2.double(), -2.5 * want, rtol=1e-6, atol=1e-14)
    assert torch.allclose(g1.double(), want, rtol=1e-6, atol=1e-14)            # the first result was not touched
    loss = metrics.l2_loss(t, p)
    g3, = torch.autograd.grad(loss, p, torch.tensor(0.75, device="cuda:0"))   # first backward with a scale
    assert torch.allclose(g3.double(), 0.75 * want, rtol=1e-6, atol=1e-14)
    with torch.no_grad():
        plain = metrics.l2_loss(t, p)
    assert plain.grad_fn is None and abs(float(plain) - float(loss)) <= 1e-7 * float(loss)
    lib = _lib.load()
    assert lib.hdrnet_l2_loss_with_grad_f32(None, None, 16, None, None, None, 0, None) == 1
    assert lib.hdrnet_l2_loss_grad_scale_f32(None, None, 16, None) == 1
    assert lib.hdrnet_l2_loss_grad_scale_f32(g3.data_ptr(), g3.data_ptr(), 0, None) == 1


def _flat_adam_vs_torch(dev, steps=4):
    import copy
    from hdrnet_amd import optim
    torch.manual_seed(8)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(5 * 36, 7)).to(dev)
    net = net.to(memory_format=torch.channels_last)
    ref = copy.deepcopy(net)
    opt = optim.FlatAdam(net.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    ropt = torch.optim.Adam(ref.parameters(), lr=3e-3, betas=(0.9, 0.99), eps=1e-8)
    for p in net.parameters():  # storage re-bound to the flat buffer, values and strides kept
        assert p.data_ptr() >= opt.flat.data_ptr() and p.data_ptr() < opt.flat.data_ptr() + 4 * opt.flat.numel()
        assert p.data_ptr() % 16 == 0
    for (_, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q) and p.stride() == q.stride()
    for i in range(steps):
        x = torch.rand(6, 3, 8, 8, device=dev)
        opt.bucket.release()
        net(x).square().mean().backward()
        opt.bucket.gather()
        opt.step()
        ropt.zero_grad(set_to_none=True)
        ref(x).square().mean().backward()
        ropt.step()
    for (name, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=2e-5, atol=2e-7), (name, float((p - q).abs().max()))
    assert float(opt.steps) == steps


def test_flat_adam_equals_torch_adam_cpu():
    _flat_adam_vs_torch("cpu")


def _flat_adam_epsilon_hat_vs_tensorflow_formula(dev, steps=5):
    """epsilon_hat=True against tf.train.AdamOptimizer's update (tensorflow/python/training/adam.py: lr_t = lr *
    sqrt(1 - b2^t) / (1 - b1^t); m, v as usual; var -= lr_t * m / (sqrt(v) + eps)) restated in float64 -- the optimizer
    hdrnet/bin/train.py:113 builds.  eps is large here so that the two placements of epsilon differ visibly."""
    from hdrnet_amd import optim
    torch.manual_seed(11)
    lr, b1, b2, eps = 2e-3, 0.9, 0.99, 1e-3
    w = torch.nn.Parameter(torch.randn(37, 5, device=dev))
    w0 = w.detach().clone()
    opt = optim.FlatAdam([w], lr=lr, betas=(b1, b2), eps=eps, epsilon_hat=True)
    other = torch.nn.Parameter(w0.clone())
    oopt = optim.FlatAdam([other], lr=lr, betas=(b1, b2), eps=eps)
    var, m, v = w0.double().cpu(), torch.zeros(37, 5, dtype=torch.float64), torch.zeros(37, 5, dtype=torch.float64)
    for t in range(1, steps + 1):
        g = torch.randn(37, 5, device=dev) * 0.01
        for o in (opt, oopt):
            o.bucket.flat[:g.numel()].copy_(g.reshape(-1))
            o.step()
        g64 = g.double().cpu()
        lr_t = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
        m = b1 * m + (1 - b1) * g64
        v = b2 * v + (1 - b2) * g64 * g64
        var = var - lr_t * m / (v.sqrt() + eps)
    assert torch.allclose(w.detach().double().cpu(), var, rtol=1e-5, atol=1e-7)
    assert float((other.detach().double().cpu() - var).abs().max()) > 1e-4        # torch's placement is another update


def test_flat_adam_epsilon_hat_is_tensorflows_adam_cpu():
    _flat_adam_epsilon_hat_vs_tensorflow_formula("cpu")


@pytest.mark.gpu
def test_flat_adam_epsilon_hat_kernel_is_tensorflows_adam():
    _flat_adam_epsilon_hat_vs_tensorflow_formula("cuda:0")


@pytest.mark.gpu
def test_flat_adam_kernel_equals_torch_adam():
    _flat_adam_vs_torch("cuda:0", steps=6)


def test_train_header_symbols_are_exported_and_bound():
    """include/hdrnet_amd_train.h <-> _lib.TRAIN_SIGNATURES <-> the library's exports."""
    import ctypes
    import re
    from hdrnet_amd import _lib, build
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hdrnet_amd_train.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(hdrnet_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(_lib.TRAIN_SIGNATURES)
    lib = ctypes.CDLL(build.build())
    for n in names:
        assert hasattr(lib, n)
        m = re.search(r"\b" + n + r"\s*\(([^)]*)\)", src)
        assert len(m.group(1).split(",")) == len(_lib.TRAIN_SIGNATURES[n][1])
    for n in ("hdrnet_adam_step_f32", "hdrnet_adam_step_tf_f32"):
        getattr(lib, n).argtypes = _lib.TRAIN_SIGNATURES[n][1]
        assert getattr(lib, n)(None, None, None, None, 16, None, 1e-3, 0.9, 0.999, 1e-8, None) == 1
    lib.hdrnet_l2_loss_with_grad_f32.argtypes = _lib.TRAIN_SIGNATURES["hdrnet_l2_loss_with_grad_f32"][1]
    assert lib.hdrnet_l2_loss_with_grad_f32(None, None, 16, None, None, None, 0, None) == 1
    lib.hdrnet_l2_loss_grad_scale_f32.argtypes = _lib.TRAIN_SIGNATURES["hdrnet_l2_loss_grad_scale_f32"][1]
    assert lib.hdrnet_l2_loss_grad_scale_f32(None, None, 16, None) == 1
    lib.hdrnet_resize_add_f32.argtypes = _lib.TRAIN_SIGNATURES["hdrnet_resize_add_f32"][1]
    assert lib.hdrnet_resize_add_f32(None, None, None, 1, 4, 4, 8, 8, 3, None) == 1      # null tensors
    lib.hdrnet_resize_bilinear_grad_f32.argtypes = _lib.TRAIN_SIGNATURES["hdrnet_resize_bilinear_grad_f32"][1]
    assert lib.hdrnet_resize_bilinear_grad_f32(None, None, 1, 4, 4, 8, 8, 3, None) == 1


# ---- parameter-state generation: caches of derived arrays vs the writers that bypass version counters ----
def test_derived_array_caches_follow_flat_adam_cpu():
    """optim.FlatAdam updates the flat parameter buffer behind the parameters' version counters (a raw pointer on the
    GPU, the flat tensor on the CPU): the caches of derived arrays (guide network folded with batch norm, the curves
    guide's exported layout) are keyed on those counters PLUS hdrnet_amd._state.generation_of(parameter), which
    FlatAdam.step() bumps for ITS parameters -- eval -> train -> eval must see the new weights."""
    from hdrnet_amd import optim
    torch.manual_seed(3)
    for guide, get in ((models._PointwiseNNGuide(16), lambda g: g.folded()),
                       (models._CurvesGuide(), lambda g: g.exported())):
        guide.eval()
        before = [t.clone() for t in get(guide)]
        assert all(a is b for a, b in zip(get(guide), get(guide)))  # cached
        opt = optim.FlatAdam([p for p in guide.parameters() if p.requires_grad], lr=0.05)
        versions = [p._version for p in opt.bucket.params]
        opt.bucket.flat.fill_(1.0)
        opt.step()
        assert [p._version for p in opt.bucket.params] == versions  # the update really is invisible to the counters
        after = get(guide)
        assert any(not torch.equal(a, b) for a, b in zip(before, after)), "the cache served the old parameters"
        fresh = guide.folded(detach=False) if isinstance(guide, models._PointwiseNNGuide) else guide.exported_differentiable()
        for a, b in zip(after, fresh):
            torch.testing.assert_close(a, b.detach(), rtol=0, atol=0)


def test_a_train_step_on_one_model_leaves_another_model_s_caches_alone_cpu():
    """the generation is scoped to the tensors a writer actually wrote.  FlatAdam steps on model A
    must not invalidate the derived-array caches (nor the GraphedInference staleness key) of an untouched model B -- a
    frozen teacher or an EMA copy evaluated during training -- while A's own caches still follow A's parameters."""
    from hdrnet_amd import _state, optim
    from hdrnet_amd.runtime import GraphedInference
    torch.manual_seed(4)
    a, b = models._PointwiseNNGuide(16).eval(), models._PointwiseNNGuide(16).eval()
    fa0, fb0 = a.folded(), b.folded()
    key_b = GraphedInference._parameter_state(type("G", (), {"module": b})())
    opt = optim.FlatAdam([p for p in a.parameters() if p.requires_grad], lr=0.05)
    opt.bucket.flat.fill_(1.0)
    for _ in range(3):
        opt.step()
    assert all(x is y for x, y in zip(b.folded(), fb0)), "B's cache was invalidated by A's optimizer"
    assert GraphedInference._parameter_state(type("G", (), {"module": b})()) == key_b
    assert any(not torch.equal(x, y) for x, y in zip(a.folded(), fa0)), "A's cache served the old parameters"
    assert all(_state.generation_of(p) == 3 for p in opt.bucket.params)
    assert all(_state.generation_of(p) == 0 for p in b.parameters())
    # two writers on one tensor (an optimizer and a graphed step): their generations add up
    w = _state.Writer(opt.bucket.params)
    w.bump()
    assert all(_state.generation_of(p) == 4 for p in opt.bucket.params)
    w.attach(opt.bucket.params)  # idempotent
    assert all(len(getattr(p, "_hdrnet_writers")) == 2 for p in opt.bucket.params)


def test_flat_adam_state_dict_round_trip_and_detached_parameter_cpu():
    from hdrnet_amd import optim
    torch.manual_seed(5)
    lin = torch.nn.Linear(7, 5)
    opt = optim.FlatAdam(lin.parameters(), lr=0.01)
    for _ in range(3):
        opt.bucket.flat.copy_(torch.randn_like(opt.bucket.flat))
        opt.step()
    state = opt.state_dict()
    params = opt.flat.clone()
    g = torch.randn_like(opt.bucket.flat)
    opt.bucket.flat.copy_(g)
    opt.step()
    want = opt.flat.clone()
    lin2 = torch.nn.Linear(7, 5)
    opt2 = optim.FlatAdam(lin2.parameters(), lr=0.5)
    opt2.flat.copy_(params)
    opt2.load_state_dict(state)
    opt2.bucket.flat.copy_(g)
    opt2.step()
    torch.testing.assert_close(opt2.flat, want, rtol=0, atol=0)
    with pytest.raises(ValueError):
        optim.FlatAdam(torch.nn.Linear(3, 3).parameters()).load_state_dict(state)
    lin2.weight.data = lin2.weight.data.clone()  # what module.to(...) does: the parameter leaves the flat buffer
    with pytest.raises(RuntimeError, match="flat buffer"):
        opt2.step()


@pytest.mark.gpu
def test_eval_train_eval_sees_the_trained_weights_and_stale_graph_raises():
    """ADVICE r04 (high) + VERDICT r04 item 6: an eval forward fills the derived-array caches (native coefficient
    network weights, folded guide); FlatAdam steps and GraphedTrainStep replays then change parameters and batch-norm
    statistics without touching a version counter.  The next eval forward must use the NEW state (compared with the
    torch-op composition, native = False), and a GraphedInference captured before the training must refuse to replay."""
    from hdrnet_amd import metrics, optim
    from hdrnet_amd.runtime import GraphedInference, GraphedTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=True)).to(dev)
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 96, 128, 3, device=dev)
    target = torch.rand(2, 96, 128, 3, device=dev)
    m.eval()
    with torch.no_grad():
        out0 = m(low, full).clone()
    gi = GraphedInference(m, [low, full])
    torch.testing.assert_close(gi(low, full), out0, rtol=1e-6, atol=1e-6)
    m.train()
    opt = optim.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-2, epsilon_hat=True)
    step = GraphedTrainStep(m, lambda out, tgt: metrics.l2_loss(tgt, out), opt, [low, full], [target])
    for _ in range(4):
        step([low, full], [target])
    m.eval()
    with torch.no_grad():
        out1 = m(low, full).clone()
        m.coefficients.native = False
        m.fuse_guide = False
        try:
            ref = m(low, full).clone()
        finally:
            del m.coefficients.native, m.fuse_guide  # back to the class defaults
    assert (out1 - out0).abs().max() > 1e-3, "training did not move the output: the test would prove nothing"
    torch.testing.assert_close(out1, ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="recapture"):
        gi(low, full)
    gi.recapture()
    torch.testing.assert_close(gi(low, full), out1, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_training_model_a_leaves_a_captured_teacher_replayable():
    """GraphedTrainStep replays + FlatAdam steps on a student must neither make a frozen teacher's
    GraphedInference raise "parameters changed" nor invalidate the teacher's derived-array caches; the student's own
    captured inference graph still refuses to replay."""
    from hdrnet_amd import metrics, optim
    from hdrnet_amd.runtime import GraphedInference, GraphedTrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    student = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev)
    teacher = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).eval()
    low = torch.rand(2, 256, 256, 3, device=dev)
    full = torch.rand(2, 96, 128, 3, device=dev)
    with torch.no_grad():
        want = teacher(low, full).clone()
    g_teacher = GraphedInference(teacher, [low, full])
    g_student = GraphedInference(student, [low, full])
    student.train()
    opt = optim.FlatAdam([p for p in student.parameters() if p.requires_grad], lr=1e-2, epsilon_hat=True)
    step = GraphedTrainStep(student, lambda out, tgt: metrics.l2_loss(tgt, out), opt, [low, full], [want])
    for _ in range(3):
        step([low, full], [want])
        torch.testing.assert_close(g_teacher(low, full), want, rtol=1e-6, atol=1e-6)  # no "call recapture()"
    student.eval()
    with pytest.raises(RuntimeError, match="recapture"):
        g_student(low, full)


def test_exact_inference_switch_cpu():
    """one switch for the three inference-only numerics choices (fast sigmoid, prescaled guide parameters
    with their |input| <= 65536 contract, prepared curve tables)."""
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=False))
    assert m.fast_sigmoid and m.prescale_guide and m.prepare_curves
    assert m.exact_inference() is m
    assert not (m.fast_sigmoid or m.prescale_guide or m.prepare_curves)
    m.exact_inference(False)
    assert m.fast_sigmoid and m.prescale_guide and m.prepare_curves
    assert models.HDRNetCurves.fast_sigmoid  # the class defaults were never touched


@pytest.mark.gpu
def test_exact_inference_handles_inputs_beyond_the_prescale_range():
    """Un-normalised HDR floats (|x| >> 65536): the default prescaled guide saturates, exact_inference() follows the
    composed torch graph."""
    dev = torch.device("cuda:0")
    torch.manual_seed(31)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).eval()
    low = torch.rand(1, 256, 256, 3, device=dev)
    full = torch.rand(1, 64, 256, 3, device=dev) * 3.0e6
    with torch.no_grad():
        got = m.exact_inference()(low, full).clone()
        m.fuse_guide = False
        try:
            want = m(low, full).clone()
        finally:
            del m.fuse_guide
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
