"""Interleaved A/B of the training step's loss: the two-pass pair of include/hdrnet_amd.h (forward reads prediction and
target, backward reads them again) against the forward that also writes the unit gradient (include/hdrnet_amd_train.h)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hdrnet_amd import _lib, models, metrics, optim
from hdrnet_amd.runtime import GraphedTrainStep


class TwoPass(torch.autograd.Function):   # round 4's first form, kept here as the A side
    @staticmethod
    def forward(ctx, prediction, target):
        p, t = prediction.detach().contiguous(), target.detach().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        lib = _lib.load()
        wbytes = lib.hdrnet_l2_loss_workspace_bytes(p.numel())
        ws = torch.empty((wbytes,), dtype=torch.uint8, device=p.device)
        _lib.check(lib.hdrnet_l2_loss_f32(p.data_ptr(), t.data_ptr(), p.numel(), loss.data_ptr(), ws.data_ptr(), wbytes,
                                          torch.cuda.current_stream().cuda_stream), "L2Loss")
        ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        p, t = ctx.saved_tensors
        g = grad_output.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty_like(p)
        _lib.check(_lib.load().hdrnet_l2_loss_grad_f32(p.data_ptr(), t.data_ptr(), g.data_ptr(), p.numel(), d.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream), "L2LossGrad")
        return d, None


dev = torch.device("cuda:0")
B, H, W = 4, 1080, 1920
low = torch.rand(B, 256, 256, 3, device=dev); full = torch.rand(B, H, W, 3, device=dev); tgt = torch.rand(B, H, W, 3, device=dev)


def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


steps = {}
for kind in ("two-pass", "fused"):
    torch.manual_seed(0)
    m = models.HDRNetPointwiseNNGuide(dict(batch_norm=False)).to(dev).train()
    opt = optim.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4, epsilon_hat=True)
    fn = (lambda o, t: TwoPass.apply(o, t)) if kind == "two-pass" else (lambda o, t: metrics.l2_loss(t, o))
    steps[kind] = GraphedTrainStep(m, fn, opt, [low, full], [tgt])
for r in range(4):
    for kind, step in steps.items():
        si, st = step.static_inputs, step.static_targets
        print("round %d %-8s copy feed %.4f ms | static feed %.4f | loss %.6f" % (
            r, kind, timeit(lambda: step([low, full], [tgt])), timeit(lambda: step(si, st)), float(step.static_loss)), flush=True)
