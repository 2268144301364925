"""Image metrics of the reference's training loop (``hdrnet/metrics.py``), on torch tensors.

``l2_loss`` is what ``hdrnet/bin/train.py:95`` minimises.  The literal ``(target - prediction).square().mean()`` is
five bandwidth-bound passes over 100 MB each at 4 x 1080p (240 us of a 1.55-ms training step); ``F.mse_loss`` still is
five launches (the squares written out, a zeros_like of the gradient: 143 us); csrc/metrics.hip does it in two
passes (profiles/r04/train_step.md) -- and when the prediction wants a gradient the forward pass writes the unit gradient
while it has both operands, so the backward has nothing left to read when grad_output is 1.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

__all__ = ["l2_loss", "psnr"]


class _L2Loss(torch.autograd.Function):
    """csrc/metrics.hip.  With a gradient wanted: ``hdrnet_l2_loss_with_grad_f32`` (the loss and the unit gradient
    (2 / n) (prediction - target) in one pass) and ``hdrnet_l2_loss_grad_scale_f32`` in the backward (nothing but a scalar
    read when grad_output is 1); a second backward through a retained graph recomputes with ``hdrnet_l2_loss_grad_f32``.
    Without: ``hdrnet_l2_loss_f32``."""

    @staticmethod
    def forward(ctx, prediction, target):
        from . import _lib
        p, t = prediction.detach().contiguous(), target.detach().contiguous()
        dev, n = p.device, p.numel()
        loss = torch.empty((), dtype=torch.float32, device=dev)
        lib = _lib.load()
        ctx.unit = None
        with torch.cuda.device(dev):
            wbytes = lib.hdrnet_l2_loss_workspace_bytes(n)
            ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if ctx.needs_input_grad[0]:
                ctx.unit = torch.empty_like(p)
                rc = lib.hdrnet_l2_loss_with_grad_f32(p.data_ptr(), t.data_ptr(), n, loss.data_ptr(), ctx.unit.data_ptr(),
                                                      ws.data_ptr(), wbytes, stream)
                if rc != 0:
                    raise RuntimeError(f"hdrnet_l2_loss_with_grad_f32 failed (rc={rc})")
            else:
                _lib.check(lib.hdrnet_l2_loss_f32(p.data_ptr(), t.data_ptr(), n, loss.data_ptr(), ws.data_ptr(), wbytes,
                                                  stream), "L2Loss")
        ctx.save_for_backward(p, t)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        from . import _lib
        p, t = ctx.saved_tensors
        dev = p.device
        g = grad_output.detach().to(torch.float32).reshape(1).contiguous()
        lib = _lib.load()
        if ctx.unit is not None:  # first backward: the forward's unit gradient, scaled in place
            dpred, ctx.unit = ctx.unit, None
            with torch.cuda.device(dev):
                rc = lib.hdrnet_l2_loss_grad_scale_f32(dpred.data_ptr(), g.data_ptr(), p.numel(),
                                                       torch.cuda.current_stream(dev).cuda_stream)
            if rc != 0:
                raise RuntimeError(f"hdrnet_l2_loss_grad_scale_f32 failed (rc={rc})")
            return dpred, None
        dpred = torch.empty_like(p)
        with torch.cuda.device(dev):
            rc = lib.hdrnet_l2_loss_grad_f32(p.data_ptr(), t.data_ptr(), g.data_ptr(), p.numel(), dpred.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "L2LossGrad")
        return dpred, None


def l2_loss(target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
    """``tf.reduce_mean(tf.square(target - prediction))`` (hdrnet/metrics.py:8-11).  On the GPU in fp32 with no
    gradient wanted for the target: the two-pass HIP kernels; otherwise ``F.mse_loss``."""
    if (prediction.is_cuda and target.is_cuda and prediction.dtype == torch.float32 and target.dtype == torch.float32
            and prediction.shape == target.shape and not target.requires_grad and prediction.numel() > 0
            # the kernels read float4s: a contiguous but offset view (x.flatten()[1:]) takes the stock op instead of
            # failing in the C-ABI's alignment check
            and prediction.is_contiguous() and target.is_contiguous()
            and prediction.data_ptr() % 16 == 0 and target.data_ptr() % 16 == 0):
        return _L2Loss.apply(prediction, target)
    return F.mse_loss(prediction, target)


def psnr(target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
    """Mean PSNR over the batch, ``-10 / ln 10 * log(mean_per_image(square(target - prediction)))``
    (hdrnet/metrics.py:14-20)."""
    squares = (target - prediction).square().reshape(target.shape[0], -1)
    return ((-10.0 / math.log(10.0)) * torch.log(squares.mean(dim=1))).mean()
