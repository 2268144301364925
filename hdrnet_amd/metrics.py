"""Image metrics of the reference's training loop (``hdrnet/metrics.py``), on torch tensors.

``l2_loss`` is what ``hdrnet/bin/train.py:95`` minimises.  It is written with ``F.mse_loss`` -- ONE fused pass forward
and one backward over the full-resolution batch; the literal ``(target - prediction).square().mean()`` is five
bandwidth-bound passes over 100 MB each at 4 x 1080p (240 us of a 1.55-ms training step, profiles/r04/train_step.md).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

__all__ = ["l2_loss", "psnr"]


def l2_loss(target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
    """``tf.reduce_mean(tf.square(target - prediction))`` (hdrnet/metrics.py:8-11)."""
    return F.mse_loss(prediction, target)


def psnr(target: torch.Tensor, prediction: torch.Tensor) -> torch.Tensor:
    """Mean PSNR over the batch, ``-10 / ln 10 * log(mean_per_image(square(target - prediction)))``
    (hdrnet/metrics.py:14-20)."""
    squares = (target - prediction).square().reshape(target.shape[0], -1)
    return ((-10.0 / math.log(10.0)) * torch.log(squares.mean(dim=1))).mean()
