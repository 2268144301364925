"""The two slice wrappers of the reference's ``hdrnet/layers.py`` (:99-148), on torch.

They exist so that the one product call site of the hot path keeps its shape:
``layers.bilateral_slice_apply(coeffs, guide, im, has_offset=True, name='slice')``
(``hdrnet/models.py:193-196``) with ``coeffs`` the 6-D ``[B, GH, GW, GD, n_out, n_in]``
tensor the coefficient network emits (``models.py:134-139``).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hdrnet_ops

__all__ = ["bilateral_slice", "bilateral_slice_apply"]


def bilateral_slice(grid: torch.Tensor, guide: torch.Tensor, name: Optional[str] = None) -> torch.Tensor:
    """Slices into a bilateral grid using the guide map (hdrnet/layers.py:99-121).

    grid: ``[B, GH, GW, GD, C]`` or 6-D ``[B, GH, GW, GD, n_out, n_in]``; guide ``[B, H, W]``.
    Returns ``[B, H, W, C]``, or ``[B, H, W, n_out, n_in]`` for a 6-D grid.
    """
    del name
    if grid.dim() == 6:
        n_out, n_in = grid.shape[4], grid.shape[5]
        # tf.concat(tf.unstack(grid, axis=5), 4): channel = j * n_out + i  (layers.py:113-115)
        flat = torch.cat(torch.unbind(grid, dim=5), dim=4)
        sliced = hdrnet_ops.bilateral_slice(flat, guide)
        # tf.stack(tf.split(sliced, n_in, axis=3), axis=4)                   (layers.py:119-120)
        return torch.stack(torch.split(sliced, n_out, dim=3), dim=4)
    return hdrnet_ops.bilateral_slice(grid, guide)


def bilateral_slice_apply(grid: torch.Tensor, guide: torch.Tensor, input_image: torch.Tensor,
                          has_offset: bool = True, name: Optional[str] = None) -> torch.Tensor:
    """Slices a bilateral grid and applies the per-pixel affine (hdrnet/layers.py:125-148).

    grid: ``[B, GH, GW, GD, n_out * n_in]`` or 6-D ``[B, GH, GW, GD, n_out, n_in]`` (reshaped,
    zero-copy, to 5-D with channel = i * n_in + j, layers.py:141-144); guide ``[B, H, W]``;
    input_image ``[B, H, W, n_in - has_offset]``.  Returns ``[B, H, W, n_out]``.
    """
    del name
    if grid.dim() == 6:
        gs = grid.shape
        grid = grid.reshape(gs[0], gs[1], gs[2], gs[3], gs[4] * gs[5])
    return hdrnet_ops.bilateral_slice_apply(grid, guide, input_image, has_offset=has_offset)
