"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's JAX twin.

``jax`` / ``jaxlib`` are not installed in this image (and cannot be: no network),
so ``/root/reference/jax/bilateral_slice.py`` cannot be imported.  This module
restates its published algorithm in float32 numpy, function by function:

* ``bilateral_slice``            <- jax/bilateral_slice.py:299-380
* ``bilateral_slice_guide_vjp``  <- jax/bilateral_slice.py:26-108
* ``bilateral_slice_grid_vjp``   <- jax/bilateral_slice.py:257-295
  (spatial weights :138-160, symmetric pad :163-181, range weights :184-254)
* numerics                       <- jax/numerics.py:20-97

Like the JAX functions these are UNBATCHED: grid (gh, gw, gd, gc), guide (h, w).
``batched`` mirrors the ``jax.vmap(..., in_axes=0)`` wrapper of
hdrnet/hdrnet_ops_jax_tf2_test.py:23-24.

Parity status: pinned indirectly -- tests/test_oracle_pinning.py checks it
against oracle/_ref (the reference C++ op) at the reference's own JAX==op bar
(``assertAllClose`` defaults rtol=atol=1e-6, hdrnet_ops_jax_tf2_test.py:48).
"""
from __future__ import annotations

import numpy as np

F = np.float32
_EPS = F(1.0e-8)


# ---- jax/numerics.py ------------------------------------------------------------
def lerp_weight(x, xs):
    """jax/numerics.py:20-40."""
    dx = (x - xs).astype(F)
    return np.maximum(F(1.0) - np.abs(dx), F(0.0)).astype(F)


def smoothed_abs(x, eps=_EPS):
    """jax/numerics.py:43-45."""
    return np.sqrt(np.multiply(x, x, dtype=F) + eps).astype(F)


def smoothed_abs_grad(x, eps=_EPS):
    """jax/numerics.py:48-60."""
    return (x / np.sqrt(np.multiply(x, x, dtype=F) + eps)).astype(F)


def smoothed_lerp_weight(x, xs, eps=_EPS):
    """jax/numerics.py:63-89."""
    dx = (x - xs).astype(F)
    return np.maximum(F(1.0) - smoothed_abs(dx, eps), F(0.0)).astype(F)


def smoothed_lerp_weight_grad(x, xs, eps=_EPS):
    """jax/numerics.py:92-97."""
    dx = (x - xs).astype(F)
    abs_dx = smoothed_abs(dx, eps)
    return np.where(abs_dx > F(1.0), F(0.0), smoothed_abs_grad(dx, eps)).astype(F)


# ---- shared coordinate / weight set-up (bilateral_slice.py:316-355 == :41-81) ----
def _corners(grid_shape, guide, rows=None):
    """rows = (r0, r1, H): `guide` holds only image rows r0..r1-1 of an H-row image (the
    CPU-baseline leg of bench.py evaluates a bounded row window of a full-size frame; the
    reference evaluates whole images, rows=None)."""
    gh, gw, gd = grid_shape[:3]
    h, w = guide.shape
    r0 = 0
    if rows is not None:
        r0, r1, h_full = rows
        assert r1 - r0 == h
        h = h_full
    ii, jj = np.meshgrid(np.arange(r0, r0 + guide.shape[0]), np.arange(w), indexing="ij")
    scale_i = F(gh / h)  # python-float scale applied in f32, as jnp weak typing does
    scale_j = F(gw / w)
    gif = (ii.astype(F) + F(0.5)) * scale_i
    gjf = (jj.astype(F) + F(0.5)) * scale_j
    gkf = guide.astype(F) * F(gd)
    gi0 = np.floor(gif - F(0.5)).astype(np.int32)
    gj0 = np.floor(gjf - F(0.5)).astype(np.int32)
    gk0 = np.floor(gkf - F(0.5)).astype(np.int32)
    return gif, gjf, gkf, gi0, gj0, gk0


def _gather8(grid, gi0, gj0, gk0):
    gh, gw, gd = grid.shape[:3]
    gi = (gi0.clip(0, gh - 1), (gi0 + 1).clip(0, gh - 1))
    gj = (gj0.clip(0, gw - 1), (gj0 + 1).clip(0, gw - 1))
    gk = (gk0.clip(0, gd - 1), (gk0 + 1).clip(0, gd - 1))
    return {(a, b, c): grid[gi[a], gj[b], gk[c], :] for a in (0, 1) for b in (0, 1) for c in (0, 1)}


def _weighted_sum(vals, wi, wj, wk):
    # Same summation order as bilateral_slice.py:373-380 (000,001,010,...,111).
    acc = None
    for a in (0, 1):
        for b in (0, 1):
            for c in (0, 1):
                w = (wi[a] * wj[b] * wk[c]).astype(F)[..., None]
                term = (w * vals[(a, b, c)]).astype(F)
                acc = term if acc is None else (acc + term).astype(F)
    return acc


def bilateral_slice(grid, guide, rows=None):
    """grid (gh,gw,gd,gc), guide (h,w) -> (h,w,gc).  jax/bilateral_slice.py:299-380.
    rows: see _corners (row window of a taller image; not part of the reference API)."""
    grid = np.asarray(grid, F)
    guide = np.asarray(guide, F)
    gif, gjf, gkf, gi0, gj0, gk0 = _corners(grid.shape, guide, rows)
    wi = (lerp_weight(gi0.astype(F) + F(0.5), gif), lerp_weight(gi0.astype(F) + F(1.5), gif))
    wj = (lerp_weight(gj0.astype(F) + F(0.5), gjf), lerp_weight(gj0.astype(F) + F(1.5), gjf))
    wk = (smoothed_lerp_weight(gk0.astype(F) + F(0.5), gkf),
          smoothed_lerp_weight(gk0.astype(F) + F(1.5), gkf))
    return _weighted_sum(_gather8(grid, gi0, gj0, gk0), wi, wj, wk)


def bilateral_slice_guide_vjp(grid, guide, codomain_tangent):
    """-> (h,w).  jax/bilateral_slice.py:26-108."""
    grid = np.asarray(grid, F)
    guide = np.asarray(guide, F)
    ct = np.asarray(codomain_tangent, F)
    gd = grid.shape[2]
    gif, gjf, gkf, gi0, gj0, gk0 = _corners(grid.shape, guide)
    wi = (lerp_weight(gi0.astype(F) + F(0.5), gif), lerp_weight(gi0.astype(F) + F(1.5), gif))
    wj = (lerp_weight(gj0.astype(F) + F(0.5), gjf), lerp_weight(gj0.astype(F) + F(1.5), gjf))
    dwk = (F(gd) * smoothed_lerp_weight_grad(gk0.astype(F) + F(0.5), gkf),
           F(gd) * smoothed_lerp_weight_grad(gk0.astype(F) + F(1.5), gkf))
    grid_val = _weighted_sum(_gather8(grid, gi0, gj0, gk0), wi, wj, dwk)
    return np.sum(grid_val * ct, axis=-1, dtype=F)


def _compute_scale_pad(image_extent, grid_extent):
    """jax/bilateral_slice.py:111-135."""
    scale = image_extent / grid_extent
    return scale, int(np.ceil(0.5 * scale))


def _compute_spatial_weights(image_extent, grid_extent):
    """(image_extent_padded, grid_extent).  jax/bilateral_slice.py:138-160."""
    scale, half_pad = _compute_scale_pad(image_extent, grid_extent)
    indices = np.arange(image_extent + 2 * half_pad) - half_pad
    gfl = ((indices.astype(F) + F(0.5)) / F(scale)).astype(F)
    gif, gi = np.meshgrid(gfl, np.arange(grid_extent), indexing="ij")
    return lerp_weight(gi.astype(F) + F(0.5), gif)


def _symmetric_pad_ij(image, grid_shape):
    """jax/bilateral_slice.py:163-181."""
    _, pi = _compute_scale_pad(image.shape[0], grid_shape[0])
    _, pj = _compute_scale_pad(image.shape[1], grid_shape[1])
    pads = [(pi, pi), (pj, pj)] + [(0, 0)] * (image.ndim - 2)
    return np.pad(image, pads, mode="symmetric")


def _compute_range_weights(guide, grid_shape):
    """(h', w', gd).  jax/bilateral_slice.py:184-254."""
    gp = _symmetric_pad_ij(np.asarray(guide, F), grid_shape)
    gd = grid_shape[2]
    gk = gp * F(gd)
    kf = np.floor(gk - F(0.5))
    kc = np.ceil(gk - F(0.5))
    wf = smoothed_lerp_weight(kf.astype(F) + F(0.5), gk)
    wc = smoothed_lerp_weight(kc.astype(F) + F(0.5), gk)
    kf = kf.astype(np.int32)
    kc = kc.astype(np.int32)
    lo = (kc == 0) & (gk < F(0.5))
    hi = (kf == gd - 1) & (gk > F(gd - 0.5))
    wf = np.where(lo, F(0), wf)   # :232
    wc = np.where(hi, F(0), wc)   # :233-234
    wc = np.where(lo, F(1), wc)   # :235
    wf = np.where(hi, F(1), wf)   # :236-237
    kfc = kf.clip(0, gd - 1)
    kcc = kc.clip(0, gd - 1)
    ii, jj = np.meshgrid(np.arange(gp.shape[0]), np.arange(gp.shape[1]), indexing="ij")
    rw = np.zeros(gp.shape + (gd,), F)
    np.add.at(rw, (ii, jj, kfc), wf.astype(F))
    np.add.at(rw, (ii, jj, kcc), wc.astype(F))
    return rw


def bilateral_slice_grid_vjp(guide, codomain_tangent, grid_shape):
    """-> (gh,gw,gd,gc).  jax/bilateral_slice.py:257-295 (einsum 'ia,jb,ijc,ijd->abcd')."""
    guide = np.asarray(guide, F)
    w_i = _compute_spatial_weights(guide.shape[0], grid_shape[0])
    w_j = _compute_spatial_weights(guide.shape[1], grid_shape[1])
    w_k = _compute_range_weights(guide, grid_shape)
    ct = _symmetric_pad_ij(np.asarray(codomain_tangent, F), grid_shape)
    # Contract in steps (i first, then j) instead of one 6-index einsum.
    t = (w_k[:, :, :, None] * ct[:, :, None, :]).astype(F)      # [i, j, c, d]
    t = np.tensordot(w_i.T, t, axes=([1], [0])).astype(F)        # [a, j, c, d]
    t = np.tensordot(w_j.T, t, axes=([1], [1])).astype(F)        # [b, a, c, d]
    return np.ascontiguousarray(t.transpose(1, 0, 2, 3))


def batched(fn, *arrays, **kw):
    """The jax.vmap(in_axes=0) wrapper of hdrnet_ops_jax_tf2_test.py:23-24."""
    return np.stack([fn(*[a[b] for a in arrays], **kw) for b in range(arrays[0].shape[0])])
