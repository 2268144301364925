"""TEST INFRASTRUCTURE ONLY -- the CPU oracle for the bilateral-grid hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``hdrnet_amd/`` does (enforced by
``tests/test_no_oracle_in_product.py``).

Two checkers live here, both operating on C-contiguous float32 numpy arrays in
the reference's NHWC layouts (``hdrnet/ops/bilateral_slice_apply_op.cc:201-227``):

* ``port``  -- ``liboracle.so``: the C99 restatement ``bilateral_oracle.c``.
* ``ref``   -- ``_ref/libhdrnet_ref.so``: the reference's own
  ``bilateral_slice_apply.cc`` / ``bilateral_slice.cc`` compiled unchanged
  (``oracle/Makefile``).  Exists wherever ``/root/reference`` existed at build
  time; the prebuilt library travels to the GPU box.

``jax_np`` restates ``jax/bilateral_slice.py`` in numpy (jax is not installed).
``tf1_shim/tensorflow`` is an eager numpy stand-in for the TensorFlow 1.x calls of the reference's graph code
(``hdrnet/models.py``, ``layers.py``), used by ``tests/golden/make_tf_shim_fixtures.py`` only.
"""
from .cpu_oracle import (  # noqa: F401
    Oracle,
    build,
    have_ref,
    port,
    ref,
)


def pointwise_nn_guide(inp, conv1, conv2):
    """float32 numpy restatement of HDRNetPointwiseNNGuide._guide with batch-norm folded
    (hdrnet/models.py:203-210; parameter layout of hdrnet/bin/freeze_graph.py:170-184):
    guide = sigmoid(conv2[n] + sum_k conv2[k] * relu(conv1[k][Cin] + sum_j conv1[k][j] * in_j))."""
    import numpy as np
    inp = np.asarray(inp, np.float32)
    conv1 = np.asarray(conv1, np.float32)
    conv2 = np.asarray(conv2, np.float32)
    h = inp @ conv1[:, :-1].T + conv1[:, -1]
    t = (np.maximum(h, np.float32(0)) @ conv2[:-1] + conv2[-1]).astype(np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-t))).astype(np.float32)


def curves_guide(inp, ccm34, shifts, slopes, mix):
    """float32 numpy restatement of HDRNetCurves._guide (hdrnet/models.py:145-190) on the exported
    parameters (hdrnet/bin/freeze_graph.py:107-127): ccm34 [3, 4] (row = output channel: 3 weights
    + bias), shifts / slopes [npts, 3], mix [4] (3 weights + bias):
    t = in @ ccm + bias; c = sum_k slopes_k * relu(t - shifts_k); guide = clip(c @ mix_w + mix_b, 0, 1)."""
    import numpy as np
    inp = np.asarray(inp, np.float32)
    ccm34, shifts, slopes, mix = (np.asarray(a, np.float32) for a in (ccm34, shifts, slopes, mix))
    t = (inp @ ccm34[:, :-1].T + ccm34[:, -1]).astype(np.float32)            # [..., 3]
    c = (slopes * np.maximum(t[..., None, :] - shifts, np.float32(0))).sum(-2, dtype=np.float32)
    g = (c @ mix[:-1] + mix[-1]).astype(np.float32)
    return np.clip(g, np.float32(0), np.float32(1))


def resize_bilinear_align_corners(x, height, width):
    """float32 numpy restatement of ``tf.image.resize_images(x, (height, width), BILINEAR,
    align_corners=True)`` on NHWC arrays -- the resize of HDRNetGaussianPyrNN
    (hdrnet/models.py:253-266, :283-286).  TensorFlow is a dependency of the reference that is not
    under /root/reference (hdrnet/requirements.txt: tensorflow_gpu==2.12.0); its published legacy
    algorithm (tensorflow/core/kernels/image/resize_bilinear_op.cc, half_pixel_centers = false):
    scale = (in - 1) / float(out - 1) (in / float(out) if out == 1); src = i * scale;
    lower = floor(src); upper = min(ceil(src), in - 1); lerp = src - lower;
    out = top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp."""
    import numpy as np
    x = np.asarray(x, np.float32)
    B, Hin, Win, C = x.shape

    def axis(n_in, n_out):
        scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(n_in) / np.float32(n_out)
        src = (np.arange(n_out, dtype=np.float32) * scale).astype(np.float32)
        lo = np.floor(src)
        hi = np.minimum(np.ceil(src), n_in - 1)
        return lo.astype(np.int64), hi.astype(np.int64), (src - lo).astype(np.float32)

    y0, y1, ly = axis(Hin, int(height))
    x0, x1, lx = axis(Win, int(width))
    lx = lx[None, None, :, None]
    ly = ly[None, :, None, None]
    r0, r1 = x[:, y0], x[:, y1]
    top = r0[:, :, x0] + (r0[:, :, x1] - r0[:, :, x0]) * lx
    bot = r1[:, :, x0] + (r1[:, :, x1] - r1[:, :, x0]) * lx
    return (top + (bot - top) * ly).astype(np.float32)
