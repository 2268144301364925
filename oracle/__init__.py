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
