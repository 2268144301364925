"""hdrnet_amd -- MI355X-native BilateralSlice / BilateralSliceApply (google/hdrnet hot path).

Host side mirrors the reference's operator boundary:

* ``hdrnet_amd.hdrnet_ops``  <->  ``hdrnet/hdrnet_ops.py``  (bilateral_slice,
  bilateral_slice_apply and their registered gradients)
* ``hdrnet_amd.layers``      <->  the two slice wrappers of ``hdrnet/layers.py:99-148``

Compute is hand-written HIP for gfx950 behind the C-ABI of ``include/hdrnet_amd.h``
(``hdrnet_amd/lib/libhdrnet_amd.so``).  PyTorch is used for device memory, streams
and ``torch.distributed`` only.
"""
from . import hdrnet_ops, layers  # noqa: F401
from .hdrnet_ops import bilateral_slice, bilateral_slice_apply  # noqa: F401

__version__ = "0.1.0"
