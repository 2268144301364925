"""Python interface to the MI355X bilateral-grid ops -- the drop-in for
``hdrnet/hdrnet_ops.py`` of the reference.

Same two callables, same argument order / keyword, same NHWC float32 layouts, same
registered gradients:

=============================================  =======================================
reference (TensorFlow custom op)               here (torch.autograd.Function on ROCm)
=============================================  =======================================
``hdrnet_ops.bilateral_slice(grid, guide)``    ``bilateral_slice(grid, guide)``
  hdrnet/hdrnet_ops.py:30                        -> [B, H, W, C]
``hdrnet_ops.bilateral_slice_apply(grid,       ``bilateral_slice_apply(grid, guide,
  guide, input, has_offset=...)``  :31           input, has_offset=...)`` -> [B, H, W, Cout]
``@RegisterGradient('BilateralSlice')`` :34    ``_BilateralSlice.backward``  (dgrid, dguide)
``@RegisterGradient('BilateralSliceApply')``   ``_BilateralSliceApply.backward``
  :41-48                                         (dgrid, dguide, dinput)
=============================================  =======================================

Shapes (``bilateral_slice_apply_op.cc:147-193``): grid ``[B, GH, GW, GD, Cout*Cj]`` with
``Cj = Cin + has_offset`` and channel ``c = i*Cj + j``; guide ``[B, H, W]``; input
``[B, H, W, Cin]``.  Violations raise ``ValueError`` (TF: ``InvalidArgument``).

Every call goes through the C-ABI of ``include/hdrnet_amd.h`` on the current HIP stream
of the tensors' device, asynchronously.  There is no CPU or eager fallback: tensors
must live on an AMD GPU and ``libhdrnet_amd.so`` must load, otherwise the call raises.
"""
from __future__ import annotations

import contextlib
import ctypes
import threading
from typing import Optional, Tuple

import torch

from . import _lib

__all__ = ["bilateral_slice", "bilateral_slice_apply", "bilateral_slice_apply_rows", "bilateral_slice_apply_nnguide",
           "bilateral_slice_apply_io", "bilateral_slice_apply_curves", "bilateral_slice_apply_upadd", "resize_bilinear", "input_moments",
           "CoefficientWeights", "coefficients", "coefficients_train", "coefficients_train_supported", "guide_fold_batch", "guide_nn_prescale", "curves_guide_prepare",
           "kernel_override", "last_kernel"]

_tls = threading.local()


def _flags() -> int:
    return getattr(_tls, "flags", _lib.KERNEL_AUTO)


@contextlib.contextmanager
def kernel_override(which: str, variant: int = 0):
    """Force a kernel family inside the block: 'auto' | 'generic' | 'fast' (tests/bench).
    `variant` > 0 selects a benchmark-only kernel of the TOOLS build (csrc/apply_fwd_variants.hip,
    tools/ab_bench.py); the product library these ops call has no variants and answers any non-zero
    variant with HDRNET_INVALID_ARGUMENT (raised as HdrnetInvalidArgument)."""
    table = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "fast": _lib.KERNEL_FAST}
    old = _flags()
    _tls.flags = table[which] | ((int(variant) & 0xFF) << 8)
    try:
        yield
    finally:
        _tls.flags = old


def last_kernel() -> str:
    """Name of the kernel variant this thread's last call launched."""
    return _lib.last_kernel()


# ---- argument checking (mirrors the OP_REQUIRES of the reference op wrappers) --------
def _require_f32(name: str, t: torch.Tensor) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the op is registered for float only), got {t.dtype}")


def _require_gpu(name: str, t: torch.Tensor) -> None:
    # Checked AFTER the shape rules so that shape errors surface as ValueError on any
    # device (tests/test_host_logic.py runs them without a GPU).
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: hdrnet_amd ops run on an MI355X (HIP) device only; "
            "there is no CPU path in this package")


def _check_slice(grid: torch.Tensor, guide: torch.Tensor) -> Tuple[int, ...]:
    _require_f32("grid", grid)
    _require_f32("guide", guide)
    if grid.dim() != 5:
        raise ValueError("Grid should be 5D (batch_size, grid_height, grid_width, grid_depth, "
                         f"grid_channels), got {tuple(grid.shape)}")
    if guide.dim() != 3:
        raise ValueError(f"Guide image should be 3D (batch_size, height, width), got {tuple(guide.shape)}")
    if grid.device != guide.device:
        raise ValueError("grid and guide must be on the same device")
    B, GH, GW, GD, C = grid.shape
    if guide.shape[0] != B:
        raise ValueError("Batch sizes should match.")
    if min(GH, GW, GD, C) <= 0:
        raise ValueError(f"grid extents must be positive, got {tuple(grid.shape)}")
    _require_gpu("grid", grid)
    _require_gpu("guide", guide)
    return B, guide.shape[1], guide.shape[2], GH, GW, GD, C


def _check_apply(grid, guide, inp, has_offset: bool) -> Tuple[int, ...]:
    _require_f32("grid", grid)
    _require_f32("guide", guide)
    _require_f32("input", inp)
    if grid.dim() != 5:
        raise ValueError("Input grid should be 5D (batch_size, height, width, depth, "
                         f"output_channels * input_channels), got {tuple(grid.shape)}")
    if guide.dim() != 3:
        raise ValueError(f"Guide image should be 3D (batch_size, height, width), got {tuple(guide.shape)}")
    if inp.dim() != 4:
        raise ValueError("Input image should be 4D (batch_size, height, width, input_channels), "
                         f"got {tuple(inp.shape)}")
    if not (grid.device == guide.device == inp.device):
        raise ValueError("grid, guide and input must be on the same device")
    B, GH, GW, GD, C = grid.shape
    if tuple(inp.shape[:3]) != tuple(guide.shape):
        raise ValueError("Input and guide size should match.")
    if guide.shape[0] != B:
        raise ValueError("Batch sizes should match.")
    Cin = inp.shape[3]
    Cj = Cin + (1 if has_offset else 0)
    if Cj <= 0 or C % Cj != 0 or C == 0:
        if has_offset:
            raise ValueError("Slicing with affine offset, grid should have "
                             "output_channels * (input_channels + 1) channels.")
        raise ValueError("Slicing without affine offset, grid should have "
                         "output_channels * input_channels channels.")
    if min(GH, GW, GD) <= 0:
        raise ValueError(f"grid extents must be positive, got {tuple(grid.shape)}")
    for n, t in (("grid", grid), ("guide", guide), ("input", inp)):
        _require_gpu(n, t)
    return B, guide.shape[1], guide.shape[2], GH, GW, GD, Cin, C // Cj


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- raw forward / backward launches (no autograd) ------------------------------------
def _apply_forward(grid, guide, inp, has_offset: bool) -> torch.Tensor:
    B, H, W, GH, GW, GD, Cin, Cout = _check_apply(grid, guide, inp, has_offset)
    grid, guide, inp = grid.contiguous(), guide.contiguous(), inp.contiguous()
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=guide.device)
    lib = _lib.load()
    with torch.cuda.device(guide.device):
        rc = lib.hdrnet_bilateral_slice_apply_f32_ex(
            grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), out.data_ptr(),
            B, H, W, GH, GW, GD, Cin, Cout, int(has_offset), _flags(), _stream(guide.device))
    _lib.check(rc, "BilateralSliceApply")
    return out


def _apply_backward(grid, guide, inp, dout, has_offset: bool, need, flags: Optional[int] = None):
    B, H, W, GH, GW, GD, Cin, Cout = _check_apply(grid, guide, inp, has_offset)
    if tuple(dout.shape) != (B, H, W, Cout):
        raise ValueError(f"backprop should have shape {(B, H, W, Cout)}, got {tuple(dout.shape)}")
    _require_f32("backprop", dout)
    _require_gpu("backprop", dout)
    grid, guide, inp, dout = grid.contiguous(), guide.contiguous(), inp.contiguous(), dout.contiguous()
    dev = guide.device
    dgrid = torch.empty_like(grid) if need[0] else None
    dguide = torch.empty_like(guide) if need[1] else None
    dinput = torch.empty_like(inp) if need[2] else None
    lib = _lib.load()
    with torch.cuda.device(dev):  # workspace sizes may depend on the CURRENT device (CU count)
        wbytes = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(
            B, H, W, GH, GW, GD, Cin, Cout, int(has_offset)) if need[0] else 0
        ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev) if wbytes else None
        rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
            grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
            _ptr(dgrid), _ptr(dguide), _ptr(dinput),
            B, H, W, GH, GW, GD, Cin, Cout, int(has_offset),
            _ptr(ws), wbytes, _flags() if flags is None else flags, _stream(dev))
    _lib.check(rc, "BilateralSliceApplyGrad")
    return dgrid, dguide, dinput


def _slice_forward(grid, guide) -> torch.Tensor:
    B, H, W, GH, GW, GD, C = _check_slice(grid, guide)
    grid, guide = grid.contiguous(), guide.contiguous()
    out = torch.empty((B, H, W, C), dtype=torch.float32, device=guide.device)
    lib = _lib.load()
    with torch.cuda.device(guide.device):
        rc = lib.hdrnet_bilateral_slice_f32_ex(
            grid.data_ptr(), guide.data_ptr(), out.data_ptr(),
            B, H, W, GH, GW, GD, C, _flags(), _stream(guide.device))
    _lib.check(rc, "BilateralSlice")
    return out


def _slice_backward(grid, guide, dout, need, flags: Optional[int] = None):
    B, H, W, GH, GW, GD, C = _check_slice(grid, guide)
    if dout.dim() != 4:
        raise ValueError("Codomain tangent should be 4D (batch, height, width, nchannels).")
    if tuple(dout.shape) != (B, H, W, C):
        raise ValueError(f"backprop should have shape {(B, H, W, C)}, got {tuple(dout.shape)}")
    _require_f32("backprop", dout)
    _require_gpu("backprop", dout)
    grid, guide, dout = grid.contiguous(), guide.contiguous(), dout.contiguous()
    dev = guide.device
    dgrid = torch.empty_like(grid) if need[0] else None
    dguide = torch.empty_like(guide) if need[1] else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        wbytes = lib.hdrnet_bilateral_slice_grad_workspace_bytes(B, H, W, GH, GW, GD, C) if need[0] else 0
        ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev) if wbytes else None
        rc = lib.hdrnet_bilateral_slice_grad_f32_ex(
            grid.data_ptr(), guide.data_ptr(), dout.data_ptr(), _ptr(dgrid), _ptr(dguide),
            B, H, W, GH, GW, GD, C, _ptr(ws), wbytes, _flags() if flags is None else flags, _stream(dev))
    _lib.check(rc, "BilateralSliceGrad")
    return dgrid, dguide


# ---- autograd registration (hdrnet/hdrnet_ops.py:34-48) -------------------------------
class _BilateralSlice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, guide):
        ctx.save_for_backward(grid, guide)
        # autograd runs backward on its own thread: the (thread-local) kernel override in
        # effect at forward time is carried along explicitly.
        ctx.flags = _flags()
        return _slice_forward(grid, guide)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        grid, guide = ctx.saved_tensors
        need = ctx.needs_input_grad
        if not (need[0] or need[1]):
            return None, None
        return _slice_backward(grid, guide, grad, need, ctx.flags)


class _BilateralSliceApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, guide, inp, has_offset):
        ctx.save_for_backward(grid, guide, inp)
        ctx.has_offset = bool(has_offset)
        ctx.flags = _flags()
        return _apply_forward(grid, guide, inp, bool(has_offset))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        grid, guide, inp = ctx.saved_tensors
        need = ctx.needs_input_grad[:3]
        if not any(need):
            return None, None, None, None
        dgrid, dguide, dinput = _apply_backward(grid, guide, inp, grad, ctx.has_offset, need, ctx.flags)
        return dgrid, dguide, dinput, None


def bilateral_slice(grid: torch.Tensor, guide: torch.Tensor, name: Optional[str] = None) -> torch.Tensor:
    """``BilateralSlice``: out[b,y,x,c] = trilinear sample of grid[b,...,c] at
    ((x+.5)*GW/W, (y+.5)*GH/H, guide[b,y,x]*GD).  (bilateral_slice_op.cc:274-290)"""
    del name
    return _BilateralSlice.apply(grid, guide)


def bilateral_slice_apply(grid: torch.Tensor, guide: torch.Tensor, input: torch.Tensor,  # noqa: A002
                          has_offset: bool, name: Optional[str] = None) -> torch.Tensor:
    """``BilateralSliceApply``: slice then per-pixel (Cout x Cj) . [input; 1].
    ``has_offset`` is a required attribute, as in the reference op
    (bilateral_slice_apply_op.cc:382-386)."""
    del name
    return _BilateralSliceApply.apply(grid, guide, input, has_offset)


def bilateral_slice_apply_rows(grid: torch.Tensor, guide_rows: torch.Tensor, input_rows: torch.Tensor,
                               frame_height: int, y0: int, has_offset: bool) -> torch.Tensor:
    """Row-split ``BilateralSliceApply`` forward (the optional intra-image split):
    ``guide_rows`` [B, rows, W] and ``input_rows`` [B, rows, W, Cin] hold rows ``y0 .. y0 + rows - 1`` of
    frames that are ``frame_height`` rows high; the grid is the whole frame's.  The y coordinate is the
    reference's expression on the frame height (bilateral_slice_apply.cc:38,42), so the bands of any
    partition (``dist.row_range``), concatenated, equal the whole-frame op bit for bit.  Forward only."""
    if grid.requires_grad or guide_rows.requires_grad or input_rows.requires_grad:
        if torch.is_grad_enabled():
            raise RuntimeError("bilateral_slice_apply_rows is forward-only (inference); detach the operands")
    B, rows, W, GH, GW, GD, Cin, Cout = _check_apply(grid, guide_rows, input_rows, has_offset)
    frame_height, y0 = int(frame_height), int(y0)
    if y0 < 0 or y0 + rows > frame_height:
        raise ValueError(f"row band [{y0}, {y0 + rows}) outside the frame's {frame_height} rows")
    grid, guide_rows, input_rows = grid.contiguous(), guide_rows.contiguous(), input_rows.contiguous()
    out = torch.empty((B, rows, W, Cout), dtype=torch.float32, device=guide_rows.device)
    lib = _lib.load()
    with torch.cuda.device(guide_rows.device):
        rc = lib.hdrnet_bilateral_slice_apply_rows_f32_ex(
            grid.data_ptr(), guide_rows.data_ptr(), input_rows.data_ptr(), out.data_ptr(),
            B, frame_height, y0, rows, W, GH, GW, GD, Cin, Cout, int(has_offset), _flags(),
            _stream(guide_rows.device))
    _lib.check(rc, "BilateralSliceApplyRows")
    return out


def _guide_flags(fast_sigmoid: bool, prescaled: bool) -> int:
    return (_lib.GUIDE_SIGMOID_FAST if fast_sigmoid else 0) | (_lib.GUIDE_RELU_PRESCALED if prescaled else 0)


def guide_nn_prescale(guide_conv1: torch.Tensor, guide_conv2: torch.Tensor, x_max: float = 65536.0):
    """The PRESCALED form of a folded guide network (``hdrnet_guide_nn_prescale_f32``; Cin = 3): per feature k the
    first-layer row reordered to ``{w0, b, w1, w2} * 2**-e_k`` and the mixing weight ``* 2**e_k`` with
    ``2**e_k >= 2 (|b_k| + x_max sum_j |w_kj|)``.  Passed back to the guide-network forwards with ``prescaled=True``
    (``HDRNET_GUIDE_RELU_PRESCALED``) the guide is the plain evaluation's bit for bit for every input with
    ``|x| <= x_max`` -- the kernels then take ``relu`` from the clamp modifier of the feature's last multiply-add
    (128 instead of 208 vector instructions per 256 pixels).  Returns ``(conv1 [n, 4], conv2 [n + 1])``; prepare once
    per parameter set.  Inference only: the guide network's VJP reads the exported layout."""
    _require_f32("guide_conv1", guide_conv1)
    _require_f32("guide_conv2", guide_conv2)
    _require_gpu("guide_conv1", guide_conv1)
    _require_gpu("guide_conv2", guide_conv2)
    if guide_conv1.dim() != 2 or guide_conv1.shape[1] != 4 or tuple(guide_conv2.shape) != (guide_conv1.shape[0] + 1,):
        raise ValueError(f"guide_nn_prescale: guide_conv1 [n, 4] (Cin = 3) and guide_conv2 [n + 1] expected, got "
                         f"{tuple(guide_conv1.shape)}, {tuple(guide_conv2.shape)}")
    c1, c2 = guide_conv1.detach().contiguous(), guide_conv2.detach().contiguous()
    o1, o2 = torch.empty_like(c1), torch.empty_like(c2)
    dev = c1.device
    with torch.cuda.device(dev):
        rc = _lib.load().hdrnet_guide_nn_prescale_f32(c1.data_ptr(), c2.data_ptr(), c1.shape[0], 3, float(x_max),
                                                      o1.data_ptr(), o2.data_ptr(), _stream(dev))
    _lib.check(rc, "GuideNNPrescale")
    return o1, o2


def _check_nnguide(grid, input, guide_conv1, guide_conv2, has_offset):  # noqa: A002
    _require_f32("guide_conv1", guide_conv1)
    _require_f32("guide_conv2", guide_conv2)
    if input.dim() != 4:
        raise ValueError(f"Input image should be 4D (batch_size, height, width, input_channels), got {tuple(input.shape)}")
    Cin = input.shape[3]
    if guide_conv1.dim() != 2 or guide_conv1.shape[1] != Cin + 1:
        raise ValueError(f"guide_conv1 should be [n, Cin + 1] = [n, {Cin + 1}], got {tuple(guide_conv1.shape)}")
    n = guide_conv1.shape[0]
    if tuple(guide_conv2.shape) != (n + 1,):
        raise ValueError(f"guide_conv2 should be [n + 1] = [{n + 1}], got {tuple(guide_conv2.shape)}")
    fake_guide = input[..., 0]  # shape carrier for the shared rule checks; never read
    dims = _check_apply(grid, fake_guide, input, has_offset)
    for nm, t in (("guide_conv1", guide_conv1), ("guide_conv2", guide_conv2)):
        _require_gpu(nm, t)
    return dims + (n,)


def _nnguide_forward(grid, inp, c1, c2, has_offset: bool, want_guide: bool, fast_sigmoid: bool = False,
                     prescaled: bool = False):
    B, H, W, GH, GW, GD, Cin, Cout, n = _check_nnguide(grid, inp, c1, c2, has_offset)
    grid, inp, c1, c2 = grid.contiguous(), inp.contiguous(), c1.contiguous(), c2.contiguous()
    dev = inp.device
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
    gout = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_guide else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.hdrnet_bilateral_slice_apply_nnguide_f32_ex(
            grid.data_ptr(), inp.data_ptr(), c1.data_ptr(), c2.data_ptr(), out.data_ptr(), _ptr(gout),
            B, H, W, GH, GW, GD, Cin, Cout, int(bool(has_offset)), n,
            _guide_flags(fast_sigmoid, prescaled), _stream(dev))
    _lib.check(rc, "BilateralSliceApplyNNGuide")
    return out, gout


def _guide_backward(inp, guide, dguide, c1, c2, dinput, accumulate: bool):
    """VJP of the folded guide network: returns (dconv1, dconv2); adds the guide path's share to
    ``dinput`` in place (or stores it, or skips it when ``dinput`` is None)."""
    inp, guide, dguide = inp.contiguous(), guide.contiguous(), dguide.contiguous()
    c1, c2 = c1.contiguous(), c2.contiguous()
    dev = inp.device
    npx, Cin, n = guide.numel(), inp.shape[-1], c1.shape[0]
    dc1, dc2 = torch.empty_like(c1), torch.empty_like(c2)
    lib = _lib.load()
    with torch.cuda.device(dev):
        wbytes = lib.hdrnet_pointwise_guide_grad_workspace_bytes(npx, Cin, n)
        ws = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
        rc = lib.hdrnet_pointwise_guide_grad_f32(
            inp.data_ptr(), guide.data_ptr(), dguide.data_ptr(), c1.data_ptr(), c2.data_ptr(),
            _ptr(dinput), int(bool(accumulate)), dc1.data_ptr(), dc2.data_ptr(), npx, Cin, n,
            ws.data_ptr(), wbytes, _stream(dev))
    _lib.check(rc, "PointwiseGuideGrad")
    return dc1, dc2


class _BilateralSliceApplyNNGuide(torch.autograd.Function):
    """guide network + slice-apply as ONE differentiable op: forward = the fused kernel (the
    guide is written once, for the backward); backward = slice-apply VJP, then the guide
    network's VJP, which adds its share into the same dinput buffer."""

    @staticmethod
    def forward(ctx, grid, inp, c1, c2, has_offset):
        out, guide = _nnguide_forward(grid, inp, c1, c2, bool(has_offset), want_guide=True)
        ctx.save_for_backward(grid, inp, c1, c2, guide)
        ctx.has_offset = bool(has_offset)
        ctx.flags = _flags()
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        grid, inp, c1, c2, guide = ctx.saved_tensors
        need_grid, need_inp, need_c1, need_c2 = ctx.needs_input_grad[:4]
        need_net = need_c1 or need_c2
        if not (need_grid or need_inp or need_net):
            return None, None, None, None, None
        dgrid, dguide, dinput = _apply_backward(grid, guide, inp, grad, ctx.has_offset,
                                                (need_grid, need_net or need_inp, need_inp), ctx.flags)
        dc1 = dc2 = None
        if need_net or need_inp:
            dc1, dc2 = _guide_backward(inp, guide, dguide, c1, c2, dinput, accumulate=True)
        return dgrid, dinput, dc1 if need_c1 else None, dc2 if need_c2 else None, None


def bilateral_slice_apply_nnguide(grid: torch.Tensor, input: torch.Tensor,  # noqa: A002
                                  guide_conv1: torch.Tensor, guide_conv2: torch.Tensor,
                                  has_offset: bool = True, return_guide: bool = False, fast_sigmoid: bool = False,
                                  prescaled: bool = False):
    """Fusion of ``HDRNetPointwiseNNGuide._guide`` (hdrnet/models.py:203-210, batch norm folded)
    with ``bilateral_slice_apply``: the guide is computed in registers and sliced immediately.

    ``fast_sigmoid`` (forward without autograd only): the hardware exp / reciprocal sigmoid instead of
    ``tf.nn.sigmoid``'s form -- <= 2 ulp of the guide, ~1e-6 of the output's scale, ~10 % faster; an explicit choice
    (``HDRNET_GUIDE_SIGMOID_FAST``).  A differentiable call always uses the exact form: the backward reads the guide.

    ``prescaled`` (forward without autograd only): ``guide_conv1`` / ``guide_conv2`` are ``guide_nn_prescale``'s arrays
    (``HDRNET_GUIDE_RELU_PRESCALED``) -- the same guide bit for bit for inputs within the prescale's ``x_max``.

    ``guide_conv1`` is ``[n, Cin + 1]`` (weights then bias of feature k) and ``guide_conv2``
    ``[n + 1]`` (mixing weights then bias) -- the layout ``hdrnet/bin/freeze_graph.py:170-184``
    exports as ``guide_conv1.bin`` / ``guide_conv2.bin``.

    Differentiable in ``grid``, ``input``, ``guide_conv1`` and ``guide_conv2`` (the guide is then
    written once for the backward, whose guide-network VJP exists for ``Cin`` in {1, 3} and ``n`` in
    {4, 8, 16}).  ``return_guide=True`` returns ``(out, guide)`` without autograd.  Same shape rules
    as ``bilateral_slice_apply``; raises ``ValueError`` where the fused kernel has no
    specialisation."""
    if return_guide:
        return _nnguide_forward(grid.detach(), input.detach(), guide_conv1.detach(), guide_conv2.detach(),
                                has_offset, want_guide=True, fast_sigmoid=fast_sigmoid, prescaled=prescaled)
    if torch.is_grad_enabled() and any(t.requires_grad for t in (grid, input, guide_conv1, guide_conv2)):
        if prescaled:
            raise ValueError("bilateral_slice_apply_nnguide: prescaled guide parameters are inference-only "
                             "(the guide network's VJP reads the exported layout)")
        dims = _check_nnguide(grid, input, guide_conv1, guide_conv2, has_offset)
        # the guide-network VJP has specialisations for a few widths only: say so NOW, not from
        # inside backward() on the autograd thread
        with torch.cuda.device(input.device):
            if _lib.load().hdrnet_pointwise_guide_grad_workspace_bytes(1024, dims[6], dims[8]) == 0:
                raise ValueError(
                    f"bilateral_slice_apply_nnguide: no guide-network VJP for Cin = {dims[6]}, n_feats = {dims[8]} "
                    "(Cin in {1, 3}, n_feats in {4, 8, 16}); compose the guide network from torch ops and "
                    "call bilateral_slice_apply, or run without autograd")
        return _BilateralSliceApplyNNGuide.apply(grid, input, guide_conv1, guide_conv2, has_offset)
    return _nnguide_forward(grid.detach(), input.detach(), guide_conv1.detach(), guide_conv2.detach(),
                            has_offset, want_guide=False, fast_sigmoid=fast_sigmoid, prescaled=prescaled)[0]


class _BilateralSliceApplyCurves(torch.autograd.Function):
    """curves guide + slice-apply as ONE differentiable op (the standard model's training path):
    forward = the fused kernel (guide written once for the backward); backward = slice-apply VJP,
    then the curves guide's VJP, which adds its share into the same dinput buffer."""

    @staticmethod
    def forward(ctx, grid, inp, ccm, shifts, slopes, mix, has_offset):
        out, guide = _apply_io_curves(grid, inp, (ccm, shifts, slopes, mix), None, torch.float32,
                                      bool(has_offset), True)
        ctx.save_for_backward(grid, inp, ccm, shifts, slopes, mix, guide)
        ctx.has_offset = bool(has_offset)
        ctx.flags = _flags()
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        grid, inp, ccm, shifts, slopes, mix, guide = ctx.saved_tensors
        need = ctx.needs_input_grad
        need_net = any(need[2:6])
        if not (need[0] or need[1] or need_net):
            return (None,) * 7
        dgrid, dguide, dinput = _apply_backward(grid, guide, inp, grad, ctx.has_offset,
                                                (need[0], need_net or need[1], need[1]), ctx.flags)
        grads = [None] * 4
        if need_net or need[1]:
            inp_c, dguide = inp.contiguous(), dguide.contiguous()
            params = [t.contiguous() for t in (ccm, shifts, slopes, mix)]
            outs = [torch.empty_like(t) for t in params]
            npx, npts = dguide.numel(), shifts.shape[0]
            lib = _lib.load()
            with torch.cuda.device(inp.device):
                wbytes = lib.hdrnet_curves_guide_grad_workspace_bytes(npx, 3, npts)
                ws = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=inp.device)
                rc = lib.hdrnet_curves_guide_grad_f32(
                    inp_c.data_ptr(), dguide.data_ptr(), *[t.data_ptr() for t in params], _ptr(dinput), 1,
                    *[t.data_ptr() for t in outs], npx, 3, npts, ws.data_ptr(), wbytes, _stream(inp.device))
            _lib.check(rc, "CurvesGuideGrad")
            grads = [g if n else None for g, n in zip(outs, need[2:6])]
        return (dgrid, dinput, *grads, None)


def bilateral_slice_apply_curves(grid: torch.Tensor, input: torch.Tensor, ccm: torch.Tensor,  # noqa: A002
                                 shifts: torch.Tensor, slopes: torch.Tensor, mix: torch.Tensor,
                                 has_offset: bool = True, prepared: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``HDRNetCurves._guide`` (hdrnet/models.py:145-190) fused with ``bilateral_slice_apply``, fp32,
    differentiable in ``grid``, ``input`` and the four guide parameter arrays (layouts of
    hdrnet/bin/freeze_graph.py:107-127: ccm [3, 4], shifts / slopes [16, 3], mix [4]).  The wire-format
    inference variant is ``bilateral_slice_apply_io(..., guide_curves=...)``.  ``prepared`` (forward without autograd
    only): the tables of ``curves_guide_prepare(shifts, slopes)``."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (grid, input, ccm, shifts, slopes, mix)):
        if prepared is not None:
            raise ValueError("bilateral_slice_apply_curves: prepared tables are inference-only")
        if input.dim() != 4 or input.shape[3] != 3 or shifts.dim() != 2 or shifts.shape[0] != 16:
            raise ValueError("the differentiable curves-guide op needs Cin = 3 and 16 knots per channel")
        return _BilateralSliceApplyCurves.apply(grid, input, ccm, shifts, slopes, mix, has_offset)
    return _apply_io_curves(grid, input, (ccm, shifts, slopes, mix), None, torch.float32, has_offset, False, prepared)


def input_moments(input: torch.Tensor):  # noqa: A002
    """``(sum_px in_j [Cin], sum_px in_i * in_j [Cin, Cin])`` over every pixel of ``input``
    ``[..., Cin]`` in one pass.  The first guide convolution is linear, so these give the batch
    statistics its batch norm needs in training mode (hdrnet/layers.py:40-58) without the
    n-channel full-resolution tensor.  No autograd."""
    _require_f32("input", input)
    _require_gpu("input", input)
    inp = input.detach().contiguous()
    Cin = inp.shape[-1]
    npx = inp.numel() // max(Cin, 1)
    dev = inp.device
    sums = torch.empty((Cin,), dtype=torch.float32, device=dev)
    mom = torch.empty((Cin, Cin), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        wbytes = lib.hdrnet_input_moments_workspace_bytes(npx, Cin)
        ws = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
        rc = lib.hdrnet_input_moments_f32(inp.data_ptr(), npx, Cin, sums.data_ptr(), mom.data_ptr(),
                                          ws.data_ptr(), wbytes, _stream(dev))
    _lib.check(rc, "InputMoments")
    return sums, mom


class _GuideFoldBatch(torch.autograd.Function):
    """Training-mode fold of the guide network's batch norm (``hdrnet_guide_fold_batch_f32`` and its VJP): one launch
    each way where the same float64 math on ~100 numbers was 33 + 30 torch launches of a graph-captured step."""

    @staticmethod
    def forward(ctx, w1, beta, w2, b2, gamma, sums, moments, running_mean, running_var, num_batches_tracked,
                npx, eps, momentum):
        Cin, n = w1.shape
        dev = w1.device
        args = [t.detach().contiguous() for t in (sums, moments, w1, gamma, beta, w2, b2.reshape(1))]
        conv1 = torch.empty((n, Cin + 1), dtype=torch.float32, device=dev)
        conv2 = torch.empty((n + 1,), dtype=torch.float32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.hdrnet_guide_fold_batch_f32(
                args[0].data_ptr(), args[1].data_ptr(), int(npx), args[2].data_ptr(), args[3].data_ptr(),
                args[4].data_ptr(), args[5].data_ptr(), args[6].data_ptr(), float(eps), float(momentum), Cin, n,
                conv1.data_ptr(), conv2.data_ptr(), _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                _stream(dev))
        _lib.check(rc, "GuideFoldBatch")
        for t in (running_mean, running_var, num_batches_tracked):
            if t is not None:  # written through a raw pointer: tell autograd / the fold caches keyed on ._version
                torch.autograd.graph.increment_version(t)
        ctx.save_for_backward(*args[:5])
        ctx.meta = (int(npx), float(eps), Cin, n, b2.shape)
        ctx.leaves = (w1, beta, w2, b2)  # where their gradients may be written directly (_grad_out)
        return conv1, conv2

    @staticmethod
    def backward(ctx, dconv1, dconv2):
        sums, moments, w1, gamma, beta = ctx.saved_tensors
        npx, eps, Cin, n, b2_shape = ctx.meta
        dev = w1.device
        dconv1, dconv2 = dconv1.contiguous(), dconv2.contiguous()
        outs = []
        for leaf, shape in zip(ctx.leaves, ((Cin, n), (n,), (n,), tuple(b2_shape))):
            g = _grad_out(leaf) if tuple(leaf.shape) == shape and leaf.is_contiguous() else None
            outs.append(g if g is not None and g.is_contiguous() else torch.empty(shape, dtype=torch.float32, device=dev))
        dw1, dbeta, dw2, db2 = outs
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.hdrnet_guide_fold_batch_grad_f32(
                sums.data_ptr(), moments.data_ptr(), npx, w1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, Cin, n,
                dconv1.data_ptr(), dconv2.data_ptr(), dw1.data_ptr(), dbeta.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                _stream(dev))
        _lib.check(rc, "GuideFoldBatchGrad")
        return dw1, dbeta, dw2, db2, None, None, None, None, None, None, None, None, None


def guide_fold_batch(w1: torch.Tensor, beta: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, gamma: torch.Tensor,
                     sums: torch.Tensor, moments: torch.Tensor, npx: int, eps: float, momentum: float = 0.0,
                     running_mean: Optional[torch.Tensor] = None, running_var: Optional[torch.Tensor] = None,
                     num_batches_tracked: Optional[torch.Tensor] = None):
    """``(conv1 [n, Cin + 1], conv2 [n + 1])``: the point-wise guide network with batch norm in TRAINING mode folded
    into its first layer, the batch statistics taken from the input's ``(sums, moments)`` (``input_moments``) --
    ``hdrnet/layers.py:40-58`` with ``is_training=True`` in the export layout of ``freeze_graph.py:170-184``.
    ``w1`` is ``[Cin, n]``.  Differentiable in ``w1``, ``beta``, ``w2``, ``b2``; moves the running statistics in place."""
    for name, t in (("w1", w1), ("beta", beta), ("w2", w2), ("b2", b2), ("gamma", gamma), ("sums", sums), ("moments", moments)):
        _require_f32(name, t)
        _require_gpu(name, t)
    if w1.dim() != 2 or w1.shape[0] not in (1, 3):
        raise ValueError(f"w1 should be [Cin in (1, 3), n_feats], got {tuple(w1.shape)}")
    return _GuideFoldBatch.apply(w1, beta, w2, b2, gamma, sums, moments, running_mean, running_var,
                                 num_batches_tracked, npx, eps, momentum)


class CoefficientWeights:
    """The coefficient network's parameters in the layout ``hdrnet_coefficients_f32`` reads
    (include/hdrnet_amd.h): convolutions ``[Cout][kh][kw][Cin]``, fully connected layers ``[in][out]``,
    batch norm folded, fp32, contiguous, on one device.  Holds the tensors alive next to the C struct.

    ``params``: net_input_size, spatial_bin, luma_bins, channel_multiplier (hdrnet/bin/train.py:227-236);
    ``splat`` / ``global_conv`` / ``fc`` / ``local``: lists of ``(weight, bias)`` (bias ``None`` only for the second
    local conv, the reference's ``use_bias=False``); ``pred``: ``(weight, bias)``.
    """

    def __init__(self, params, n_out: int, n_in: int, n_levels: int, splat, global_conv, fc, local, pred):
        self.params = dict(params)
        self.n_out, self.n_in, self.n_levels = int(n_out), int(n_in), int(n_levels)
        groups = dict(splat=list(splat), global_conv=list(global_conv), fc=list(fc), local=list(local), pred=[pred])
        if len(groups["global_conv"]) != 2 or len(groups["fc"]) != 3 or len(groups["local"]) != 2 or not 1 <= len(groups["splat"]) <= 8:
            raise ValueError("coefficient network: 1-8 splat layers, 2 global convs, 3 fc layers, 2 local convs")
        self.device = groups["pred"][0][0].device
        self._keep = []

        def prep(t):
            if t is None:
                return None
            if t.dtype != torch.float32:
                raise ValueError(f"coefficient network parameters must be float32, got {t.dtype}")
            if t.device != self.device:
                raise ValueError("coefficient network parameters must live on one device")
            t = t.detach().contiguous()
            self._keep.append(t)
            return t.data_ptr()

        net = _lib.CoeffNet()
        net.net_input_size = int(self.params["net_input_size"])
        net.spatial_bin = int(self.params["spatial_bin"])
        net.luma_bins = int(self.params["luma_bins"])
        net.channel_multiplier = int(self.params["channel_multiplier"])
        net.n_out, net.n_in, net.n_levels = self.n_out, self.n_in, self.n_levels
        for i, (w, b) in enumerate(groups["splat"]):
            net.splat_w[i], net.splat_b[i] = prep(w), prep(b)
        for i, (w, b) in enumerate(groups["global_conv"]):
            net.global_conv_w[i], net.global_conv_b[i] = prep(w), prep(b)
        for i, (w, b) in enumerate(groups["fc"]):
            net.fc_w[i], net.fc_b[i] = prep(w), prep(b)
        for i, (w, b) in enumerate(groups["local"]):
            net.local_w[i], net.local_b[i] = prep(w), prep(b)
        net.pred_w, net.pred_b = prep(pred[0]), prep(pred[1])
        self.net = net
        self.n_splat = len(groups["splat"])

    def supported(self, batch: int = 1) -> bool:
        """False: hyper-parameters outside the kernels' reach (run the stock-op graph instead)."""
        import ctypes
        return _lib.load().hdrnet_coefficients_workspace_bytes(ctypes.byref(self.net), int(batch)) > 0


def coefficients(lowres_input: torch.Tensor, weights: CoefficientWeights) -> torch.Tensor:
    """``HDRNetCurves._coefficients`` (hdrnet/models.py:62-142) in inference mode, on the HIP kernels of
    csrc/coeff_net.hip: ``lowres_input [B, N, N, 3]`` -> ``[B, sb, sb, gd, n_out, n_in]`` (``n_levels`` > 1:
    ``[n_levels, B, sb, sb, gd, n_out / n_levels, n_in]``, every level's grid contiguous).  No autograd."""
    import ctypes
    _require_f32("lowres_input", lowres_input)
    if lowres_input.dim() != 4 or lowres_input.shape[3] != 3:
        raise ValueError(f"lowres_input should be [batch, N, N, 3], got {tuple(lowres_input.shape)}")
    _require_gpu("lowres_input", lowres_input)
    p = weights.params
    N, sb, gd = int(p["net_input_size"]), int(p["spatial_bin"]), int(p["luma_bins"])
    if lowres_input.shape[1] != N or lowres_input.shape[2] != N:
        raise ValueError(f"lowres_input is {tuple(lowres_input.shape[1:3])}, the network was built for {N} x {N}")
    if lowres_input.device != weights.device:
        raise ValueError("lowres_input and the network parameters live on different devices")
    low = lowres_input.detach().contiguous()
    B, dev = low.shape[0], low.device
    L = weights.n_levels
    shape = (B, sb, sb, gd, weights.n_out // L, weights.n_in)
    out = torch.empty((L,) + shape if L > 1 else shape, dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        wbytes = lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(weights.net), B)
        ws = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
        rc = lib.hdrnet_coefficients_f32(low.data_ptr(), ctypes.byref(weights.net), out.data_ptr(), B,
                                         ws.data_ptr(), wbytes, _stream(dev))
    _lib.check(rc, "Coefficients")
    return out


def _live_net(hyper, n_out: int, n_in: int, params, n_splat: int):
    """``hdrnet_coeff_net`` over the module's LIVE parameters (no copies): Conv2d weights must be in channels_last memory
    order (= [Cout][kh][kw][Cin]), Linear weights are taken as they are (fc_layout = 1).  ``params`` in the order
    splat (w, b) x n_splat, global conv (w, b) x 2, fc (w, b) x 3, local1 (w, b), local2 w, prediction (w, b)."""
    net = _lib.CoeffNet()
    net.net_input_size, net.spatial_bin = int(hyper["net_input_size"]), int(hyper["spatial_bin"])
    net.luma_bins, net.channel_multiplier = int(hyper["luma_bins"]), int(hyper["channel_multiplier"])
    net.n_out, net.n_in, net.n_levels, net.fc_layout = int(n_out), int(n_in), 1, 1
    it = iter(params)
    for i in range(n_splat):
        net.splat_w[i], net.splat_b[i] = next(it).data_ptr(), next(it).data_ptr()
    for i in range(2):
        net.global_conv_w[i], net.global_conv_b[i] = next(it).data_ptr(), next(it).data_ptr()
    for i in range(3):
        net.fc_w[i], net.fc_b[i] = next(it).data_ptr(), next(it).data_ptr()
    net.local_w[0], net.local_b[0] = next(it).data_ptr(), next(it).data_ptr()
    net.local_w[1] = next(it).data_ptr()
    net.pred_w, net.pred_b = next(it).data_ptr(), next(it).data_ptr()
    return net


def _params_ok(params) -> bool:
    for p in params:
        if p.dtype != torch.float32 or not p.is_cuda:
            return False
        if p.dim() == 4:
            if not p.is_contiguous(memory_format=torch.channels_last):
                return False
        elif not p.is_contiguous():
            return False
    return True


def coefficients_train_supported(hyper, n_out: int, n_in: int, params, n_splat: int, batch: int) -> bool:
    """True if ``coefficients_train`` can run this network (no batch norm is the caller's business): parameters fp32 on
    the GPU in torch's own layouts, hyper-parameters and batch within the kernels' reach."""
    import ctypes
    params = list(params)
    if len(params) != 2 * n_splat + 4 + 6 + 2 + 1 + 2 or not _params_ok(params):
        return False
    net = _live_net(hyper, n_out, n_in, params, n_splat)
    return _lib.load().hdrnet_coefficients_grad_workspace_bytes(ctypes.byref(net), int(batch)) > 0


def _grad_out(p: torch.Tensor) -> torch.Tensor:
    """Where a kernel-computed gradient of parameter ``p`` goes: a fresh alias of the parameter's segment of a flat
    gradient bucket (``dist.GradBucket`` registers it) while ``.grad`` is released -- autograd ASSIGNS a gradient it is
    handed when ``.grad`` is None, adopting the alias, so the bucket's gather has nothing to copy for this parameter --
    otherwise (no bucket, ``.grad`` bound: accumulation, or the segment already handed out since the release: the parameter
    is used twice and autograd must ADD the second gradient) a new tensor in the parameter's layout.

    That autograd adopts (rather than copies) a gradient it is handed for a leaf whose ``.grad`` is None is PyTorch's current
    behaviour, not a documented contract.  Nothing here depends on it for correctness: should a version copy instead,
    ``.grad`` is a tensor of its own holding the same values, and ``GradBucket.gather()`` -- which compares data pointers,
    not flags -- copies it into the segment as it does for every gradient autograd computed itself; the only loss is the
    copy this alias saves (``tests/test_coeff_net.py`` asserts the adopted case so that a change is noticed)."""
    v = getattr(p, "_hdrnet_grad_view", None)
    if (v is not None and p.grad is None and not getattr(p, "_hdrnet_grad_claimed", True) and v.device == p.device
            and v.dtype == p.dtype and v.shape == p.shape and v.stride() == p.stride()):
        p._hdrnet_grad_claimed = True
        return v.detach()
    return torch.empty_like(p)  # preserve_format: channels_last weights get channels_last grads


class _CoefficientsTrain(torch.autograd.Function):
    """Forward = the inference launch sequence on the live parameters, its workspace kept; backward =
    ``hdrnet_coefficients_grad_f32`` (csrc/coeff_net_train.hip)."""

    @staticmethod
    def forward(ctx, lowres, hyper, n_out, n_in, n_splat, *params):
        import ctypes
        low = lowres.detach().contiguous()
        B, dev = low.shape[0], low.device
        net = _live_net(hyper, n_out, n_in, params, n_splat)
        sb, gd = int(hyper["spatial_bin"]), int(hyper["luma_bins"])
        out = torch.empty((B, sb, sb, gd, n_out, n_in), dtype=torch.float32, device=dev)
        lib = _lib.load()
        with torch.cuda.device(dev):
            wbytes = lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(net), B)
            ws = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
            rc = lib.hdrnet_coefficients_f32(low.data_ptr(), ctypes.byref(net), out.data_ptr(), B, ws.data_ptr(), wbytes,
                                             _stream(dev))
        _lib.check(rc, "Coefficients")
        ctx.save_for_backward(low, ws, *params)
        ctx.meta = (dict(hyper), int(n_out), int(n_in), int(n_splat))
        return out

    @staticmethod
    def backward(ctx, dcoeffs):
        import ctypes
        low, ws = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        hyper, n_out, n_in, n_splat = ctx.meta
        B, dev = low.shape[0], low.device
        net = _live_net(hyper, n_out, n_in, params, n_splat)
        grads = [_grad_out(p) for p in params]
        gr = _lib.CoeffNetGrads()
        it = iter(grads)
        for i in range(n_splat):
            gr.splat_w[i], gr.splat_b[i] = next(it).data_ptr(), next(it).data_ptr()
        for i in range(2):
            gr.global_conv_w[i], gr.global_conv_b[i] = next(it).data_ptr(), next(it).data_ptr()
        for i in range(3):
            gr.fc_w[i], gr.fc_b[i] = next(it).data_ptr(), next(it).data_ptr()
        gr.local_w[0], gr.local_b[0] = next(it).data_ptr(), next(it).data_ptr()
        gr.local_w[1] = next(it).data_ptr()
        gr.pred_w, gr.pred_b = next(it).data_ptr(), next(it).data_ptr()
        dc = dcoeffs.contiguous()
        lib = _lib.load()
        with torch.cuda.device(dev):
            wbytes = lib.hdrnet_coefficients_grad_workspace_bytes(ctypes.byref(net), B)
            ws2 = torch.empty((max(wbytes, 16),), dtype=torch.uint8, device=dev)
            rc = lib.hdrnet_coefficients_grad_f32(low.data_ptr(), ctypes.byref(net), ws.data_ptr(), dc.data_ptr(),
                                                  ctypes.byref(gr), B, ws2.data_ptr(), wbytes, _stream(dev))
        _lib.check(rc, "CoefficientsGrad")
        return (None, None, None, None, None, *grads)


def coefficients_train(lowres_input: torch.Tensor, hyper, n_out: int, n_in: int, params, n_splat: int) -> torch.Tensor:
    """``HDRNetCurves._coefficients`` (hdrnet/models.py:62-142) WITHOUT batch norm, differentiable in its parameters:
    forward and backward on the HIP kernels of csrc/coeff_net.hip / coeff_net_train.hip, reading the parameters and
    writing their gradients in torch's own layouts.  ``lowres_input [B, N, N, 3]`` (no gradient) ->
    ``[B, sb, sb, gd, n_out, n_in]``."""
    _require_f32("lowres_input", lowres_input)
    _require_gpu("lowres_input", lowres_input)
    return _CoefficientsTrain.apply(lowres_input, hyper, n_out, n_in, n_splat, *params)


def resize_bilinear(input: torch.Tensor, height: int, width: int) -> torch.Tensor:  # noqa: A002
    """NHWC ``tf.image.resize_images(input, (height, width), BILINEAR, align_corners=True)`` --
    the resize that builds HDRNetGaussianPyrNN's multi-scale input (hdrnet/models.py:253-266).
    No autograd."""
    _require_f32("input", input)
    if input.dim() != 4:
        raise ValueError(f"input should be 4D (batch, height, width, channels), got {tuple(input.shape)}")
    _require_gpu("input", input)
    inp = input.detach().contiguous()
    B, Hin, Win, C = inp.shape
    out = torch.empty((B, int(height), int(width), C), dtype=torch.float32, device=inp.device)
    lib = _lib.load()
    with torch.cuda.device(inp.device):
        rc = lib.hdrnet_resize_bilinear_f32(inp.data_ptr(), out.data_ptr(), B, Hin, Win, int(height),
                                            int(width), C, _stream(inp.device))
    _lib.check(rc, "ResizeBilinear")
    return out


class _UpsampleAdd(torch.autograd.Function):
    """``resize(coarse -> fine's size, bilinear, align_corners) + fine`` in one pass; backward: d fine = the incoming gradient
    itself, d coarse = the resize's transpose as a fixed-order gather (csrc/resize_bilinear.hip)."""

    @staticmethod
    def forward(ctx, coarse, fine):
        c, f = coarse.detach().contiguous(), fine.detach().contiguous()
        B, IH, IW, C = c.shape
        OH, OW = f.shape[1], f.shape[2]
        out = torch.empty_like(f)
        with torch.cuda.device(f.device):
            rc = _lib.load().hdrnet_resize_add_f32(c.data_ptr(), f.data_ptr(), out.data_ptr(), B, IH, IW, OH, OW, C,
                                                   _stream(f.device))
        if rc != 0:
            raise RuntimeError(f"hdrnet_resize_add_f32 failed (rc={rc})")
        ctx.dims = (B, IH, IW, OH, OW, C)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        B, IH, IW, OH, OW, C = ctx.dims
        g = grad.contiguous()
        dcoarse = None
        if ctx.needs_input_grad[0]:
            dcoarse = torch.empty((B, IH, IW, C), dtype=torch.float32, device=g.device)
            with torch.cuda.device(g.device):
                rc = _lib.load().hdrnet_resize_bilinear_grad_f32(g.data_ptr(), dcoarse.data_ptr(), B, IH, IW, OH, OW, C,
                                                                 _stream(g.device))
            if rc != 0:
                raise RuntimeError(f"hdrnet_resize_bilinear_grad_f32 failed (rc={rc})")
        return dcoarse, (g if ctx.needs_input_grad[1] else None)


def upsample_add(coarse: torch.Tensor, fine: torch.Tensor) -> torch.Tensor:
    """``tf.image.resize_images(coarse, fine's size, BILINEAR, align_corners=True) + fine`` -- the up-add of
    HDRNetGaussianPyrNN's output pyramid (hdrnet/models.py:283-287) -- NHWC fp32 on the GPU, differentiable in both."""
    for name, t in (("coarse", coarse), ("fine", fine)):
        _require_f32(name, t)
        _require_gpu(name, t)
    if coarse.dim() != 4 or fine.dim() != 4 or coarse.shape[0] != fine.shape[0] or coarse.shape[3] != fine.shape[3]:
        raise ValueError(f"upsample_add: [B, h, w, C] and [B, H, W, C] expected, got {tuple(coarse.shape)}, {tuple(fine.shape)}")
    return _UpsampleAdd.apply(coarse, fine)


def bilateral_slice_apply_upadd(grid: torch.Tensor, input: torch.Tensor, coarse: torch.Tensor,  # noqa: A002
                                guide: Optional[torch.Tensor] = None,
                                guide_conv1: Optional[torch.Tensor] = None,
                                guide_conv2: Optional[torch.Tensor] = None,
                                has_offset: bool = True, fast_sigmoid: bool = False,
                                prescaled: bool = False) -> torch.Tensor:
    """One level of ``HDRNetGaussianPyrNN._output`` (hdrnet/models.py:277-289) in one pass:
    ``bilateral_slice_apply(grid, guide, input) + resize_bilinear(coarse -> H x W, align_corners)``.
    Give either a ``guide`` map or the folded guide network (``guide_conv1``, ``guide_conv2``), which
    is then evaluated in registers (``fast_sigmoid``, ``prescaled``: as ``bilateral_slice_apply_nnguide``).  Inference
    only (no autograd)."""
    if (guide is None) == (guide_conv1 is None):
        raise ValueError("give either guide or (guide_conv1, guide_conv2)")
    if input.dim() != 4:
        raise ValueError(f"Input image should be 4D (batch_size, height, width, input_channels), got {tuple(input.shape)}")
    if guide is None:
        B, H, W, GH, GW, GD, Cin, Cout, n = _check_nnguide(grid, input, guide_conv1, guide_conv2, has_offset)
    else:
        B, H, W, GH, GW, GD, Cin, Cout = _check_apply(grid, guide, input, has_offset)
        n = 0
    _require_f32("coarse", coarse)
    _require_gpu("coarse", coarse)
    if coarse.dim() != 4 or coarse.shape[0] != B or coarse.shape[3] != Cout:
        raise ValueError(f"coarse should be [B, Hc, Wc, Cout] = [{B}, *, *, {Cout}], got {tuple(coarse.shape)}")
    grid, inp, coarse = grid.detach().contiguous(), input.detach().contiguous(), coarse.detach().contiguous()
    gd = None if guide is None else guide.detach().contiguous()
    c1 = None if guide_conv1 is None else guide_conv1.detach().contiguous()
    c2 = None if guide_conv2 is None else guide_conv2.detach().contiguous()
    dev = inp.device
    out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.hdrnet_bilateral_slice_apply_upadd_f32_ex(
            grid.data_ptr(), _ptr(gd), inp.data_ptr(), coarse.data_ptr(), coarse.shape[1], coarse.shape[2],
            out.data_ptr(), B, H, W, GH, GW, GD, Cin, Cout, int(bool(has_offset)), _ptr(c1), _ptr(c2), n,
            _guide_flags(fast_sigmoid, prescaled and guide is None), _stream(dev))
    _lib.check(rc, "BilateralSliceApplyUpAdd")
    return out


_DTYPE_CODE = {torch.float32: 0, torch.uint8: 1, torch.uint16: 2}


def _check_io(grid, input, has_offset):  # noqa: A002
    """Rank / batch / channel / dtype rules shared by the wire-format entry points (the OP_REQUIRES of
    bilateral_slice_apply_op.cc:147-193 on an input that may be uint8 / uint16).  Returns
    (B, H, W, GH, GW, GD, Cin, Cout)."""
    if not isinstance(input, torch.Tensor) or input.dim() != 4:
        raise ValueError("Input image should be 4D (batch_size, height, width, input_channels)")
    if input.dtype not in _DTYPE_CODE:
        raise TypeError(f"input must be float32, uint8 or uint16, got {input.dtype}")
    _require_f32("grid", grid)
    if grid.dim() != 5:
        raise ValueError(f"Input grid should be 5D, got {tuple(grid.shape)}")
    B, H, W, Cin = input.shape
    if grid.shape[0] != B:
        raise ValueError("Batch sizes should match.")
    GH, GW, GD, C = grid.shape[1:]
    Cj = Cin + (1 if has_offset else 0)
    if Cj <= 0 or C % Cj:
        raise ValueError(
            "Slicing with affine offset, grid should have output_channels * (input_channels + 1) channels."
            if has_offset else
            "Slicing without affine offset, grid should have output_channels * input_channels channels.")
    return B, H, W, GH, GW, GD, Cin, C // Cj


def curves_guide_prepare(shifts: torch.Tensor, slopes: torch.Tensor) -> Optional[torch.Tensor]:
    """The curves guide's lookup tables, PREPARED once per parameter set (``hdrnet_curves_guide_prepare_f32``; Cin = 3,
    at most 16 knots per channel): each channel's knot range cut into 64 uniform cells, per cell the knot inside it, the
    curve's float64-summed value there and the slopes on either side.  Passed to ``bilateral_slice_apply_curves`` /
    ``bilateral_slice_apply_io`` as ``prepared`` / ``curves_prepared``, a pixel finds its cell by arithmetic and reads ONE
    table entry, where the plain call has every workgroup sort the knots and every pixel walk a search tree.  Same guide to
    1e-6.  A SET-UP call: it waits for the stream (one word comes back -- whether the cells separate the knots) and must not
    run inside a stream capture.  Returns an opaque float32 tensor, or ``None`` when two knots of a channel share a cell
    (closer than 1/63 of the channel's knot range): call the ops without ``prepared`` then.  ``shifts`` / ``slopes``:
    ``[npts, 3]`` (hdrnet/bin/freeze_graph.py:107-127)."""
    _require_f32("shifts", shifts)
    _require_f32("slopes", slopes)
    _require_gpu("shifts", shifts)
    _require_gpu("slopes", slopes)
    if shifts.dim() != 2 or shifts.shape[1] != 3 or tuple(slopes.shape) != tuple(shifts.shape) or not 0 < shifts.shape[0] <= 16:
        raise ValueError(f"curves_guide_prepare: shifts / slopes [npts <= 16, 3] expected, got {tuple(shifts.shape)}, "
                         f"{tuple(slopes.shape)}")
    sh, sl = shifts.detach().contiguous(), slopes.detach().contiguous()
    dev = sh.device
    lib = _lib.load()
    nbytes = lib.hdrnet_curves_guide_prepared_bytes(3)
    out = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
    usable = ctypes.c_int(0)
    with torch.cuda.device(dev):
        rc = lib.hdrnet_curves_guide_prepare_f32(sh.data_ptr(), sl.data_ptr(), sh.shape[0], 3, out.data_ptr(), nbytes,
                                                 ctypes.byref(usable), _stream(dev))
    _lib.check(rc, "CurvesGuidePrepare")
    return out if usable.value else None


def _apply_io_curves(grid, input, curves, input_white_level, out_dtype, has_offset, return_guide,  # noqa: A002
                     prepared=None):
    if len(curves) != 4:
        raise ValueError("guide_curves should be (ccm, shifts, slopes, mix)")
    ccm, shifts, slopes, mix = curves
    B, H, W, GH, GW, GD, Cin, Cout = _check_io(grid, input, has_offset)
    npts = shifts.shape[0] if shifts.dim() == 2 else -1
    for nm, t, shape in (("ccm", ccm, (Cin, Cin + 1)), ("shifts", shifts, (npts, Cin)),
                         ("slopes", slopes, (npts, Cin)), ("mix", mix, (Cin + 1,))):
        _require_f32(f"guide_curves.{nm}", t)
        if tuple(t.shape) != shape or npts <= 0:
            raise ValueError(f"guide_curves.{nm} should be {list(shape)}, got {list(t.shape)}")
    if input_white_level is None:
        input_white_level = {torch.float32: 1.0, torch.uint8: 255.0, torch.uint16: 65535.0}[input.dtype]
    for nm, t in (("grid", grid), ("input", input), ("guide_curves.ccm", ccm), ("guide_curves.shifts", shifts),
                  ("guide_curves.slopes", slopes), ("guide_curves.mix", mix)):
        _require_gpu(nm, t)
    grid, inp = grid.detach().contiguous(), input.detach().contiguous()
    ccm, shifts, slopes, mix = (t.detach().contiguous() for t in (ccm, shifts, slopes, mix))
    dev = inp.device
    lib = _lib.load()
    if prepared is not None:
        _require_f32("prepared", prepared)
        _require_gpu("prepared", prepared)
        if not prepared.is_contiguous() or prepared.numel() * 4 < lib.hdrnet_curves_guide_prepared_bytes(Cin):
            raise ValueError("prepared should be the tensor curves_guide_prepare returned")
    out = torch.empty((B, H, W, Cout), dtype=out_dtype, device=dev)
    gout = torch.empty((B, H, W), dtype=torch.float32, device=dev) if return_guide else None
    with torch.cuda.device(dev):
        rc = lib.hdrnet_bilateral_slice_apply_io_curves_prepared(
            grid.data_ptr(), inp.data_ptr(), out.data_ptr(), B, H, W, GH, GW, GD, Cin, Cout,
            int(bool(has_offset)), _DTYPE_CODE[input.dtype], float(input_white_level), _DTYPE_CODE[out_dtype],
            ccm.data_ptr(), shifts.data_ptr(), slopes.data_ptr(), mix.data_ptr(), npts, _ptr(prepared), _ptr(gout),
            _stream(dev))
    _lib.check(rc, "BilateralSliceApplyIOCurves")
    return (out, gout) if return_guide else out


def bilateral_slice_apply_io(grid: torch.Tensor, input: torch.Tensor,  # noqa: A002
                             guide: Optional[torch.Tensor] = None,
                             guide_conv1: Optional[torch.Tensor] = None,
                             guide_conv2: Optional[torch.Tensor] = None,
                             input_white_level: Optional[float] = None,
                             out_dtype: torch.dtype = torch.float32,
                             has_offset: bool = True,
                             guide_curves: Optional[Tuple[torch.Tensor, ...]] = None,
                             return_guide: bool = False, fast_sigmoid: bool = False, prescaled: bool = False,
                             curves_prepared: Optional[torch.Tensor] = None):
    """Inference forward with the product's wire formats fused in: ``input`` may be uint8 / uint16
    (``value / input_white_level``: 255, 65535, or 32767 for HDR+ -- hdrnet/data_pipeline.py:202-232,
    :267-274) and the output may be uint8 ``= (uint8)(255 * clip(out, 0, 1))`` (hdrnet/bin/run.py:95).
    The guide is ONE of: a ``guide`` map; the folded point-wise guide network (``guide_conv1``,
    ``guide_conv2``); or ``guide_curves = (ccm [Cin, Cin+1], shifts [npts, Cin], slopes [npts, Cin],
    mix [Cin+1])``, the standard model's curves guide (hdrnet/models.py:145-190) in the layout
    hdrnet/bin/freeze_graph.py:107-127 exports -- then evaluated in registers, as the reference's
    standard GL shader does (benchmark/assets/std.frag:36-45).  ``return_guide`` (guide network or curves) also
    returns the guide map the kernel computed.  ``fast_sigmoid`` (guide network): the hardware exp / reciprocal
    sigmoid, <= 2 ulp of the guide away from the default (tf.nn.sigmoid's form) and ~10 % faster -- an explicit
    choice of the caller (HDRNET_GUIDE_SIGMOID_FAST), never implied by another argument.  ``prescaled`` (guide network):
    ``guide_conv1`` / ``guide_conv2`` are ``guide_nn_prescale``'s arrays (HDRNET_GUIDE_RELU_PRESCALED).
    ``curves_prepared`` (``guide_curves``): the tables of ``curves_guide_prepare``.  No autograd."""
    if input.dim() != 4:
        raise ValueError(f"Input image should be 4D (batch_size, height, width, input_channels), got {tuple(input.shape)}")
    if input.dtype not in _DTYPE_CODE:
        raise TypeError(f"input must be float32, uint8 or uint16, got {input.dtype}")
    if out_dtype not in (torch.float32, torch.uint8):
        raise TypeError(f"out_dtype must be float32 or uint8, got {out_dtype}")
    if guide_curves is not None:
        if guide is not None or guide_conv1 is not None or guide_conv2 is not None:
            raise ValueError("give exactly one of guide, (guide_conv1, guide_conv2), guide_curves")
        return _apply_io_curves(grid, input, guide_curves, input_white_level, out_dtype, has_offset, return_guide,
                                curves_prepared)
    if return_guide and guide is not None:
        raise ValueError("return_guide needs a guide computed by the kernel (guide network or guide_curves)")
    if (guide is None) == (guide_conv1 is None or guide_conv2 is None):
        raise ValueError("give either a guide map or both guide_conv1 and guide_conv2")
    if input_white_level is None:
        input_white_level = {torch.float32: 1.0, torch.uint8: 255.0, torch.uint16: 65535.0}[input.dtype]
    B, H, W, GH, GW, GD, Cin, Cout = _check_io(grid, input, has_offset)
    n = 0
    if guide is not None:
        _require_f32("guide", guide)
        if tuple(guide.shape) != (B, H, W):
            raise ValueError("Input and guide size should match.")
        _require_gpu("guide", guide)
        guide = guide.detach().contiguous()
    else:
        _require_f32("guide_conv1", guide_conv1)
        _require_f32("guide_conv2", guide_conv2)
        n = guide_conv1.shape[0]
        if guide_conv1.dim() != 2 or guide_conv1.shape[1] != Cin + 1 or tuple(guide_conv2.shape) != (n + 1,):
            raise ValueError("guide_conv1 should be [n, Cin + 1] and guide_conv2 [n + 1]")
        guide_conv1, guide_conv2 = guide_conv1.detach().contiguous(), guide_conv2.detach().contiguous()
    _require_gpu("grid", grid)
    _require_gpu("input", input)
    grid, inp = grid.detach().contiguous(), input.detach().contiguous()
    dev = inp.device
    out = torch.empty((B, H, W, Cout), dtype=out_dtype, device=dev)
    gout = torch.empty((B, H, W), dtype=torch.float32, device=dev) if return_guide else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.hdrnet_bilateral_slice_apply_io_ex(
            grid.data_ptr(), _ptr(guide), inp.data_ptr(), out.data_ptr(), B, H, W, GH, GW, GD, Cin, Cout,
            int(bool(has_offset)), _DTYPE_CODE[input.dtype], float(input_white_level), _DTYPE_CODE[out_dtype],
            _ptr(guide_conv1) if guide is None else None, _ptr(guide_conv2) if guide is None else None,
            n, _ptr(gout), _guide_flags(fast_sigmoid, prescaled and guide is None), _stream(dev))
    _lib.check(rc, "BilateralSliceApplyIO")
    return (out, gout) if return_guide else out
