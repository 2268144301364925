"""The three HDRNet model graphs of the reference (``hdrnet/models.py``) as torch modules.

SURVEY.md section 8f row 1 -- the direct CALLER of the hot path.  The low-resolution coefficient
network and the point-wise guide are ordinary PyTorch-ROCm ops (stock convs / matmuls on the
same stream); the full-resolution work is ``layers.bilateral_slice_apply`` = the HIP kernels.
Nothing here is a kernel; it exists so that BASELINE.json's configs #3 (full inference at 4K)
and #4 (training step) can be run end to end with the reference's graph.

Conventions follow the reference so that weights could be ported one to one:

* images are NHWC float32 (``lowres_input [B, 256, 256, 3]``, ``fullres_input [B, H, W, 3]``);
* convs use TensorFlow ``padding='SAME'`` (asymmetric for stride 2: the extra row / column goes
  at the END -- ``tf_same_pad``), ``tf.contrib.layers.batch_norm`` defaults (no scale,
  ``center=True``, eps 1e-3, decay 0.999; ``hdrnet/layers.py:40-58``);
* the prediction conv's channel ``(j * n_out + i) * gd + z`` is unrolled to
  ``coeffs[b, gy, gx, z, i, j]`` exactly as ``models.py:134-138`` does;
* fully connected layers see the ``(h, w, c)``-ordered flattening of ``models.py:92-93``.

Parity status of this file: UNPINNED against TensorFlow (no TF1 here to run the reference graph);
``tests/test_models.py`` pins the pieces that have closed forms (SAME padding, unroll order,
guide formulas, the hot-path composition against the CPU oracle).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _state, layers

__all__ = ["HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN", "default_params", "tf_same_pad"]


def default_params(**overrides) -> Dict:
    """Model hyper-parameters with the defaults of ``hdrnet/bin/train.py:227-236``."""
    p = dict(batch_norm=False, net_input_size=256, luma_bins=8, spatial_bin=16,
             channel_multiplier=1, guide_complexity=16)
    p.update(overrides)
    return p


def tf_same_pad(x: torch.Tensor, kernel: int, stride: int) -> torch.Tensor:
    """Pad an NCHW tensor the way TF's padding='SAME' does (total = max((ceil(n/s)-1)*s + k - n, 0),
    floor(total/2) before, the rest after)."""
    def pads(n):
        total = max((math.ceil(n / stride) - 1) * stride + kernel - n, 0)
        return total // 2, total - total // 2
    (t, b), (l, r) = pads(x.shape[2]), pads(x.shape[3])
    return F.pad(x, (l, r, t, b)) if (t or b or l or r) else x


def _param_key(tensors) -> tuple:
    """Identity + in-place version of every tensor + the generation of the raw writers attached to THAT tensor
    (``_state``): changes when a parameter is stepped, loaded or replaced -- also by the writers that bypass the version
    counters (``optim.FlatAdam``'s raw-pointer update, a replayed training hipGraph) -- and only then: another model's
    optimizer step does not touch this key."""
    return tuple((t.data_ptr(), t._version, _state.generation_of(t)) for t in tensors)


def _cacheable() -> bool:
    """Derived tensors made while a hipGraph is being captured belong to the graph's pool: never keep those."""
    return not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


class _BN(nn.Module):
    """tf.contrib.layers.batch_norm(center=True, scale=False): beta only, eps 1e-3, decay 0.999."""

    def __init__(self, ch: int, dims: int):
        super().__init__()
        cls = nn.BatchNorm2d if dims == 2 else nn.BatchNorm1d
        self.bn = cls(ch, eps=1e-3, momentum=1e-3, affine=True)
        nn.init.ones_(self.bn.weight)
        nn.init.zeros_(self.bn.bias)
        self.bn.weight.requires_grad_(False)

    def forward(self, x):
        # nn.BatchNorm's own forward also increments num_batches_tracked (a launch per layer and step; the counter only
        # matters for momentum=None, and TensorFlow's batch_norm has none): the functional form with the same arguments
        m = self.bn
        return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, self.training, m.momentum, m.eps)


class _Conv(nn.Module):
    """``layers.conv`` (hdrnet/layers.py:25-59): SAME conv, optional BN (then no conv bias), ReLU."""

    def __init__(self, cin, cout, k, stride=1, use_bias=True, batch_norm=False, activation=F.relu):
        super().__init__()
        self.k, self.stride, self.activation = k, stride, activation
        # stride 1, odd kernel: SAME padding is symmetric -- the convolution's own padding, no pad copy (a launch each way)
        self.own_pad = stride == 1 and k % 2 == 1
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2 if self.own_pad else 0,
                              bias=use_bias and not batch_norm)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_in", nonlinearity="relu")  # variance_scaling(2, FAN_IN)
        if self.conv.bias is not None:
            nn.init.zeros_(self.conv.bias)
        self.bn = _BN(cout, 2) if batch_norm else None

    def forward(self, x):
        x = self.conv(x if self.own_pad else tf_same_pad(x, self.k, self.stride))
        if self.bn is not None:
            x = self.bn(x)
        return self.activation(x) if self.activation is not None else x


class _FC(nn.Module):
    """``layers.fc`` (hdrnet/layers.py:62-93)."""

    def __init__(self, cin, cout, use_bias=True, batch_norm=False, activation=F.relu):
        super().__init__()
        self.activation = activation
        self.fc = nn.Linear(cin, cout, bias=use_bias and not batch_norm)
        nn.init.kaiming_normal_(self.fc.weight, mode="fan_in", nonlinearity="relu")
        if self.fc.bias is not None:
            nn.init.zeros_(self.fc.bias)
        self.bn = _BN(cout, 1) if batch_norm else None

    def forward(self, x):
        x = self.fc(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.activation(x) if self.activation is not None else x


class _Coefficients(nn.Module):
    """``HDRNetCurves._coefficients`` (hdrnet/models.py:62-142): splat -> global / local -> fusion ->
    1x1 prediction -> unroll to ``[B, GH, GW, gd, n_out, n_in]``."""

    def __init__(self, params: Dict, n_out: int, n_in: int):
        super().__init__()
        gd, cm, sb = params["luma_bins"], params["channel_multiplier"], params["spatial_bin"]
        bn = params["batch_norm"]
        self.gd, self.n_out, self.n_in = gd, n_out, n_in
        self.hyper = dict(net_input_size=params["net_input_size"], spatial_bin=sb, luma_bins=gd, channel_multiplier=cm)
        self.n_levels = 1  # HDRNetGaussianPyrNN: 3 (the native kernels then write every level's grid contiguous)
        self._exported = None  # (key, hdrnet_ops.CoefficientWeights)
        n_ds = int(math.log2(params["net_input_size"] / sb))
        splat, cin = [], 3
        for i in range(n_ds):
            splat.append(_Conv(cin, cm * (2 ** i) * gd, 3, stride=2, batch_norm=bn if i > 0 else False))
            cin = cm * (2 ** i) * gd
        self.splat = nn.Sequential(*splat)
        self.global_conv = nn.Sequential(_Conv(cin, 8 * cm * gd, 3, stride=2, batch_norm=bn),
                                         _Conv(8 * cm * gd, 8 * cm * gd, 3, stride=2, batch_norm=bn))
        side = sb // 4  # grid is sb x sb after the splat; two stride-2 convs
        self.fc1 = _FC(side * side * 8 * cm * gd, 32 * cm * gd, batch_norm=bn)
        self.fc2 = _FC(32 * cm * gd, 16 * cm * gd, batch_norm=bn)
        self.fc3 = _FC(16 * cm * gd, 8 * cm * gd, activation=None)
        self.local1 = _Conv(cin, 8 * cm * gd, 3, batch_norm=bn)
        self.local2 = _Conv(8 * cm * gd, 8 * cm * gd, 3, use_bias=False, activation=None)
        self.pred = _Conv(8 * cm * gd, gd * n_out * n_in, 1, activation=None)
        # MIOpen picks NHWC implicit-GEMM kernels for these shapes and the activations already are
        # channels-last (an NHWC tensor viewed as NCHW): keep the weights in that format too, or every
        # call converts them (8 extra launches of 67 per inference)
        self.to(memory_format=torch.channels_last)

    # Inference on the HIP kernels of csrc/coeff_net.hip (10 launches instead of ~67 stock-op launches); training
    # and anything that needs autograd stays on the torch ops below.  ``native = False`` forces the torch ops.
    native = True

    @staticmethod
    def _fold(weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional["_BN"]):
        """Batch norm (running statistics) folded into weight / bias, as hdrnet/bin/freeze_graph.py:170-184 folds
        the guide's: ``w * gamma / sqrt(var + eps)`` per output channel, ``beta - mean * that``."""
        if bn is None:
            return weight, bias
        m = bn.bn
        inv = torch.rsqrt(m.running_var + m.eps) * m.weight
        w = weight * inv.reshape(-1, *([1] * (weight.dim() - 1)))
        b = m.bias - m.running_mean * inv
        if bias is not None:
            b = b + bias * inv
        return w, b

    def exported(self):
        """The parameters in the layout of ``hdrnet_coefficients_f32`` (include/hdrnet_amd.h): convolutions
        ``[Cout][kh][kw][Cin]``, fully connected ``[in][out]``, batch norm folded (eval-mode statistics).  Cached;
        rebuilt when any parameter or buffer was modified in place or replaced."""
        from . import hdrnet_ops
        key = _param_key(list(self.parameters()) + list(self.buffers())) + (self.n_levels,)
        if self._exported is not None and self._exported[0] == key:
            return self._exported[1]
        with torch.no_grad():
            def own(t):  # a SNAPSHOT: never an alias of the parameter (a later in-place update must not half-reach
                return t.float().clone(memory_format=torch.contiguous_format)  # a captured graph)

            def conv(layer: _Conv):
                w, b = self._fold(layer.conv.weight, layer.conv.bias, layer.bn)
                return own(w.permute(0, 2, 3, 1)), (None if b is None else own(b))

            def fc(layer: _FC):
                w, b = self._fold(layer.fc.weight, layer.fc.bias, layer.bn)
                return own(w.t()), own(b)

            weights = hdrnet_ops.CoefficientWeights(
                self.hyper, self.n_out, self.n_in, self.n_levels,
                splat=[conv(layer) for layer in self.splat],
                global_conv=[conv(layer) for layer in self.global_conv],
                fc=[fc(self.fc1), fc(self.fc2), fc(self.fc3)],
                local=[conv(self.local1), conv(self.local2)],
                pred=conv(self.pred))
        if _cacheable():
            self._exported = (key, weights)
        return weights

    def _use_native(self, lowres_nhwc: torch.Tensor) -> bool:
        if not (self.native and lowres_nhwc.is_cuda and lowres_nhwc.dtype == torch.float32 and not self.training):
            return False
        if torch.is_grad_enabled() and (lowres_nhwc.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        N = self.hyper["net_input_size"]
        if lowres_nhwc.dim() != 4 or tuple(lowres_nhwc.shape[1:]) != (N, N, 3):
            return False
        return self.exported().supported(max(int(lowres_nhwc.shape[0]), 1))

    # Training (and any differentiable evaluation) of the network WITHOUT batch norm -- the reference's own training
    # configuration for the guide-network model (scripts/ll/train_nn_guide.sh) -- on the HIP kernels as well: forward
    # + backward in ~35 launches instead of ~130 stock-op launches.  With batch norm, or when the input itself needs a
    # gradient, the torch ops below run.  ``native_training = False`` forces them.
    native_training = True

    def _train_params(self):
        convs = list(self.splat) + list(self.global_conv)
        ps = []
        for layer in convs:
            ps += [layer.conv.weight, layer.conv.bias]
        for layer in (self.fc1, self.fc2, self.fc3):
            ps += [layer.fc.weight, layer.fc.bias]
        ps += [self.local1.conv.weight, self.local1.conv.bias, self.local2.conv.weight,
               self.pred.conv.weight, self.pred.conv.bias]
        return ps

    def _use_native_training(self, lowres_nhwc: torch.Tensor) -> bool:
        if not (self.native and self.native_training and lowres_nhwc.is_cuda and lowres_nhwc.dtype == torch.float32):
            return False
        if not torch.is_grad_enabled() or lowres_nhwc.requires_grad:
            return False  # (the pyramid model's 9 x 4 coefficients come out in the reference's order: n_levels plays no role here)
        if any(m.bn is not None for m in self.modules() if isinstance(m, (_Conv, _FC))):
            return False
        N = self.hyper["net_input_size"]
        if lowres_nhwc.dim() != 4 or tuple(lowres_nhwc.shape[1:]) != (N, N, 3):
            return False
        ps = self._train_params()
        if any(p is None for p in ps) or not any(p.requires_grad for p in ps):
            return False
        from . import hdrnet_ops
        return hdrnet_ops.coefficients_train_supported(self.hyper, self.n_out, self.n_in, ps, len(self.splat),
                                                       int(lowres_nhwc.shape[0]))

    def levels(self, lowres_nhwc: torch.Tensor) -> List[torch.Tensor]:
        """Per pyramid level the 5-D grid ``[B, GH, GW, gd, (n_out / n_levels) * n_in]`` of
        ``coeffs[:, :, :, :, l*k:(l+1)*k, :]`` (hdrnet/models.py:279), each contiguous."""
        L = self.n_levels
        if self._use_native(lowres_nhwc):
            from . import hdrnet_ops
            out = hdrnet_ops.coefficients(lowres_nhwc, self.exported())
            out = out if L > 1 else out[None]
            return [out[l].reshape(*out.shape[1:5], -1) for l in range(L)]
        coeffs = self.forward(lowres_nhwc)
        gs, k = coeffs.shape, self.n_out // L
        return [coeffs[:, :, :, :, l * k:(l + 1) * k, :].reshape(gs[0], gs[1], gs[2], gs[3], k * gs[5]) for l in range(L)]

    def forward(self, lowres_nhwc: torch.Tensor) -> torch.Tensor:
        if self._use_native_training(lowres_nhwc):
            from . import hdrnet_ops
            return hdrnet_ops.coefficients_train(lowres_nhwc, self.hyper, self.n_out, self.n_in, self._train_params(),
                                                 len(self.splat))
        if self._use_native(lowres_nhwc):
            from . import hdrnet_ops
            out = hdrnet_ops.coefficients(lowres_nhwc, self.exported())
            if self.n_levels > 1:  # level-major -> the reference's [B, GH, GW, gd, n_out, n_in]
                L, B, GH, GW, gd, k, n_in = out.shape
                out = out.permute(1, 2, 3, 4, 0, 5, 6).reshape(B, GH, GW, gd, L * k, n_in)
            return out
        x = lowres_nhwc.permute(0, 3, 1, 2)  # the convs run NCHW; the tensor is 256 x 256
        splat = self.splat(x)
        g = self.global_conv(splat)
        g = g.permute(0, 2, 3, 1).reshape(g.shape[0], -1)  # (h, w, c) flattening, models.py:92-93
        g = self.fc3(self.fc2(self.fc1(g)))
        loc = self.local2(self.local1(splat))
        fusion = F.relu(loc + g[:, :, None, None])
        pred = self.pred(fusion)  # [B, (j*n_out + i)*gd + z, GH, GW]
        B, _, GH, GW = pred.shape
        pred = pred.reshape(B, self.n_in, self.n_out, self.gd, GH, GW)
        return pred.permute(0, 4, 5, 3, 2, 1).contiguous()  # [B, GH, GW, gd, n_out, n_in]


class _CurvesGuide(nn.Module):
    """``HDRNetCurves._guide`` (models.py:145-190): 3x3 colour matrix, 16-knot per-channel curves,
    1x1 channel mixing, clip to [0, 1]."""

    def __init__(self, nchans: int = 3, npts: int = 16):
        super().__init__()
        self.ccm = nn.Parameter(torch.eye(nchans) + torch.randn(1) * 1e-4)
        self.ccm_bias = nn.Parameter(torch.zeros(nchans))
        self.shifts = nn.Parameter(torch.linspace(0, 1, npts + 1)[:-1].repeat(nchans, 1))  # [c, k]
        slopes = torch.zeros(nchans, npts)
        slopes[:, 0] = 1.0
        self.slopes = nn.Parameter(slopes)
        self.mix_w = nn.Parameter(torch.full((nchans,), 1.0 / nchans))
        self.mix_b = nn.Parameter(torch.zeros(()))

    def forward(self, im: torch.Tensor) -> torch.Tensor:
        # Written with broadcasts + reductions rather than `@`: on a [8.3 M, 3] activation rocBLAS
        # turns `x @ w` / its backward into gemv calls that take 70 ms of a 100 ms training step.
        g = (im.unsqueeze(-1) * self.ccm).sum(-2) + self.ccm_bias
        g = (self.slopes * F.relu(g.unsqueeze(-1) - self.shifts)).sum(-1)
        g = (g * self.mix_w).sum(-1) + self.mix_b
        return g.clamp(0.0, 1.0)

    def exported(self):
        """The parameters in the layout hdrnet/bin/freeze_graph.py:107-127 writes (and the GL
        renderer loads, benchmark/src/renderer.cc:197-225): ccm [3, 4] = (ccm; bias)^T, shifts /
        slopes [npts, 3], mix [4] = (weights, bias)."""
        key = _param_key(self.parameters())
        cached = getattr(self, "_exported_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        with torch.no_grad():
            out = tuple(t.detach() for t in self.exported_differentiable())
        if _cacheable():
            self._exported_cache = (key, out)  # inference replays these four arrays: no re-packing launches per frame
        return out

    def prepared(self):
        """The curves' lookup tables prepared once per parameter state (``hdrnet_ops.curves_guide_prepare``: uniform cells
        instead of a per-workgroup sort + per-pixel tree search), or None -- off the GPU, for other shapes, when two knots of
        a channel share a cell, or while a stream capture is running with nothing cached (the set-up call synchronises)."""
        _, shifts, slopes, _ = self.exported()
        if not (shifts.is_cuda and shifts.shape[1] == 3 and shifts.shape[0] <= 16):
            return None
        key = _param_key(self.parameters())
        cached = getattr(self, "_prepared_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if not _cacheable():
            return None
        from . import hdrnet_ops
        out = hdrnet_ops.curves_guide_prepare(shifts, slopes)
        self._prepared_cache = (key, out)
        return out

    def exported_differentiable(self):
        """Same arrays, attached to the autograd graph (training through the fused op)."""
        ccm = torch.cat([self.ccm, self.ccm_bias[None, :]], dim=0).t().contiguous()
        mix = torch.cat([self.mix_w, self.mix_b.reshape(1)]).contiguous()
        return ccm, self.shifts.t().contiguous(), self.slopes.t().contiguous(), mix


class _PointwiseNNGuide(nn.Module):
    """``HDRNetPointwiseNNGuide._guide`` (models.py:203-210): 1x1 conv 3 -> n (+BN, ReLU), 1x1 conv
    n -> 1, sigmoid.  Point-wise, so it is written on the channel axis of the NHWC image."""

    def __init__(self, n_feats: int, nchans: int = 3):
        super().__init__()
        self.w1 = nn.Parameter(torch.empty(nchans, n_feats))
        nn.init.kaiming_normal_(self.w1.t(), mode="fan_in", nonlinearity="relu")
        self.bn = nn.BatchNorm1d(n_feats, eps=1e-3, momentum=1e-3)
        nn.init.ones_(self.bn.weight)
        self.bn.weight.requires_grad_(False)
        self.w2 = nn.Parameter(torch.empty(n_feats))
        nn.init.normal_(self.w2, std=math.sqrt(2.0 / n_feats))
        self.b2 = nn.Parameter(torch.zeros(()))

    def forward(self, im: torch.Tensor) -> torch.Tensor:
        h = im @ self.w1  # [B, H, W, n]
        shape = h.shape
        h = self.bn(h.reshape(-1, shape[-1])).reshape(shape)
        # (relu(h) * w2).sum(-1), not `relu(h) @ w2`: the backward of the latter is a rocBLAS gemv on
        # an [8.3 M, 16] matrix that takes 75 ms (this is only the un-fused composition; the model
        # normally runs the fused kernels)
        return torch.sigmoid((F.relu(h) * self.w2).sum(-1) + self.b2)

    def folded(self, detach: bool = True):
        """Batch-norm (running statistics) folded into the first layer, in the reference's export
        layout (hdrnet/bin/freeze_graph.py:170-184): conv1 [n, Cin + 1], conv2 [n + 1]."""
        if detach:  # inference: the fold is ~8 tiny launches -- done once per parameter state, not once per frame
            key = _param_key(list(self.parameters()) + list(self.buffers()))
            cached = getattr(self, "_folded_cache", None)
            if cached is not None and cached[0] == key:
                return cached[1]
        inv = torch.rsqrt(self.bn.running_var + self.bn.eps) * self.bn.weight
        w = self.w1 * inv  # [Cin, n]
        b = self.bn.bias - self.bn.running_mean * inv
        conv1 = torch.cat([w.t(), b[:, None]], dim=1).contiguous()
        conv2 = torch.cat([self.w2, self.b2.reshape(1)]).contiguous()
        if not detach:
            return conv1, conv2
        out = (conv1.detach(), conv2.detach())
        if _cacheable():
            self._folded_cache = (key, out)
        return out

    # the range of |input| the prescaled form promises its bit-equality for (hdrnet_ops.guide_nn_prescale); the model's
    # inputs are in [0, 1], its wire formats' raw samples at most 65535
    prescale_x_max = 65536.0

    def inference_params(self, prescale: bool):
        """``(conv1, conv2, prescaled)`` for an inference forward: the folded arrays, in their PRESCALED form
        (``HDRNET_GUIDE_RELU_PRESCALED``: the same guide bit for bit, 80 fewer vector instructions per 256 pixels) when
        ``prescale`` and the network has three input channels on a GPU -- prepared once per parameter state."""
        conv1, conv2 = self.folded()
        if not (prescale and conv1.is_cuda and conv1.shape[1] == 4):
            return conv1, conv2, False
        key = _param_key(list(self.parameters()) + list(self.buffers()))
        cached = getattr(self, "_prescaled_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        from . import hdrnet_ops
        p1, p2 = hdrnet_ops.guide_nn_prescale(conv1, conv2, self.prescale_x_max)
        out = (p1, p2, True)
        if _cacheable():
            self._prescaled_cache = (key, out)
        return out

    def folded_batch(self, sums: torch.Tensor, moments: torch.Tensor, npx: int):
        """Training-mode fold: batch norm normalises the conv output h = x . w1 with ITS mean and
        biased variance over the batch (hdrnet/layers.py:40-58, is_training=True).  h is linear
        in x, so  mean_h = w1^T mean_x  and  var_h[k] = w1[:,k]^T Cov_x w1[:,k]  -- computed here
        from the input's first and second moments (``hdrnet_ops.input_moments``) in float64,
        differentiable in w1 / beta / w2 / b2 (the moments are constants: the full-resolution image
        is data).  Updates the running statistics exactly as ``nn.BatchNorm1d`` would."""
        if sums.is_cuda and self.w1.shape[0] in (1, 3):  # one HIP launch each way instead of ~33 + ~30 torch launches
            from . import hdrnet_ops
            return hdrnet_ops.guide_fold_batch(
                self.w1, self.bn.bias, self.w2, self.b2, self.bn.weight, sums, moments, npx, self.bn.eps,
                self.bn.momentum, self.bn.running_mean, self.bn.running_var, self.bn.num_batches_tracked)
        return self._folded_batch_torch(sums, moments, npx)

    def _folded_batch_torch(self, sums: torch.Tensor, moments: torch.Tensor, npx: int):
        """The same fold as differentiable torch math (the definition the kernel is tested against)."""
        n = float(npx)
        mean_x = sums.double() / n
        cov_x = moments.double() / n - torch.outer(mean_x, mean_x)  # biased
        w1 = self.w1.double()
        mean_h = mean_x @ w1  # [n]
        var_h = ((cov_x @ w1) * w1).sum(0).clamp_min(0.0)
        with torch.no_grad():
            m = self.bn.momentum
            self.bn.running_mean.mul_(1 - m).add_(m * mean_h.float())
            self.bn.running_var.mul_(1 - m).add_(m * (var_h * (n / max(n - 1.0, 1.0))).float())
            self.bn.num_batches_tracked += 1
        inv = torch.rsqrt(var_h + self.bn.eps) * self.bn.weight.double()
        w = w1 * inv
        b = self.bn.bias.double() - mean_h * inv
        conv1 = torch.cat([w.t(), b[:, None]], dim=1).float().contiguous()
        conv2 = torch.cat([self.w2, self.b2.reshape(1)]).contiguous()
        return conv1, conv2


class HDRNetCurves(nn.Module):
    """``hdrnet/models.py:23-196``.  ``forward(lowres_input, fullres_input)`` = ``inference``."""

    n_out, n_in = 3, 4

    def __init__(self, params: Optional[Dict] = None):
        super().__init__()
        self.params = default_params(**(params or {}))
        self.coefficients = _Coefficients(self.params, self.n_out, self.n_in)
        self.guide = self._make_guide()

    def _make_guide(self) -> nn.Module:
        return _CurvesGuide()

    fuse_guide = True
    # Inference through the fused guide-network kernels uses the hardware exp / reciprocal sigmoid (<= 2 ulp of the guide,
    # ~10 % faster, HDRNET_GUIDE_SIGMOID_FAST) -- the MODEL's explicit choice, passed to every such call; False = the
    # exact tf.nn.sigmoid form everywhere.  Training forwards always use the exact form.
    fast_sigmoid = True
    # ... and the guide network's PRESCALED parameters (HDRNET_GUIDE_RELU_PRESCALED): bit-identical guide for |input| <=
    # _PointwiseNNGuide.prescale_x_max, relu from the clamp modifier.  Inference only, like fast_sigmoid.
    prescale_guide = True
    # ... and the curves guide's lookup tables prepared once per parameter state (hdrnet_curves_guide_prepare_f32)
    prepare_curves = True

    # INPUT-RANGE CONTRACT of the inference defaults above.  The prescaled guide network equals the exported one bit for
    # bit only while |full-resolution input| <= _PointwiseNNGuide.prescale_x_max (65536: normalised images and every
    # uint8 / uint16 wire format); beyond it a hidden feature saturates at 2^e_k and the guide is silently wrong --
    # un-normalised HDR floats must switch it off.  fast_sigmoid moves the guide by <= 2 ulp, prepare_curves by <= 5e-7.
    # ``exact_inference()`` switches all three off at once: tf.nn.sigmoid's form, the exported parameter layout and
    # the sorted knot tables -- no range limit, the arithmetic of the training forward.
    def exact_inference(self, on: bool = True):
        """``on``: inference uses the exact guide arithmetic (no fast sigmoid, no prescaled parameters, no prepared curve
        tables) -- for inputs beyond ``prescale_x_max`` or bit-level comparisons with the training graph.  ``False``
        restores the class defaults.  Returns ``self``."""
        if on:
            self.fast_sigmoid = self.prescale_guide = self.prepare_curves = False
        else:
            for k in ("fast_sigmoid", "prescale_guide", "prepare_curves"):
                self.__dict__.pop(k, None)
        return self

    def forward(self, lowres_input: torch.Tensor, fullres_input: torch.Tensor) -> torch.Tensor:
        coeffs = self.coefficients(lowres_input)
        if (self.fuse_guide and isinstance(self.guide, _CurvesGuide) and fullres_input.is_cuda
                and fullres_input.shape[3] == 3 and fullres_input.shape[2] % 4 == 0):
            # curves guide evaluated in registers inside the slice-apply kernel, as the reference's
            # standard GL shader does (benchmark/assets/std.frag:32-53); with autograd on, the
            # backward is the slice-apply VJP + the curves guide's VJP kernel
            from . import hdrnet_ops
            gs = coeffs.shape
            differentiable = torch.is_grad_enabled() and (
                fullres_input.requires_grad or any(p.requires_grad for p in self.parameters()))
            arrays = self.guide.exported_differentiable() if differentiable else self.guide.exported()
            # inference: the curves' tables prepared once per parameter state (same guide to 1e-6, ~0.9 x the time)
            prepared = None if (differentiable or not self.prepare_curves) else self.guide.prepared()
            return hdrnet_ops.bilateral_slice_apply_curves(
                coeffs.reshape(gs[0], gs[1], gs[2], gs[3], gs[4] * gs[5]), fullres_input, *arrays, has_offset=True,
                prepared=prepared)
        guide = self.guide(fullres_input)
        # models.py:193-196 -- the one call site of the hot path
        return layers.bilateral_slice_apply(coeffs, guide, fullres_input, has_offset=True, name="slice")


class HDRNetPointwiseNNGuide(HDRNetCurves):
    """``hdrnet/models.py:199-210``.  The guide network is FUSED into the slice-apply kernel
    (SURVEY.md section 8f row 2): the 16-channel full-resolution intermediate is never
    materialised, in inference (the guide never touches HBM) or in training (the guide is written
    once for the backward; batch-norm statistics come from the input's moments; the network's VJP
    is one more pass).  ``fuse_guide = False`` composes the un-fused ops instead."""

    def _make_guide(self) -> nn.Module:
        return _PointwiseNNGuide(self.params["guide_complexity"])

    def forward(self, lowres_input: torch.Tensor, fullres_input: torch.Tensor) -> torch.Tensor:
        n_feats = self.params["guide_complexity"]
        fusable = (self.fuse_guide and fullres_input.is_cuda and fullres_input.shape[2] % 4 == 0
                   and fullres_input.shape[3] in (1, 3))
        differentiable = torch.is_grad_enabled() and (
            fullres_input.requires_grad or any(p.requires_grad for p in self.parameters()))
        if differentiable and n_feats not in (4, 8, 16):
            fusable = False  # no guide-network VJP kernel for this width: compose the ops
        if self.training and torch.is_grad_enabled() and fullres_input.requires_grad:
            # The fused training path takes the batch-norm statistics from the input's moments as
            # constants: the statistics' own contribution to d/d(fullres_input) would be dropped.
            # The composed graph (and the reference's TF graph) carries it, so compose here.
            fusable = False
        if not fusable:
            return super().forward(lowres_input, fullres_input)
        from . import hdrnet_ops
        coeffs = self.coefficients(lowres_input)
        gs = coeffs.shape
        if self.training:
            # batch statistics of the (never materialised) conv1 output, from the input's moments
            sums, moments = hdrnet_ops.input_moments(fullres_input)
            npx = fullres_input.numel() // fullres_input.shape[3]
            conv1, conv2 = self.guide.folded_batch(sums, moments, npx)
        prescaled = False
        if self.training:
            pass
        elif differentiable:
            conv1, conv2 = self.guide.folded(detach=False)
        else:
            conv1, conv2, prescaled = self.guide.inference_params(self.prescale_guide)
        # inference (nothing differentiable): the model opts into the hardware sigmoid -- explicitly, HDRNET_GUIDE_SIGMOID_FAST
        return hdrnet_ops.bilateral_slice_apply_nnguide(
            coeffs.reshape(gs[0], gs[1], gs[2], gs[3], gs[4] * gs[5]), fullres_input, conv1, conv2,
            has_offset=True, fast_sigmoid=self.fast_sigmoid and not self.training and not differentiable,
            prescaled=prescaled)


class _SplitLevels(torch.autograd.Function):
    """The pyramid's per-level coefficient grids, ``coeffs[:, :, :, :, 3 l : 3 l + 3, :]`` flattened to 12 channels
    (hdrnet/models.py:280), as contiguous tensors in ONE copy (level-major), and their gradients put back with ONE stack:
    as three slices + reshapes autograd makes three copies forward and, backward, three zero fills, three slice copies and
    two full-size adds (~65 us of a graph-captured training step)."""

    @staticmethod
    def forward(ctx, coeffs, n_levels):
        gs = coeffs.shape  # [B, GH, GW, GD, n_out = 3 L, n_in]
        ctx.gs = gs
        per = gs[4] // n_levels * gs[5]
        lm = coeffs.reshape(gs[0], gs[1], gs[2], gs[3], n_levels, per).permute(4, 0, 1, 2, 3, 5).contiguous()
        return tuple(lm[l] for l in range(n_levels))

    @staticmethod
    def backward(ctx, *grads):
        gs = ctx.gs
        full = [g if g is not None else torch.zeros(gs[0], gs[1], gs[2], gs[3], gs[4] // len(grads) * gs[5],
                                                    dtype=grads[0].dtype if grads[0] is not None else torch.float32,
                                                    device=next(x for x in grads if x is not None).device) for g in grads]
        return torch.stack(full, dim=4).reshape(gs), None


class HDRNetGaussianPyrNN(HDRNetPointwiseNNGuide):
    """``hdrnet/models.py:213-289``: three pyramid levels, one guide and one 3x4 slice-apply per
    level (coefficient slices ``[:, :, :, :, 3*il:3*il+3, :]``), coarse-to-fine bilinear
    (align_corners) up-adds."""

    n_scales = 3
    n_out, n_in = 9, 4

    def __init__(self, params: Optional[Dict] = None):
        super().__init__(params)
        self.coefficients.n_levels = self.n_scales

    def _make_guide(self) -> nn.Module:
        return nn.ModuleList([_PointwiseNNGuide(self.params["guide_complexity"]) for _ in range(self.n_scales)])

    @staticmethod
    def _resize(x_nhwc: torch.Tensor, h: int, w: int) -> torch.Tensor:
        y = F.interpolate(x_nhwc.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=True)
        return y.permute(0, 2, 3, 1).contiguous()

    def forward(self, lowres_input: torch.Tensor, fullres_input: torch.Tensor) -> torch.Tensor:
        fusable = (self.fuse_guide and fullres_input.is_cuda and fullres_input.shape[3] == 3
                   and fullres_input.shape[2] % 16 == 0)
        if fusable and not self.training and not torch.is_grad_enabled():
            return self._forward_fused(lowres_input, fullres_input)
        # The fused autograd path builds the pyramid levels with the resize kernel outside autograd and
        # (in training) treats the batch-norm statistics as constants: complete for the parameters, NOT
        # for d/d(fullres_input) -- when the input itself requires a gradient, compose the torch ops.
        input_grad = torch.is_grad_enabled() and fullres_input.requires_grad
        if fusable and not input_grad and self.params["guide_complexity"] in (4, 8, 16):
            return self._forward_fused_differentiable(lowres_input, fullres_input)
        coeffs = self.coefficients(lowres_input)
        lvls: List[torch.Tensor] = [fullres_input]
        h, w = fullres_input.shape[1:3]
        for _ in range(self.n_scales - 1):
            h, w = h // 2, w // 2
            lvls.append(self._resize(lvls[-1], h, w))
        guides = [g(lvl) for g, lvl in zip(self.guide, lvls)]
        current = None
        for il, (lvl, gd) in enumerate(reversed(list(zip(lvls, guides)))):  # models.py:278
            c = coeffs[:, :, :, :, il * 3:(il + 1) * 3, :].contiguous()
            out = layers.bilateral_slice_apply(c, gd, lvl, has_offset=True)
            current = out if current is None else self._resize(current, out.shape[1], out.shape[2]) + out
        return current

    def _forward_fused_differentiable(self, lowres_input: torch.Tensor, fullres_input: torch.Tensor) -> torch.Tensor:
        """Training / autograd path: per level the differentiable fused op (guide network + slice-
        apply, batch-norm statistics from the level's input moments) and the differentiable up-add kernel."""
        from . import hdrnet_ops
        coeffs = self.coefficients(lowres_input)
        gs = coeffs.shape
        lvls: List[torch.Tensor] = [fullres_input]
        h, w = fullres_input.shape[1:3]
        for _ in range(self.n_scales - 1):
            h, w = h // 2, w // 2
            with torch.no_grad():
                lvls.append(hdrnet_ops.resize_bilinear(lvls[-1], h, w))
        current = None
        grids = _SplitLevels.apply(coeffs, self.n_scales)
        for il, (lvl, gnet) in enumerate(reversed(list(zip(lvls, self.guide)))):  # models.py:278
            c = grids[il]  # coeffs[:, :, :, :, 3 il : 3 il + 3, :] as [B, GH, GW, GD, 12]
            if self.training:
                sums, moments = hdrnet_ops.input_moments(lvl)
                conv1, conv2 = gnet.folded_batch(sums, moments, lvl.numel() // lvl.shape[3])
            else:
                conv1, conv2 = gnet.folded(detach=False)
            out = hdrnet_ops.bilateral_slice_apply_nnguide(c, lvl, conv1, conv2, has_offset=True)
            current = out if current is None else hdrnet_ops.upsample_add(current, out)  # resize + add, one pass each way
        return current

    def _forward_fused(self, lowres_input: torch.Tensor, fullres_input: torch.Tensor) -> torch.Tensor:
        """Inference with SURVEY.md section 8f rows 2 + 4 fused: per level ONE kernel evaluates the
        level's guide network, slices, applies and adds the bilinearly up-sampled coarser result;
        the multi-scale input comes from the NHWC resize kernel.  5 launches instead of ~40, and
        no full-resolution intermediate besides the two down-sampled inputs."""
        from . import hdrnet_ops
        grids = self.coefficients.levels(lowres_input)
        lvls: List[torch.Tensor] = [fullres_input]
        h, w = fullres_input.shape[1:3]
        for _ in range(self.n_scales - 1):
            h, w = h // 2, w // 2
            lvls.append(hdrnet_ops.resize_bilinear(lvls[-1], h, w))
        current = None
        for il, (lvl, gnet) in enumerate(reversed(list(zip(lvls, self.guide)))):  # models.py:278
            c = grids[il]
            conv1, conv2, prescaled = gnet.inference_params(self.prescale_guide)
            if current is None:
                current = hdrnet_ops.bilateral_slice_apply_nnguide(c, lvl, conv1, conv2, has_offset=True,
                                                                   fast_sigmoid=self.fast_sigmoid, prescaled=prescaled)
            else:
                current = hdrnet_ops.bilateral_slice_apply_upadd(c, lvl, current, guide_conv1=conv1,
                                                                 guide_conv2=conv2, has_offset=True,
                                                                 fast_sigmoid=self.fast_sigmoid, prescaled=prescaled)
        return current
