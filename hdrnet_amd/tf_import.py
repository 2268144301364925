"""TensorFlow variable names of the reference's model graphs <-> the torch modules of ``models.py``.

The reference builds its graphs under ``tf.variable_scope('inference')`` with
``tf.contrib.layers.convolution2d`` / ``fully_connected`` / ``batch_norm`` (hdrnet/layers.py:25-93), so a
checkpoint (or ``tools/export_tf_fixtures.py``) names every tensor ``inference/<scope>/<layer>/...``:

    .../weights                      conv: [kh, kw, cin, cout]   fc: [cin, cout]
    .../biases                       [cout]                       (layers without batch norm)
    .../BatchNorm/beta               [cout]                       (center=True, scale=False)
    .../BatchNorm/moving_mean, moving_variance

with the scopes of hdrnet/models.py: ``coefficients/{splat/conv<i>, global/conv<i>, global/fc<i>,
local/conv<i>, prediction/conv1}`` (:62-142), ``guide/{ccm, ccm_bias, shifts, slopes, channel_mixing}``
(:145-190), ``guide/{conv1, conv2}`` (:203-210) and ``guide/level_<l>/{conv1, conv2}`` (:268-275).

``load_tf_variables`` fills a torch model from such a dict; ``export_tf_variables`` is its inverse (used by
the tests to prove the mapping round-trips, and by nothing else).  Whether the mapping matches a REAL
TensorFlow checkpoint can only be shown with fixtures exported on a machine that has TensorFlow
(tools/export_tf_fixtures.py; tests/test_models.py::test_tf_fixture_parity consumes them when present).
"""
from __future__ import annotations

from typing import Dict, Iterator, Tuple

import numpy as np
import torch

from . import models

PREFIX = "inference/"


def _conv_like(coeffs: "models._Coefficients") -> Iterator[Tuple[str, torch.nn.Module]]:
    for i, m in enumerate(coeffs.splat):
        yield f"coefficients/splat/conv{i + 1}", m
    for i, m in enumerate(coeffs.global_conv):
        yield f"coefficients/global/conv{i + 1}", m
    yield "coefficients/global/fc1", coeffs.fc1
    yield "coefficients/global/fc2", coeffs.fc2
    yield "coefficients/global/fc3", coeffs.fc3
    yield "coefficients/local/conv1", coeffs.local1
    yield "coefficients/local/conv2", coeffs.local2
    yield "coefficients/prediction/conv1", coeffs.pred


def _layer_tensors(scope: str, m) -> Iterator[Tuple[str, torch.Tensor, str]]:
    """(tf name, torch tensor, kind) of one ``_Conv`` / ``_FC``; kind says how the layouts relate."""
    lin = m.conv if hasattr(m, "conv") else m.fc
    yield scope + "/weights", lin.weight, "conv" if hasattr(m, "conv") else "fc"
    if lin.bias is not None:
        yield scope + "/biases", lin.bias, "same"
    if m.bn is not None:
        yield scope + "/BatchNorm/beta", m.bn.bn.bias, "same"
        yield scope + "/BatchNorm/moving_mean", m.bn.bn.running_mean, "same"
        yield scope + "/BatchNorm/moving_variance", m.bn.bn.running_var, "same"


def _nn_guide_tensors(scope: str, g: "models._PointwiseNNGuide") -> Iterator[Tuple[str, torch.Tensor, str]]:
    yield scope + "/conv1/weights", g.w1, "conv1x1"            # [1, 1, cin, n] <-> [cin, n]
    yield scope + "/conv1/BatchNorm/beta", g.bn.bias, "same"
    yield scope + "/conv1/BatchNorm/moving_mean", g.bn.running_mean, "same"
    yield scope + "/conv1/BatchNorm/moving_variance", g.bn.running_var, "same"
    yield scope + "/conv2/weights", g.w2, "conv1x1_to1"        # [1, 1, n, 1] <-> [n]
    yield scope + "/conv2/biases", g.b2, "scalar"              # [1] <-> []


def _guide_tensors(model) -> Iterator[Tuple[str, torch.Tensor, str]]:
    g = model.guide
    if isinstance(g, models._CurvesGuide):
        yield "guide/ccm", g.ccm, "same"                        # x @ ccm in both
        yield "guide/ccm_bias", g.ccm_bias, "same"
        yield "guide/shifts", g.shifts, "curve"                 # [1, 1, c, k] <-> [c, k]
        yield "guide/slopes", g.slopes, "curve5"                # [1, 1, 1, c, k] <-> [c, k]
        yield "guide/channel_mixing/weights", g.mix_w, "conv1x1_to1"
        yield "guide/channel_mixing/biases", g.mix_b, "scalar"
    elif isinstance(g, models._PointwiseNNGuide):
        yield from _nn_guide_tensors("guide", g)
    else:  # the pyramid: one network per level
        for lvl, net in enumerate(g):
            yield from _nn_guide_tensors(f"guide/level_{lvl}", net)


def _all_tensors(model) -> Iterator[Tuple[str, torch.Tensor, str]]:
    for scope, m in _conv_like(model.coefficients):
        yield from _layer_tensors(scope, m)
    yield from _guide_tensors(model)


def _to_torch(a: np.ndarray, kind: str) -> np.ndarray:
    if kind == "conv":
        return np.transpose(a, (3, 2, 0, 1))  # [kh, kw, cin, cout] -> [cout, cin, kh, kw]
    if kind == "fc":
        return a.T
    if kind == "conv1x1":
        return a.reshape(a.shape[-2], a.shape[-1])
    if kind in ("conv1x1_to1", "scalar"):
        return a.reshape(-1) if kind == "conv1x1_to1" else a.reshape(())
    if kind in ("curve", "curve5"):
        return a.reshape(a.shape[-2], a.shape[-1])
    return a


def _to_tf(a: np.ndarray, kind: str) -> np.ndarray:
    if kind == "conv":
        return np.transpose(a, (2, 3, 1, 0))
    if kind == "fc":
        return a.T
    if kind == "conv1x1":
        return a.reshape(1, 1, *a.shape)
    if kind == "conv1x1_to1":
        return a.reshape(1, 1, -1, 1)
    if kind == "scalar":
        return a.reshape(1)
    if kind == "curve":
        return a.reshape(1, 1, *a.shape)
    if kind == "curve5":
        return a.reshape(1, 1, 1, *a.shape)
    return a


def _key(variables: Dict[str, np.ndarray], name: str) -> str:
    for cand in (PREFIX + name + ":0", PREFIX + name, name + ":0", name):
        if cand in variables:
            return cand
    raise KeyError(f"TensorFlow variable '{PREFIX}{name}' not in the fixture / checkpoint dump")


# Entries of a checkpoint / fixture dump that are not model state (hdrnet/bin/train.py:120-160: the optimizer's slots and
# the step counter; a fixture's own inputs / outputs carry no 'inference/' prefix at all).
_NOT_MODEL_STATE = ("/Adam", "/Adam_1", "beta1_power", "beta2_power", "global_step", "/ExponentialMovingAverage")


def load_tf_variables(model, variables: Dict[str, np.ndarray], strict: bool = True) -> None:
    """Fill ``model`` (HDRNetCurves / HDRNetPointwiseNNGuide / HDRNetGaussianPyrNN) from a dict
    {TensorFlow variable name: array}; every tensor of the model must be present, shapes must agree.
    ``strict``: a variable under the graph's ``inference/`` scope that the mapping does NOT consume raises --
    e.g. ``.../BatchNorm/gamma`` of a checkpoint trained with ``scale=True`` (the reference builds its batch norms
    without a scale, hdrnet/layers.py:40-58; such a checkpoint would otherwise load silently wrong)."""
    consumed = set()
    with torch.no_grad():
        for name, t, kind in _all_tensors(model):
            key = _key(variables, name)
            consumed.add(key)
            a = _to_torch(np.asarray(variables[key], dtype=np.float32), kind)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError(f"{name}: TensorFlow shape maps to {a.shape}, the module has {tuple(t.shape)}")
            t.copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(t.shape).to(t.device))  # (0-d stays 0-d)
    extra = sorted(k for k in variables if k not in consumed and (k.startswith(PREFIX) or k.startswith("/" + PREFIX))
                   and not any(tag in k for tag in _NOT_MODEL_STATE))
    if extra and strict:
        raise ValueError("TensorFlow variables the model has no place for (a checkpoint of another architecture / "
                         f"batch norm with scale=True?): {extra[:8]}{' ...' if len(extra) > 8 else ''}")


def export_tf_variables(model) -> Dict[str, np.ndarray]:
    """The model's tensors under the reference's TensorFlow names and layouts (the inverse of
    ``load_tf_variables``)."""
    return {PREFIX + name + ":0": _to_tf(t.detach().cpu().numpy().astype(np.float32), kind).copy()
            for name, t, kind in _all_tensors(model)}
