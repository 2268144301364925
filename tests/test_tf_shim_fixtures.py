"""Graph-level parity of the model modules against the reference's OWN graph code.

``tests/golden/tf_shim/*.npz`` were computed by ``/root/reference/hdrnet/models.py`` + ``layers.py`` executed
unchanged on ``oracle/tf1_shim`` (an eager numpy stand-in for the TensorFlow 1.x API; this image has no
TensorFlow) over ``oracle/_ref`` (the reference's slice kernels compiled unchanged) by
``tests/golden/make_tf_shim_fixtures.py``.  The reference's Python therefore decides the wiring, the names and
shapes of every variable, the flatten / fusion / unroll orders, the guide formulas and the pyramid; TensorFlow's
kernel conventions are the shim's restatements, which the tests at the bottom check against torch.  A fixture
from real TensorFlow (``tools/export_tf_fixtures.py`` -> ``tests/golden/tf/``) would be consumed by
tests/test_models.py::test_tf_fixture_parity in the same way.
"""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from hdrnet_amd import models, tf_import

FIXTURES = os.path.join(ROOT, "tests", "golden", "tf_shim")
MODEL_FIXTURES = ["HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN",
                  "HDRNetPointwiseNNGuide__nobn_sb8_lb4_cm2_gc8", "HDRNetCurves__training_lb4",
                  "HDRNetPointwiseNNGuide__training_nobn_lb4_gc8", "HDRNetGaussianPyrNN__training_nobn_lb4"]
REFERENCE = "/root/reference"


def _load(name):
    with np.load(os.path.join(FIXTURES, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def _model(fx):
    params = json.loads(str(fx["params_json"]))
    m = getattr(models, str(fx["model"]))(params)
    m.train(bool(fx["is_training"]))
    tf_import.load_tf_variables(m, {k[len("var/"):]: a for k, a in fx.items() if k.startswith("var/")})
    return m


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_reference_graph_coefficients_guide_pyramid(name):
    """hdrnet/models.py:62-142 (_coefficients), :145-190 / :203-210 (_guide), :253-275 (the pyramid's levels)."""
    fx = _load(name)
    m = _model(fx)
    m.fuse_guide = False
    lo, hi = torch.from_numpy(fx["lowres_input"]), torch.from_numpy(fx["fullres_input"])
    with torch.no_grad():
        coeffs = m.coefficients(lo)
        assert tuple(coeffs.shape) == fx["bilateral_coefficients"].shape       # [B, GH, GW, GD, n_out, n_in]
        # float32 torch against the shim's float64: 1.1e-5 at |c| <= 14.6 in inference mode; with the batch's own statistics
        # (training mode) a low-variance channel amplifies the convolutions' rounding, and that rounding depends on torch's
        # CPU thread count / convolution backend: 1.6e-5 to 4.1e-5 observed
        tol = dict(rtol=1e-4, atol=1e-4) if bool(fx["is_training"]) else dict(rtol=2e-5, atol=5e-5)
        np.testing.assert_allclose(coeffs.numpy(), fx["bilateral_coefficients"], **tol)
        gtol = 2e-5 if bool(fx["is_training"]) else 5e-6   # (training: float32 batch statistics, 7.5e-6 on a 16 x 24 level)
        if str(fx["model"]) == "HDRNetGaussianPyrNN":
            lvls = [hi]
            for _ in range(2):
                lvls.append(m._resize(lvls[-1], lvls[-1].shape[1] // 2, lvls[-1].shape[2] // 2))
            for l, lvl in enumerate(lvls):
                np.testing.assert_allclose(lvl.numpy(), fx["multiscale_%d" % l], rtol=0, atol=1e-5)
                np.testing.assert_allclose(m.guide[l](lvl).numpy(), fx["guide_%d" % l], rtol=0, atol=gtol)
        else:
            np.testing.assert_allclose(m.guide(hi).numpy(), fx["guide"], rtol=0, atol=gtol)


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_variable_names_and_shapes_are_the_reference_graphs(name):
    """Every variable the reference's graph creates is consumed by tf_import (strict), and the torch module's own
    export has exactly those names and shapes -- no tensor of the model lacks a TensorFlow counterpart."""
    fx = _load(name)
    m = _model(fx)          # strict=True: raises on a variable the mapping has no place for
    ours = tf_import.export_tf_variables(m)
    theirs = {k[len("var/"):]: a for k, a in fx.items() if k.startswith("var/")}
    assert sorted(ours) == sorted(theirs)
    for k in ours:
        assert ours[k].shape == theirs[k].shape, k
        np.testing.assert_array_equal(ours[k], theirs[k])


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_reference_graph_output_through_the_hip_path(name, fuse):
    """The whole model on the GPU (coefficient network, guide, slice-apply -- HIP kernels wherever the model
    routes to them) against the reference's graph over the reference's CPU slice kernel (hdrnet/models.py:36-59,
    :193-196, :277-289)."""
    fx = _load(name)
    m = _model(fx).cuda()
    m.fuse_guide = fuse
    lo, hi = torch.from_numpy(fx["lowres_input"]).cuda(), torch.from_numpy(fx["fullres_input"]).cuda()
    with torch.no_grad():
        got = m(lo, hi).cpu().numpy()
    assert got.shape == fx["output"].shape
    np.testing.assert_allclose(got, fx["output"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["gradients_fd", "gradients_fd_training", "gradients_fd_curves", "gradients_fd_pyramid"])
@pytest.mark.parametrize("fuse", [False, True])
def test_reference_graph_gradients_through_the_hip_path(fuse, fixture):
    """GRADIENTS at graph level: d l2_loss(target, inference(...)) / d (entries of every variable) as central differences of
    the reference's graph code in float64 (tests/golden/make_tf_shim_fixtures.py: gradient_fixture, the slice-apply in
    float64, step 2e-6: the reference's own analytic gradient -- its C++ op under torch autograd on the CPU -- is within
    2e-5 of these quotients; a step of 1e-4 straddles the z taps' kinks and is 2.4e-3 off) against autograd through this package's model on the GPU -- the coefficient network's backward kernels, the
    slice-apply VJPs, the guide network's VJP, the loss kernels, fp32."""
    from hdrnet_amd import metrics
    fx = _load(fixture)
    m = _model(fx).cuda()            # inference-mode graph (moving statistics) / training mode (the guide's batch statistics:
                                     # input moments + hdrnet_guide_fold_batch_f32 and its VJP when fused), gradients wanted
    m.fuse_guide = fuse
    lo, hi = torch.from_numpy(fx["lowres_input"]).cuda(), torch.from_numpy(fx["fullres_input"]).cuda()
    target = torch.from_numpy(fx["target"]).cuda()
    loss = metrics.l2_loss(target, m(lo, hi))
    np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=2e-5)
    loss.backward()
    tensors = {tf_import.PREFIX + name + ":0": (t, kind) for name, t, kind in tf_import._all_tensors(m)}
    checked, worst = 0, 0.0
    scale = float(np.abs(fx["fd_grad"]).max())
    n_param = sum("moving_" not in str(n) for n in fx["fd_names"])
    for name, flat, want in zip(fx["fd_names"], fx["fd_index"], fx["fd_grad"]):
        t, kind = tensors[str(name)]
        if t.grad is None:           # the batch norm's moving statistics: buffers here, variables in TensorFlow
            assert "moving_" in str(name), name
            continue
        got = float(tf_import._to_tf(t.grad.detach().cpu().numpy().astype(np.float64), kind).reshape(-1)[int(flat)])
        err = abs(got - float(want))
        worst = max(worst, err / (abs(float(want)) + 1e-3 * scale))
        assert err <= 5e-4 * abs(float(want)) + 1e-5 * scale, (str(name), int(flat), got, float(want))
        checked += 1
    assert checked == n_param, (checked, n_param)
    print("graph-level gradients: %d entries, worst relative error %.2e" % (checked, worst))


@pytest.mark.gpu
def test_layer_wrappers_match_the_reference_wrappers():
    """hdrnet/layers.py:99-148 on a 6-D grid: the unstack / concat / split / stack channel orders of
    bilateral_slice and the reshape of bilateral_slice_apply."""
    from hdrnet_amd import layers
    fx = _load("layers_wrappers")
    grid, guide, inp = (torch.from_numpy(fx[k]).cuda() for k in ("grid", "guide", "input"))
    sliced = layers.bilateral_slice(grid, guide)
    assert tuple(sliced.shape) == fx["sliced"].shape
    np.testing.assert_allclose(sliced.cpu().numpy(), fx["sliced"], rtol=1e-5, atol=1e-5)
    out = layers.bilateral_slice_apply(grid, guide, inp, has_offset=True)
    np.testing.assert_allclose(out.cpu().numpy(), fx["slice_apply"], rtol=1e-5, atol=1e-5)


def test_layer_wrappers_fixture_is_self_consistent():
    """layers.apply(layers.bilateral_slice(...)) == layers.bilateral_slice_apply(...) in the reference itself."""
    fx = _load("layers_wrappers")
    np.testing.assert_allclose(fx["applied"], fx["slice_apply"], rtol=1e-5, atol=1e-5)
    want = np.einsum("bhwij,bhwj->bhwi", fx["sliced"][..., :3], fx["input"]) + fx["sliced"][..., 3]
    np.testing.assert_allclose(fx["applied"], want, rtol=1e-5, atol=1e-5)


def test_metrics_match_the_reference_module():
    """hdrnet/metrics.py:21-33 executed on the shim: l2_loss(target, prediction) and psnr (mean over the batch).
    The reference's ``target - prediction`` is a float32 subtraction (the operands are float32 tensors), the sums are
    the shim's float64: 2.5e-9 from an all-float64 evaluation."""
    from hdrnet_amd import metrics
    fx = _load("metrics")
    t, p = torch.from_numpy(fx["target"]).double(), torch.from_numpy(fx["prediction"]).double()
    np.testing.assert_allclose(float(metrics.l2_loss(t, p)), float(fx["l2_loss"]), rtol=1e-7)
    np.testing.assert_allclose(float(metrics.psnr(t, p)), float(fx["psnr"]), rtol=1e-7)
    d = (torch.from_numpy(fx["target"]) - torch.from_numpy(fx["prediction"])).double()     # float32 subtraction
    np.testing.assert_allclose(float(d.square().mean()), float(fx["l2_loss"]), rtol=1e-13)


@pytest.mark.gpu
def test_metrics_kernels_match_the_reference_module():
    """The same through csrc/metrics.hip (float32 inputs, the loss kernels of the training step)."""
    from hdrnet_amd import metrics
    fx = _load("metrics")
    t, p = torch.from_numpy(fx["target"]).cuda(), torch.from_numpy(fx["prediction"]).cuda()
    np.testing.assert_allclose(float(metrics.l2_loss(t, p)), float(fx["l2_loss"]), rtol=2e-6)
    np.testing.assert_allclose(float(metrics.psnr(t, p)), float(fx["psnr"]), rtol=2e-6)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "hdrnet")), reason="needs /root/reference")
def test_committed_fixtures_are_what_the_script_computes(tmp_path):
    """Provenance: the generator, run now against /root/reference, reproduces a committed fixture (to 1e-6: float64 BLAS
    sums may differ in the last place between machines), and the reference files have the hashes the fixtures recorded."""
    import hashlib
    name = "HDRNetPointwiseNNGuide__nobn_sb8_lb4_cm2_gc8"
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_tf_shim_fixtures.py"),
                    "--out", str(tmp_path), "--only", name], check=True, timeout=300,
                   stdout=subprocess.DEVNULL)
    want = _load(name)
    with np.load(os.path.join(str(tmp_path), name + ".npz")) as z:
        assert sorted(z.files) == sorted(want)
        for k in z.files:
            if z[k].dtype.kind == "f":   # (bit-identical on this machine; a BLAS with another thread count may differ by an ulp)
                np.testing.assert_allclose(z[k], want[k], rtol=1e-6, atol=1e-7, err_msg=k)
            else:
                np.testing.assert_array_equal(z[k], want[k], err_msg=k)
    for fixture in MODEL_FIXTURES + ["layers_wrappers", "metrics", "gradients_fd", "gradients_fd_training",
                                     "gradients_fd_curves", "gradients_fd_pyramid"]:
        for rel, digest in json.loads(str(_load(fixture)["reference_sha256"])).items():
            with open(os.path.join(REFERENCE, rel), "rb") as f:
                assert hashlib.sha256(f.read()).hexdigest() == digest, (fixture, rel)


# ---- the shim's own restatements of TensorFlow's kernel conventions, against torch -----------------------------
@pytest.fixture(scope="module")
def shim():
    """oracle/tf1_shim loaded under a private name (nothing in this process should see a 'tensorflow')."""
    path = os.path.join(ROOT, "oracle", "tf1_shim", "tensorflow", "__init__.py")
    spec = importlib.util.spec_from_file_location("_tf1_shim_under_test", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shim_same_padding(shim):
    """GetWindowedOutputSizeVerbose: TensorFlow's documented example (in 13, filter 6, stride 5 -> out 3, pad 1 + 2)
    and the stride-2 3x3 layers of the splat path (even extent: 0 in front, 1 behind)."""
    assert shim._same_pad(13, 6, 5) == (3, 1, 2)
    assert shim._same_pad(256, 3, 2) == (128, 0, 1)
    assert shim._same_pad(15, 3, 2) == (8, 1, 1)
    assert shim._same_pad(16, 3, 1) == (16, 1, 1)
    assert shim._same_pad(7, 1, 1) == (7, 0, 0)


@pytest.mark.parametrize("h,w,k,stride", [(16, 16, 3, 2), (15, 18, 3, 2), (9, 7, 3, 1), (8, 6, 1, 1), (13, 13, 6, 5)])
def test_shim_conv_is_a_same_padded_cross_correlation(shim, h, w, k, stride):
    rng = np.random.RandomState(h * 100 + w)
    x = rng.randn(2, h, w, 3)
    wt = rng.randn(k, k, 3, 5)
    got = shim._conv2d_same(x, wt, stride)
    _, pt, pb = shim._same_pad(h, k, stride)
    _, pl, pr = shim._same_pad(w, k, stride)
    xt = F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    want = F.conv2d(xt, torch.from_numpy(wt).permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_shim_layers_create_the_variables_tensorflow_would(shim):
    tf = shim
    tf.reset_default_graph()
    x = tf._t(np.random.RandomState(0).rand(2, 8, 8, 3))
    init = tf.contrib.layers.variance_scaling_initializer()
    with tf.variable_scope("a"):
        with tf.name_scope("not_a_prefix"):
            y = tf.contrib.layers.convolution2d(x, 4, 3, stride=2, weights_initializer=init,
                                                normalizer_fn=tf.contrib.layers.batch_norm,
                                                normalizer_params=dict(center=True, is_training=False),
                                                biases_initializer=None, scope="conv1")
        z = tf.contrib.layers.fully_connected(tf.reshape(y, [2, -1]), 5, weights_initializer=init,
                                              biases_initializer=tf.constant_initializer(0.0), scope="fc1")
    assert list(tf._STATE.variables) == ["a/conv1/weights", "a/conv1/BatchNorm/beta", "a/conv1/BatchNorm/moving_mean",
                                         "a/conv1/BatchNorm/moving_variance", "a/fc1/weights", "a/fc1/biases"]
    assert tf._STATE.variables["a/conv1/weights"].shape == (3, 3, 3, 4)
    assert tf._STATE.variables["a/fc1/weights"].shape == (64, 5) and np.asarray(z).shape == (2, 5)
    with pytest.raises(ValueError, match="already exists"):
        with tf.variable_scope("a"):
            tf.get_variable("fc1/weights", [64, 5], initializer=init)
    tf.get_variable_scope().reuse_variables()
    with pytest.raises(ValueError, match="does not exist"):
        tf.get_variable("never_made", [1], initializer=init)
    tf.reset_default_graph()


def test_shim_batch_norm_follows_contrib_layers_defaults(shim):
    tf = shim
    tf.reset_default_graph()
    rng = np.random.RandomState(3)
    x = rng.randn(4, 5, 6, 7)
    with tf.variable_scope("s"):
        tf.contrib.layers.batch_norm(tf._t(x), is_training=False)
    assert "s/BatchNorm/gamma" not in tf._STATE.variables            # scale=False by default
    mean, var, beta = rng.randn(7) * 0.2, 0.5 + rng.rand(7), rng.randn(7) * 0.1
    tf._STATE.variables["s/BatchNorm/moving_mean"][...] = mean
    tf._STATE.variables["s/BatchNorm/moving_variance"][...] = var
    tf._STATE.variables["s/BatchNorm/beta"][...] = beta
    tf.get_variable_scope().reuse_variables()
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).double()   # noqa: E731  (variables are float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    with tf.variable_scope("s"):
        got = tf.contrib.layers.batch_norm(tf._t(x), is_training=False)
    want = F.batch_norm(xt, f32(mean), f32(var), None, f32(beta), False, 0.0, 1e-3).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    with tf.variable_scope("s"):
        got = tf.contrib.layers.batch_norm(tf._t(x), is_training=True)
    want = F.batch_norm(xt, None, None, None, f32(beta), True, 0.0, 1e-3).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    tf.reset_default_graph()


@pytest.mark.parametrize("h,w,oh,ow", [(48, 80, 24, 40), (12, 20, 24, 40), (7, 9, 3, 4), (5, 5, 1, 1)])
def test_shim_resize_is_the_legacy_align_corners_bilinear(shim, h, w, oh, ow):
    import oracle
    x = np.random.RandomState(h + w).rand(2, h, w, 3).astype(np.float32)
    got = shim.image.resize_images(shim._t(x), shim._t(np.asarray([oh, ow], np.int32)), align_corners=True)
    np.testing.assert_allclose(got, oracle.resize_bilinear_align_corners(x, oh, ow), rtol=0, atol=2e-5)
    if oh > 1 and ow > 1:
        want = F.interpolate(torch.from_numpy(x).double().permute(0, 3, 1, 2), size=(oh, ow), mode="bilinear",
                             align_corners=True).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_shim_python2_integer_division_on_shape_tensors(shim):
    s = shim.shape(np.zeros((1, 49, 80, 3)))[1:3]
    half = s / 2                                   # hdrnet/models.py:259 under Python 2: integer division
    assert half.dtype.kind == "i" and half.tolist() == [24, 40]
    assert (shim._t(np.asarray([1.0, 3.0])) / 2).tolist() == [0.5, 1.5]
    with pytest.raises(TypeError):
        shim.image.resize_images(np.zeros((1, 4, 4, 3)), np.asarray([2.0, 2.0]))
