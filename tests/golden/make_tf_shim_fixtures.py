#!/usr/bin/env python
"""Golden vectors of the reference's MODEL GRAPHS, made by executing the reference's own graph code
(``/root/reference/hdrnet/layers.py`` + ``hdrnet/models.py``, loaded from where they lie, not copied) on the
eager numpy stand-in for the TensorFlow 1.x API under ``oracle/tf1_shim`` -- this image has no TensorFlow.

    python tests/golden/make_tf_shim_fixtures.py            # writes tests/golden/tf_shim/*.npz

What that pins (and what it cannot) is stated in ``oracle/tf1_shim/tensorflow/__init__.py``: the graph's
wiring, names, shapes and formulas are the reference's; the conventions of TensorFlow's own kernels (SAME
padding, batch-norm defaults, the legacy align_corners resize) are restated there from TensorFlow's source.
The custom op under the graph (``hdrnet.hdrnet_ops``: a TensorFlow op library that cannot be built here) is
bound to ``oracle/_ref`` = the reference's ``bilateral_slice_apply.cc`` / ``bilateral_slice.cc`` compiled
unchanged, so ``output`` is the reference's graph over the reference's CPU kernel.

Python 2: the reference's graph code is Python 2 (``reversed(zip(...))``, models.py:279; ``sz / 2`` on an int32
shape, :259).  The modules are executed with Python 2's ``zip`` (a list) in their globals and the shim's integer
``/``; their source is not edited.  The sha256 of both files goes into every fixture.

Fixture layout = ``tools/export_tf_fixtures.py`` (the recipe for a machine WITH TensorFlow; its output would
go to ``tests/golden/tf/``): ``var/<name>:0`` per variable, ``lowres_input``, ``fullres_input``,
``bilateral_coefficients``, ``guide`` / ``guide_<l>``, ``multiscale_<l>``, ``output``, plus ``params_json``,
``is_training``, ``source`` and the hashes.  Variables are rounded to bf16-representable float32 values (the
files compress to half); batch-norm statistics, biases and the curve parameters are moved off their trivial
initial values first, as the TensorFlow exporter does.

Consumers: tests/test_tf_shim_fixtures.py (coefficients / guide / pyramid levels of the torch modules on the
CPU; the models' output through the HIP path under ``-m gpu``; the shim's own restatements against torch; and,
where /root/reference exists, that this script reproduces the committed files bit for bit).
"""
import builtins
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(ROOT, "oracle", "tf1_shim")
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "tf_shim")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def load_reference(reference=REFERENCE):
    """(tf shim, hdrnet.layers, hdrnet.models, hdrnet.metrics, {file: sha256}) with the reference's files
    executed unchanged on the shim."""
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import tensorflow as tf
    if not hasattr(tf, "_STATE"):
        raise RuntimeError("a real TensorFlow is importable here: use tools/export_tf_fixtures.py instead")
    import oracle
    ref = oracle.ref()          # the reference's C++ compiled unchanged (oracle/Makefile)

    pkg = types.ModuleType("hdrnet")
    pkg.__path__ = []           # nothing else of the reference's package is imported
    ops = types.ModuleType("hdrnet.hdrnet_ops")

    def bilateral_slice_apply(grid, guide, input, has_offset=True, name=None):   # noqa: A002
        f = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)       # noqa: E731
        return tf._t(ref.bilateral_slice_apply(f(grid), f(guide), f(input), has_offset))

    def bilateral_slice(grid, guide, name=None):
        f = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)       # noqa: E731
        return tf._t(ref.bilateral_slice(f(grid), f(guide)))

    ops.bilateral_slice_apply, ops.bilateral_slice = bilateral_slice_apply, bilateral_slice
    pkg.hdrnet_ops = ops
    sys.modules["hdrnet"], sys.modules["hdrnet.hdrnet_ops"] = pkg, ops

    hashes = {}
    mods = {}
    for name in ("layers", "models", "metrics"):
        path = os.path.join(reference, "hdrnet", name + ".py")
        hashes["hdrnet/%s.py" % name] = _sha(path)
        mod = types.ModuleType("hdrnet." + name)
        mod.__file__ = path
        mod.__dict__["zip"] = lambda *a: list(builtins.zip(*a))     # Python 2's zip
        sys.modules["hdrnet." + name] = mod
        setattr(pkg, name, mod)
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), mod.__dict__)
        mods[name] = mod
    graph = {k: v for k, v in hashes.items() if "metrics" not in k}       # what the model fixtures executed
    return tf, mods["layers"], mods["models"], mods["metrics"], graph, {"hdrnet/metrics.py": hashes["hdrnet/metrics.py"]}


def _bf16_round(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def _perturb(tf, rng):
    """Move statistics / biases / curve parameters off their trivial initial values (tools/export_tf_fixtures.py
    does the same through session assigns), then round everything to bf16-representable values."""
    for name, v in tf._STATE.variables.items():
        s = v.shape
        if name.endswith("moving_mean"):
            v[...] = rng.randn(*s) * 0.1
        elif name.endswith("moving_variance"):
            v[...] = 0.5 + rng.rand(*s)
        elif name.endswith("BatchNorm/beta") or name.endswith("biases") or name.endswith("ccm_bias"):
            v[...] = rng.randn(*s) * 0.05
        elif name.endswith("slopes") or name.endswith("guide/ccm"):
            v[...] = v + rng.randn(*s) * 0.05
        elif name.endswith("shifts"):
            v[...] = v + rng.randn(*s) * 0.01
        v[...] = _bf16_round(v)


def run_model(tf, models, cls, params, lowres, fullres, is_training, seed):
    """Two eager passes of ``cls.inference`` under variable_scope('inference') (hdrnet/bin/freeze_graph.py:59-66,
    hdrnet/bin/train.py:113-116): the first creates the variables, the second -- after the perturbation, with
    reuse -- computes the vectors."""
    tf.reset_default_graph()
    tf.set_random_seed(seed)
    np.random.seed(seed)          # HDRNetCurves._guide draws its ccm perturbation from np.random (models.py:151)
    rng = np.random.RandomState(seed + 1)
    mdl = getattr(models, cls)
    with tf.variable_scope("inference"):
        mdl.inference(tf._t(lowres), tf._t(fullres), params, is_training=is_training)
    _perturb(tf, rng)
    tf._STATE.collections.clear()
    tf.get_variable_scope().reuse_variables()
    before = {k: v.copy() for k, v in tf._STATE.variables.items()}
    with tf.variable_scope("inference"):
        out = mdl.inference(tf._t(lowres), tf._t(fullres), params, is_training=is_training)
    assert all(np.array_equal(before[k], v) for k, v in tf._STATE.variables.items())
    f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)    # noqa: E731
    fx = {"bilateral_coefficients": f32(tf.get_collection("bilateral_coefficients")[0]), "output": f32(out)}
    guides = tf.get_collection("guide")
    if len(guides) == 1:
        fx["guide"] = f32(guides[0])
    else:
        for l, g in enumerate(guides):
            fx["guide_%d" % l] = f32(g)
    for l, m in enumerate(tf.get_collection("multiscale")):
        fx["multiscale_%d" % l] = f32(m)
    assert f32(tf.get_collection("output")[0]).shape == fx["output"].shape
    for v in tf.global_variables():
        fx["var/" + v.name] = v.value.copy()
    return fx


def slice_apply_f64(grid, guide, inp):
    """BilateralSliceApply forward (hdrnet/ops/bilateral_slice_apply.cc:24-82, has_offset) with every product and sum in
    float64 and the guide as a float64 variable -- the op under the finite-difference gradients below, where the float32
    kernel's rounding (1e-7) would drown a 1e-4 step.  grid [B, GH, GW, GD, 12], guide [B, H, W], inp [B, H, W, 3]."""
    grid, guide, inp = (np.asarray(a, dtype=np.float64) for a in (grid, guide, inp))
    B, GH, GW, GD, _ = grid.shape
    _, H, W = guide.shape
    G = grid.reshape(B, GH, GW, GD, 3, 4)
    gx = (np.arange(W) + 0.5) * GW / W
    gy = (np.arange(H) + 0.5) * GH / H
    gz = guide * GD
    fx, fy, fz = np.floor(gx - 0.5).astype(int), np.floor(gy - 0.5).astype(int), np.floor(gz - 0.5).astype(int)
    coeff = np.zeros((B, H, W, 3, 4))
    bidx = np.arange(B)[:, None, None]
    for dy in (0, 1):
        yy = fy + dy
        wy = np.maximum(1 - np.abs(yy + 0.5 - gy), 0)[None, :, None]
        yc = np.clip(yy, 0, GH - 1)[None, :, None]
        for dx in (0, 1):
            xx = fx + dx
            wx = np.maximum(1 - np.abs(xx + 0.5 - gx), 0)[None, None, :]
            xc = np.clip(xx, 0, GW - 1)[None, None, :]
            for dz in (0, 1):
                zz = fz + dz
                wz = np.maximum(1 - np.sqrt((zz + 0.5 - gz) ** 2 + 1e-8), 0)       # the smoothed tent, numerics.h:104-114
                zc = np.clip(zz, 0, GD - 1)
                coeff += (wy * wx * wz)[..., None, None] * G[bidx, yc, xc, zc]
    return np.einsum("bhwij,bhwj->bhwi", coeff[..., :3], inp) + coeff[..., 3]


def gradient_fixture(tf, models, hashes, out_dir, fname="gradients_fd", is_training=False, batch=1, seed=4321,
                     cls="HDRNetPointwiseNNGuide"):
    """Graph-level GRADIENTS of the reference's graph code: central differences of l2_loss(target, inference(...)) in
    float64 with respect to three entries of every variable (h = 2e-6), the slice-apply evaluated in float64."""
    params = dict(net_input_size=128, spatial_bin=16, luma_bins=4, channel_multiplier=1, guide_complexity=8,
                  batch_norm=False, batch_size=batch)
    rng = np.random.RandomState(seed)
    lowres = (rng.randint(0, 256, (batch, 128, 128, 3)) / 255.0).astype(np.float32)
    fullres = (rng.randint(0, 256, (batch, 32, 48, 3)) / 255.0).astype(np.float32)
    target = (rng.randint(0, 256, (batch, 32, 48, 3)) / 255.0).astype(np.float32)
    ops = sys.modules["hdrnet.hdrnet_ops"]
    f32_op = ops.bilateral_slice_apply
    fx = run_model(tf, models, cls, params, lowres, fullres, is_training, seed)  # float32 op: the variables + a check below
    mdl = getattr(models, cls)

    def loss():
        tf._STATE.collections.clear()
        with tf.variable_scope("inference"):
            out = mdl.inference(tf._t(lowres), tf._t(fullres), params, is_training=is_training)
        return float(np.mean(np.square(np.asarray(out, dtype=np.float64) - target.astype(np.float64))))

    try:
        ops.bilateral_slice_apply = lambda grid, guide, input, has_offset=True, name=None: tf._t(   # noqa: A002, E731
            slice_apply_f64(grid, guide, input))
        tf._STATE.collections.clear()
        with tf.variable_scope("inference"):
            out64 = np.asarray(mdl.inference(tf._t(lowres), tf._t(fullres), params, is_training=is_training))
        assert np.abs(out64 - fx["output"]).max() < 2e-6 * max(1.0, np.abs(out64).max()), "float64 op != the reference op"
        names, index, grads = [], [], []
        pick = np.random.RandomState(seed + 2)
        h = 2e-6   # small: the op has kinks where a z tap enters or leaves (|dz| = 1); a step that straddles them biases the quotient
        for name, v in tf._STATE.variables.items():
            cand = []
            for flat in sorted(set(pick.randint(0, v.size, 6).tolist())):
                v0 = v.reshape(-1)[flat]
                # variables are float32 storage: step to representable values and divide by the step actually taken
                hi_v, lo_v = np.float32(v0 + h), np.float32(v0 - h)
                v.reshape(-1)[flat] = hi_v
                lp = loss()
                v.reshape(-1)[flat] = lo_v
                lm = loss()
                v.reshape(-1)[flat] = v0
                cand.append((flat, (lp - lm) / (float(hi_v) - float(lo_v))))
            # the three largest of six random entries (dead ReLU units give exact zeros: one of those is kept if present)
            cand.sort(key=lambda t: -abs(t[1]))
            for flat, g in cand[:3]:
                names.append(name + ":0")
                index.append(flat)
                grads.append(g)
        l0 = loss()
    finally:
        ops.bilateral_slice_apply = f32_op
    keep = {k: a for k, a in fx.items() if k.startswith("var/")}
    np.savez_compressed(os.path.join(out_dir, fname + ".npz"), lowres_input=lowres, fullres_input=fullres,
                        target=target, loss=np.asarray(l0),
                        fd_names=np.asarray(names), fd_index=np.asarray(index, dtype=np.int64),
                        fd_grad=np.asarray(grads, dtype=np.float64), fd_step=np.asarray(h), model=np.asarray(cls),
                        params_json=np.asarray(json.dumps(params, sort_keys=True)), is_training=np.asarray(is_training),
                        reference_sha256=np.asarray(json.dumps(hashes, sort_keys=True)), **keep)
    g = np.asarray(grads)
    print(fname + ": loss %.6f, %d entries, |grad| from %.2e to %.2e" % (l0, len(g), np.abs(g).min(), np.abs(g).max()))


DEFAULT = dict(net_input_size=256, spatial_bin=16, luma_bins=8, channel_multiplier=1, guide_complexity=16,
               batch_norm=True, batch_size=1)     # hdrnet/bin/train.py:227-236

# (fixture name, model class, parameters, is_training, [B, H, W] of the full-resolution input)
CASES = [
    ("HDRNetCurves", "HDRNetCurves", DEFAULT, False, (1, 48, 80)),
    ("HDRNetPointwiseNNGuide", "HDRNetPointwiseNNGuide", DEFAULT, False, (1, 48, 80)),
    ("HDRNetGaussianPyrNN", "HDRNetGaussianPyrNN", DEFAULT, False, (1, 48, 80)),
    # how every training script of the reference runs (scripts/*/*.sh: --nobatch_norm), other hyper-parameters, B = 2
    ("HDRNetPointwiseNNGuide__nobn_sb8_lb4_cm2_gc8", "HDRNetPointwiseNNGuide",
     dict(net_input_size=128, spatial_bin=8, luma_bins=4, channel_multiplier=2, guide_complexity=8,
          batch_norm=False, batch_size=2), False, (2, 40, 56)),
    # training mode: every batch norm normalises with the batch's own statistics (hdrnet/layers.py:46: is_training)
    ("HDRNetCurves__training_lb4", "HDRNetCurves",
     dict(net_input_size=128, spatial_bin=16, luma_bins=4, channel_multiplier=1, guide_complexity=16,
          batch_norm=True, batch_size=3), True, (3, 32, 48)),
    # the guide network's own batch norm in TRAINING mode (hdrnet/models.py:205: batch_norm=True whatever the flag says):
    # statistics over every full-resolution pixel of the batch -- what hdrnet_guide_fold_batch_f32 folds from the moments
    ("HDRNetPointwiseNNGuide__training_nobn_lb4_gc8", "HDRNetPointwiseNNGuide",
     dict(net_input_size=128, spatial_bin=16, luma_bins=4, channel_multiplier=1, guide_complexity=8,
          batch_norm=False, batch_size=2), True, (2, 48, 64)),
    # the pyramid in training mode: three guide networks with batch statistics, the up-adds (hdrnet_ops.upsample_add)
    ("HDRNetGaussianPyrNN__training_nobn_lb4", "HDRNetGaussianPyrNN",
     dict(net_input_size=128, spatial_bin=16, luma_bins=4, channel_multiplier=1, guide_complexity=16,
          batch_norm=False, batch_size=1), True, (1, 64, 96)),
]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--only", default=None, help="write this one fixture only (the regeneration test)")
    args = ap.parse_args()
    out_dir = args.out
    tf, layers, models, metrics, hashes, metrics_hash = load_reference()
    os.makedirs(out_dir, exist_ok=True)
    for i, (name, cls, params, is_training, (b, h, w)) in enumerate(CASES):
        if args.only not in (None, name):
            continue
        seed = 1234 + 17 * i
        rng = np.random.RandomState(seed)
        n = params["net_input_size"]
        lowres = (rng.randint(0, 256, (b, n, n, 3)) / 255.0).astype(np.float32)      # 8-bit pixel values: the
        fullres = (rng.randint(0, 256, (b, h, w, 3)) / 255.0).astype(np.float32)     # files compress
        fx = run_model(tf, models, cls, params, lowres, fullres, is_training, seed)
        fx.update(lowres_input=lowres, fullres_input=fullres, is_training=np.asarray(is_training),
                  model=np.asarray(cls), params_json=np.asarray(json.dumps(params, sort_keys=True)),
                  source=np.asarray("oracle/tf1_shim (eager numpy stand-in for TensorFlow 1.x) + oracle/_ref"),
                  reference_sha256=np.asarray(json.dumps(hashes, sort_keys=True)))
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **fx)
        nvar = sum(k.startswith("var/") for k in fx)
        print("%-50s %3d variables  coefficients %s  output %s  %.0f KiB" % (
            name, nvar, fx["bilateral_coefficients"].shape, fx["output"].shape, os.path.getsize(path) / 1024))

    # the 6-D wrappers of hdrnet/layers.py:99-148, :153-199 on their own: slice + apply == slice_apply
    rng = np.random.RandomState(99)
    grid = rng.randn(2, 5, 6, 4, 3, 4).astype(np.float32)
    guide = rng.rand(2, 21, 34).astype(np.float32)
    inp = rng.rand(2, 21, 34, 3).astype(np.float32)
    if args.only in (None, "layers_wrappers"):
        sliced = layers.bilateral_slice(tf._t(grid), tf._t(guide))
        np.savez_compressed(os.path.join(out_dir, "layers_wrappers.npz"), grid=grid, guide=guide, input=inp,
                            sliced=np.asarray(sliced, dtype=np.float32),
                            applied=np.asarray(layers.apply(sliced, tf._t(inp)), dtype=np.float32),
                            slice_apply=np.asarray(layers.bilateral_slice_apply(tf._t(grid), tf._t(guide), tf._t(inp)),
                                                   dtype=np.float32),
                            reference_sha256=np.asarray(json.dumps(hashes, sort_keys=True)))
        print("layers_wrappers: sliced %s" % (np.asarray(sliced).shape,))

    if args.only in (None, "gradients_fd"):
        gradient_fixture(tf, models, hashes, out_dir)
    if args.only in (None, "gradients_fd_training"):   # the guide network's batch norm on the batch's own statistics
        gradient_fixture(tf, models, hashes, out_dir, "gradients_fd_training", True, 2, 4391)
    if args.only in (None, "gradients_fd_curves"):     # the reference's default class: the curves guide's VJP
        gradient_fixture(tf, models, hashes, out_dir, "gradients_fd_curves", False, 1, 4461, "HDRNetCurves")
    if args.only in (None, "gradients_fd_pyramid"):    # three levels, their up-adds, three guide networks on batch statistics
        gradient_fixture(tf, models, hashes, out_dir, "gradients_fd_pyramid", True, 1, 4531, "HDRNetGaussianPyrNN")

    # hdrnet/metrics.py:21-33 -- the training loss and the evaluation metric of hdrnet/bin/train.py:137-143
    target = (rng.randint(0, 256, (3, 17, 23, 3)) / 255.0).astype(np.float32)    # (continues the stream of the block above)
    prediction = (target + rng.randn(*target.shape) * 0.05).astype(np.float32)
    if args.only in (None, "metrics"):
        np.savez_compressed(os.path.join(out_dir, "metrics.npz"), target=target, prediction=prediction,
                            l2_loss=np.asarray(metrics.l2_loss(tf._t(target), tf._t(prediction)), dtype=np.float64),
                            psnr=np.asarray(metrics.psnr(tf._t(target), tf._t(prediction)), dtype=np.float64),
                            reference_sha256=np.asarray(json.dumps(metrics_hash, sort_keys=True)))


if __name__ == "__main__":
    main()
