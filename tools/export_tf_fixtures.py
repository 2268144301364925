#!/usr/bin/env python
"""Export golden fixtures of the reference's MODEL GRAPHS from TensorFlow -- the recipe that would lift
the "graph-level parity vs TensorFlow unpinned" cap of SURVEY.md section 8f rows 1 and 4.

STATUS: WRITTEN, NEVER RUN.  There is no TensorFlow in the build image and no network; this script is
for a machine that has (a) the reference checkout on PYTHONPATH (``import hdrnet.models`` works) and
(b) the TensorFlow the reference's graph code needs (it uses tf.contrib.layers and tf.Session, i.e. the
1.x API: TensorFlow <= 1.15, whatever hdrnet/requirements.txt pins).  The compiled custom op
(hdrnet/ops -> lib/hdrnet_ops.so) is OPTIONAL: if ``hdrnet.hdrnet_ops`` cannot be imported the script
stubs ``bilateral_slice_apply`` with a zero placeholder and exports coefficients and guide only (the
slice-apply itself is already pinned to the reference's C++ by oracle/_ref).

    python tools/export_tf_fixtures.py --out tests/golden/tf [--size 64] [--seed 1234]

For each model class (HDRNetCurves, HDRNetPointwiseNNGuide, HDRNetGaussianPyrNN; the parameters of
hdrnet/bin/train.py:227-236: net_input_size 256, spatial_bin 16, luma_bins 8, channel_multiplier 1,
guide_complexity 16, batch_norm on) it builds ``inference(lowres, fullres, params, is_training=False)``
exactly as hdrnet/bin/freeze_graph.py:59-66 does, initialises the variables randomly (graph seed),
perturbs the batch-norm moving statistics and the curve parameters away from their trivial initial
values (moving_mean = 0, moving_variance = 1, one live knot would hide layout mistakes), runs the
graph once on a random ``size x size`` full-resolution input and the 256 x 256 low-resolution input,
and writes ``<out>/<ModelName>.npz`` with

    var/<tensorflow variable name>      every global variable, TensorFlow layout
    lowres_input, fullres_input         NHWC float32
    bilateral_coefficients              [1, GH, GW, GD, n_out, n_in]   (collection 'bilateral_coefficients')
    guide / guide_<l>                   [1, H, W] (per pyramid level for HDRNetGaussianPyrNN)
    multiscale_<l>                      the pyramid's resized inputs (tf.image.resize_images, align_corners)
    output                              [1, H, W, 3] -- only when the custom op is available

tests/test_models.py::test_tf_fixture_parity loads these through hdrnet_amd/tf_import.py and compares the
torch modules' coefficients / guide / multiscale / output with them (rtol = atol = 1e-4).
"""
import argparse
import os
import sys
import types

import numpy as np


def _install_op_stub():
    """hdrnet.layers imports hdrnet.hdrnet_ops at module load; without the compiled .so give it a stub."""
    import tensorflow as tf
    stub = types.ModuleType("hdrnet.hdrnet_ops")

    def bilateral_slice_apply(grid, guide, input, has_offset=True, name=None):  # noqa: A002
        return tf.zeros_like(input, name=name)

    def bilateral_slice(grid, guide, name=None):
        return tf.zeros_like(guide, name=name)

    stub.bilateral_slice_apply = bilateral_slice_apply
    stub.bilateral_slice = bilateral_slice
    sys.modules["hdrnet.hdrnet_ops"] = stub


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join("tests", "golden", "tf"))
    ap.add_argument("--size", type=int, default=64, help="full-resolution input is size x size (multiple of 16)")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()

    import tensorflow as tf
    if hasattr(tf, "compat") and hasattr(tf.compat, "v1") and not hasattr(tf, "contrib"):
        sys.exit("the reference's graph code needs tf.contrib (TensorFlow 1.x); this TensorFlow has none")
    have_op = True
    try:
        import hdrnet.hdrnet_ops  # noqa: F401  (loads lib/hdrnet_ops.so)
    except Exception as e:  # noqa: BLE001
        print("custom op not available (%s): exporting coefficients and guide only" % e)
        have_op = False
        _install_op_stub()
    import hdrnet.models as models

    params = dict(net_input_size=256, spatial_bin=16, luma_bins=8, channel_multiplier=1,
                  guide_complexity=16, batch_norm=True, batch_size=1)
    os.makedirs(args.out, exist_ok=True)
    rng = np.random.RandomState(args.seed)
    lowres = rng.rand(1, 256, 256, 3).astype(np.float32)
    fullres = rng.rand(1, args.size, args.size, 3).astype(np.float32)

    for name in ("HDRNetCurves", "HDRNetPointwiseNNGuide", "HDRNetGaussianPyrNN"):
        tf.reset_default_graph()
        tf.set_random_seed(args.seed)
        np.random.seed(args.seed)  # HDRNetCurves._guide draws its ccm perturbation from np.random (models.py:152)
        mdl = getattr(models, name)
        lo = tf.placeholder(tf.float32, lowres.shape, name="lowres_input")
        hi = tf.placeholder(tf.float32, fullres.shape, name="fullres_input")
        with tf.variable_scope("inference"):
            out = mdl.inference(lo, hi, params, is_training=False)
        fetch = {"bilateral_coefficients": tf.get_collection("bilateral_coefficients")[0]}
        guides = tf.get_collection("guide")
        if len(guides) == 1:
            fetch["guide"] = guides[0]
        else:
            for l, g in enumerate(guides):
                fetch["guide_%d" % l] = g
        for l, m in enumerate(tf.get_collection("multiscale")):
            fetch["multiscale_%d" % l] = m
        if have_op:
            fetch["output"] = out
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            # move the statistics / curve parameters off their trivial initial values
            for v in tf.global_variables():
                shape = v.get_shape().as_list()
                if v.name.endswith("moving_mean:0"):
                    sess.run(v.assign(rng.randn(*shape).astype(np.float32) * 0.1))
                elif v.name.endswith("moving_variance:0"):
                    sess.run(v.assign((0.5 + rng.rand(*shape)).astype(np.float32)))
                elif v.name.endswith("BatchNorm/beta:0") or v.name.endswith("biases:0") or v.name.endswith("ccm_bias:0"):
                    sess.run(v.assign(rng.randn(*shape).astype(np.float32) * 0.05))
                elif v.name.endswith("slopes:0"):
                    sess.run(v.assign((sess.run(v) + rng.randn(*shape) * 0.05).astype(np.float32)))
                elif v.name.endswith("guide/ccm:0"):
                    sess.run(v.assign((sess.run(v) + rng.randn(*shape) * 0.05).astype(np.float32)))
            values = sess.run(fetch, {lo: lowres, hi: fullres})
            variables = {"var/" + v.name: sess.run(v) for v in tf.global_variables()}
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, lowres_input=lowres, fullres_input=fullres, **values, **variables)
        print("wrote %s: %d variables, fetched %s" % (path, len(variables), sorted(values)))


if __name__ == "__main__":
    main()
