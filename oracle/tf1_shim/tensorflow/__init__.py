"""TEST INFRASTRUCTURE ONLY -- an eager numpy stand-in for the part of the TensorFlow 1.x API that the
reference's GRAPH code calls, so that ``/root/reference/hdrnet/layers.py`` and ``hdrnet/models.py`` can be
executed UNCHANGED in an image that has no TensorFlow (``tests/golden/make_tf_shim_fixtures.py`` does that
and commits what they compute as golden vectors under ``tests/golden/tf_shim/``).

What this pins and what it does not.  Everything the reference's own Python decides -- which layers exist
and in which order, their channel counts, which of them carry a bias / a batch norm / an activation, the
variable names and shapes, the flatten order in front of fc1, the fusion, the unroll of the prediction into
[B, GH, GW, GD, n_out, n_in], the guide formulas, the pyramid's level order and its resize-and-add -- comes
from the reference's source, executed line by line.  What TensorFlow's own kernels decide is RESTATED here
from TensorFlow's published behaviour (TensorFlow is a dependency of the reference that is not under
/root/reference: hdrnet/requirements.txt:5), each restatement citing the TensorFlow source it follows:

* ``SAME`` padding of a strided convolution (tensorflow/core/framework/common_shape_fns.cc,
  GetWindowedOutputSizeVerbose): out = ceil(in / stride), total = max((out - 1) * stride + k - in, 0),
  before = total // 2, after = total - before (so a 3x3 / stride-2 layer on an even extent pads 0 rows in
  front and 1 behind);
* ``tf.contrib.layers.batch_norm`` (tensorflow/contrib/layers/python/layers/layers.py): defaults
  decay 0.999, center=True, scale=False, epsilon 0.001; variables beta / moving_mean / moving_variance
  under ``<scope>/BatchNorm``; inference = (x - moving_mean) * rsqrt(moving_variance + eps) + beta, training
  = the batch's mean and BIASED variance over every axis but the last;
* ``tf.contrib.layers.convolution2d`` / ``fully_connected`` (same file): variables ``weights``
  ([kh, kw, cin, cout] / [cin, cout]) and -- only when there is no normalizer and a biases_initializer --
  ``biases``; order: linear op, then normalizer OR bias, then activation;
* ``tf.image.resize_images(..., BILINEAR, align_corners=True)`` (tensorflow/core/kernels/image/
  resize_bilinear_op.cc with half_pixel_centers=false): scale = (in - 1) / (out - 1), src = dst * scale,
  taps floor(src) and min(floor(src) + 1, in - 1);
* ``tf.variable_scope`` prefixes variable names, ``tf.name_scope`` does not; creating a variable that exists
  without ``reuse`` is an error.

Arithmetic runs in float64 on float32-valued variables (TensorFlow's would be float32: the golden vectors are
therefore at least as close to the exact graph as TensorFlow's own output; the consumers hold 1e-4).
Python 2 behaviours the reference relies on (``sz / 2`` on an integer shape tensor, models.py:259) are
mirrored: ``/`` on an integer ``Tensor`` is the integer division Python 2's ``__div__`` -> ``tf.div`` was.

Nothing under ``hdrnet_amd/`` imports this package.
"""
import collections
import contextlib

import numpy as np

float32 = np.float32
float64 = np.float64
int32 = np.int32

COMPUTE = np.float64


class _Shape(list):
    def as_list(self):
        return list(self)


class Tensor(np.ndarray):
    """An ndarray with the two Tensor methods the reference calls."""

    def get_shape(self):
        return _Shape(int(s) for s in self.shape)

    def _int_div(self, other):
        if np.issubdtype(self.dtype, np.integer) and np.issubdtype(np.asarray(other).dtype, np.integer):
            return np.floor_divide(self, other)
        return np.true_divide(self, other)

    __truediv__ = _int_div
    __div__ = _int_div


def _t(x):
    return np.asarray(x).view(Tensor)


def _f(x):
    a = np.asarray(x)
    return a.astype(COMPUTE) if np.issubdtype(a.dtype, np.floating) else a


class GraphKeys(object):
    WEIGHTS = "weights"
    BIASES = "biases"
    ACTIVATIONS = "activations"
    MOVING_AVERAGE_VARIABLES = "moving_average_variables"
    UPDATE_OPS = "update_ops"


class _State(object):
    def __init__(self):
        self.reset(0)

    def reset(self, seed):
        self.variables = collections.OrderedDict()   # full name (no ':0') -> float32 ndarray
        self.scope = []
        self.reuse = False
        self.collections = collections.defaultdict(list)
        self.rng = np.random.RandomState(seed)
        self.created_in_pass = []


_STATE = _State()


def reset_default_graph():
    _STATE.reset(0)


def set_random_seed(seed):
    _STATE.rng = np.random.RandomState(seed)


def add_to_collection(name, value):
    _STATE.collections[name].append(value)


def get_collection(name):
    return list(_STATE.collections[name])


class _VariableScope(object):
    def reuse_variables(self):
        _STATE.reuse = True


def get_variable_scope():
    return _VariableScope()


@contextlib.contextmanager
def variable_scope(name_or_scope, default_name=None, reuse=None):
    name = name_or_scope if name_or_scope is not None else default_name
    if not isinstance(name, str) or not name:
        raise ValueError("variable_scope needs a name")
    _STATE.scope.append(name)
    try:
        yield _VariableScope()
    finally:
        _STATE.scope.pop()


@contextlib.contextmanager
def name_scope(name=None, default_name=None, values=None):
    yield name          # op names only: variable names are not affected


@contextlib.contextmanager
def device(name):
    yield


def constant_initializer(value=0, dtype=float32):
    def init(shape, dtype=float32):
        return np.full(shape, value, dtype=np.float32)
    return init


def zeros_initializer(dtype=float32):
    return constant_initializer(0.0)


def ones_initializer(dtype=float32):
    return constant_initializer(1.0)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True, collections=None):
    full = "/".join(_STATE.scope + [name])
    if full in _STATE.variables:
        if not _STATE.reuse:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % full)
        v = _STATE.variables[full]
        if shape is not None and tuple(int(s) for s in shape) != v.shape:
            raise ValueError("Trying to share variable %s, but specified shape %s and found shape %s."
                             % (full, tuple(shape), v.shape))
        return _t(_f(v))
    if _STATE.reuse:
        raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % full)
    if initializer is None:
        raise ValueError("the shim needs an explicit initializer for %s" % full)
    if callable(initializer):
        if shape is None:
            raise ValueError("Shape of a new variable (%s) must be fully defined." % full)
        value = initializer([int(s) for s in shape], dtype or float32)
    else:
        value = np.asarray(initializer)
        if shape is not None and tuple(int(s) for s in shape) != value.shape:
            raise ValueError("initializer shape does not match for %s" % full)
    _STATE.variables[full] = np.ascontiguousarray(value, dtype=np.float32)
    _STATE.created_in_pass.append(full)
    return _t(_f(_STATE.variables[full]))


def global_variables():
    V = collections.namedtuple("Variable", "name value")
    return [V(k + ":0", v) for k, v in _STATE.variables.items()]


# ---- plain tensor ops -----------------------------------------------------------------------------
def shape(x, name=None):
    return _t(np.asarray(np.asarray(x).shape, dtype=np.int32))


def _ints(s):
    return [int(v) for v in np.asarray(s).reshape(-1).tolist()] if not isinstance(s, (list, tuple)) \
        else [int(np.asarray(v)) for v in s]


def reshape(tensor, shape, name=None):   # noqa: A002
    return _t(np.reshape(np.asarray(tensor), _ints(shape)))


def stack(values, axis=0, name=None):
    return _t(np.stack([np.asarray(v) for v in values], axis=axis))


def unstack(value, num=None, axis=0, name=None):
    a = np.asarray(value)
    if num is not None and num != a.shape[axis]:
        raise ValueError("unstack: num does not match the axis")
    return [_t(np.take(a, i, axis=axis)) for i in range(a.shape[axis])]


def split(value, num_or_size_splits, axis=0, name=None):
    a = np.asarray(value)
    if isinstance(num_or_size_splits, (list, tuple)):
        idx = np.cumsum(num_or_size_splits)[:-1]
        return [_t(p) for p in np.split(a, idx, axis=axis)]
    if a.shape[axis] % int(num_or_size_splits):
        raise ValueError("split: dimension %d is not divisible by %d" % (a.shape[axis], num_or_size_splits))
    return [_t(p) for p in np.split(a, int(num_or_size_splits), axis=axis)]


def concat(values, axis, name=None):
    return _t(np.concatenate([np.asarray(v) for v in values], axis=axis))


def expand_dims(input, axis=None, name=None, dim=None):   # noqa: A002
    return _t(np.expand_dims(np.asarray(input), axis if axis is not None else dim))


def squeeze(input, axis=None, name=None, squeeze_dims=None):   # noqa: A002
    ax = axis if axis is not None else squeeze_dims
    return _t(np.squeeze(np.asarray(input), axis=None if ax is None else tuple(ax)))


def reduce_sum(input_tensor, axis=None, keep_dims=False, name=None, reduction_indices=None):
    ax = axis if axis is not None else reduction_indices
    return _t(np.sum(_f(input_tensor), axis=None if ax is None else tuple(np.atleast_1d(ax)), keepdims=keep_dims))


def reduce_mean(input_tensor, axis=None, keep_dims=False, name=None, reduction_indices=None):
    ax = axis if axis is not None else reduction_indices
    return _t(np.mean(_f(input_tensor), axis=None if ax is None else tuple(np.atleast_1d(ax)), keepdims=keep_dims))


def square(x, name=None):
    return _t(np.square(_f(x)))


def log(x, name=None):
    return _t(np.log(_f(x)))


def matmul(a, b, name=None):
    return _t(_f(a) @ _f(b))


def add(x, y, name=None):
    return _t(_f(x) + _f(y))


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return _t(np.clip(_f(t), clip_value_min, clip_value_max))


def zeros_like(tensor, dtype=None, name=None):
    return _t(np.zeros_like(np.asarray(tensor)))


class nn(object):
    @staticmethod
    def relu(features, name=None):
        return _t(np.maximum(_f(features), 0.0))

    @staticmethod
    def sigmoid(x, name=None):
        return _t(1.0 / (1.0 + np.exp(-_f(x))))

    @staticmethod
    def bias_add(value, bias, data_format=None, name=None):
        b = _f(bias)
        if b.ndim != 1 or b.shape[0] != np.asarray(value).shape[-1]:
            raise ValueError("bias_add: bias must be 1-D and match the last dimension")
        return _t(_f(value) + b)


# ---- tf.image -------------------------------------------------------------------------------------
def _resize_axis(a, out, axis, align_corners):
    n = a.shape[axis]
    if align_corners and out > 1:
        scale = (n - 1) / float(out - 1)
    else:
        scale = n / float(out)
    src = np.arange(out, dtype=np.float64) * scale
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, n - 1)
    w = src - lo
    shp = [1] * a.ndim
    shp[axis] = out
    w = w.reshape(shp)
    return np.take(a, lo, axis=axis) * (1.0 - w) + np.take(a, hi, axis=axis) * w


class image(object):
    class ResizeMethod(object):
        BILINEAR = 0

    @staticmethod
    def resize_images(images, size, method=0, align_corners=False):
        if method != image.ResizeMethod.BILINEAR:
            raise NotImplementedError("the shim restates the bilinear resize only")
        sz = np.asarray(size)
        if not np.issubdtype(sz.dtype, np.integer):
            raise TypeError("resize_images: size must be an int32 tensor (got %s)" % sz.dtype)
        h, w = int(sz[0]), int(sz[1])
        a = _f(images)
        if a.ndim != 4:
            raise ValueError("resize_images: 4-D NHWC in this shim")
        a = _resize_axis(a, h, 1, align_corners)   # rows, then columns: the two lerps commute exactly in
        a = _resize_axis(a, w, 2, align_corners)   # real arithmetic; TensorFlow does x inside y
        return _t(a)


# ---- tf.contrib.layers ----------------------------------------------------------------------------
def _same_pad(n, k, stride):
    out = -(-n // stride)
    total = max((out - 1) * stride + k - n, 0)
    return out, total // 2, total - total // 2


def _conv2d_same(x, w, stride):
    """NHWC x [B, H, W, Cin], HWIO w [kh, kw, Cin, Cout], padding SAME."""
    B, H, W, Cin = x.shape
    kh, kw, wc, Cout = w.shape
    if wc != Cin:
        raise ValueError("conv2d: input has %d channels, the filter expects %d" % (Cin, wc))
    oh, pt, pb = _same_pad(H, kh, stride)
    ow, pl, pr = _same_pad(W, kw, stride)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    out = np.zeros((B, oh, ow, Cout), dtype=COMPUTE)
    for ky in range(kh):
        for kx in range(kw):
            tap = xp[:, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride, :]
            out += tap @ w[ky, kx]
    return out


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


class _Layers(object):
    @staticmethod
    def variance_scaling_initializer(factor=2.0, mode="FAN_IN", uniform=False, seed=None, dtype=float32):
        if mode != "FAN_IN" or uniform:
            raise NotImplementedError
        def init(shape, dtype=float32):
            fan_in = float(np.prod(shape[:-1])) if len(shape) > 1 else float(shape[0])
            std = np.sqrt(1.3 * factor / fan_in)          # TensorFlow's truncated normal, corrected stddev
            return np.clip(_STATE.rng.standard_normal(shape), -2.0, 2.0).astype(np.float32) * np.float32(std)
        return init

    @staticmethod
    def batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, activation_fn=None,
                   param_initializers=None, updates_collections=GraphKeys.UPDATE_OPS, is_training=True,
                   reuse=None, variables_collections=None, outputs_collections=None, trainable=True, scope=None):
        x = _f(inputs)
        c = x.shape[-1]
        with variable_scope(scope, default_name="BatchNorm"):
            beta = get_variable("beta", [c], initializer=zeros_initializer()) if center else 0.0
            gamma = get_variable("gamma", [c], initializer=ones_initializer()) if scale else 1.0
            mean = get_variable("moving_mean", [c], initializer=zeros_initializer())
            var = get_variable("moving_variance", [c], initializer=ones_initializer())
        if is_training:
            axes = tuple(range(x.ndim - 1))
            mean, var = x.mean(axis=axes), x.var(axis=axes)        # biased variance (tf.nn.moments)
        out = (x - mean) / np.sqrt(var + epsilon) * gamma + beta
        if activation_fn is not None:
            out = activation_fn(out)
        return _t(out)

    @staticmethod
    def _finish(out, num_outputs, normalizer_fn, normalizer_params, biases_initializer, activation_fn):
        if normalizer_fn is not None:
            out = normalizer_fn(out, **(normalizer_params or {}))
        elif biases_initializer is not None:
            out = _f(out) + get_variable("biases", [num_outputs], initializer=biases_initializer)
        if activation_fn is not None:
            out = activation_fn(out)
        return _t(out)

    @staticmethod
    def convolution2d(inputs, num_outputs, kernel_size, stride=1, padding="SAME", data_format=None, rate=1,
                      activation_fn=nn.relu, normalizer_fn=None, normalizer_params=None,
                      weights_initializer=None, weights_regularizer=None, biases_initializer=zeros_initializer(),
                      biases_regularizer=None, reuse=None, variables_collections=None, outputs_collections=None,
                      trainable=True, scope=None):
        if padding != "SAME" or _pair(rate) != (1, 1) or data_format not in (None, "NHWC"):
            raise NotImplementedError("the shim restates SAME / rate 1 / NHWC only")
        x = _f(inputs)
        kh, kw = _pair(kernel_size)
        sh, sw = _pair(stride)
        if sh != sw:
            raise NotImplementedError
        with variable_scope(scope, default_name="Conv"):
            w = get_variable("weights", [kh, kw, x.shape[-1], num_outputs], initializer=weights_initializer)
            out = _conv2d_same(x, np.asarray(w), sh)
            return _Layers._finish(out, num_outputs, normalizer_fn, normalizer_params, biases_initializer,
                                   activation_fn)

    conv2d = convolution2d

    @staticmethod
    def fully_connected(inputs, num_outputs, activation_fn=nn.relu, normalizer_fn=None, normalizer_params=None,
                        weights_initializer=None, weights_regularizer=None, biases_initializer=zeros_initializer(),
                        biases_regularizer=None, reuse=None, variables_collections=None, outputs_collections=None,
                        trainable=True, scope=None):
        x = _f(inputs)
        with variable_scope(scope, default_name="fully_connected"):
            w = get_variable("weights", [x.shape[-1], num_outputs], initializer=weights_initializer)
            out = x @ np.asarray(w)
            return _Layers._finish(out, num_outputs, normalizer_fn, normalizer_params, biases_initializer,
                                   activation_fn)


class contrib(object):
    layers = _Layers
