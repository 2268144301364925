"""Host-side mirror of the reference's operator interface: names, argument order, shape
rules and error behaviour (hdrnet/hdrnet_ops.py:27-48; bilateral_slice_apply_op.cc:147-193;
bilateral_slice_op.cc:129-147; hdrnet/layers.py:99-148).  Runs without a GPU."""
import inspect

import numpy as np
import pytest
import torch

from hdrnet_amd import hdrnet_ops as ops
from hdrnet_amd import layers


def test_signatures_match_reference():
    assert list(inspect.signature(ops.bilateral_slice).parameters)[:2] == ["grid", "guide"]
    p = list(inspect.signature(ops.bilateral_slice_apply).parameters)
    assert p[:4] == ["grid", "guide", "input", "has_offset"]
    # has_offset is a REQUIRED attr of the reference op (bilateral_slice_apply_op.cc:386)
    assert inspect.signature(ops.bilateral_slice_apply).parameters["has_offset"].default is inspect.Parameter.empty
    p = inspect.signature(layers.bilateral_slice_apply).parameters
    assert list(p) == ["grid", "guide", "input_image", "has_offset", "name"]
    assert p["has_offset"].default is True  # hdrnet/layers.py:125
    assert list(inspect.signature(layers.bilateral_slice).parameters) == ["grid", "guide", "name"]


def _t(*shape):
    return torch.rand(*shape)


@pytest.mark.parametrize("grid,guide,inp,ho,msg", [
    (_t(1, 4, 4, 4), _t(1, 8, 8), _t(1, 8, 8, 3), True, "grid should be 5D"),
    (_t(1, 4, 4, 4, 12), _t(1, 8, 8, 1), _t(1, 8, 8, 3), True, "Guide image should be 3D"),
    (_t(1, 4, 4, 4, 12), _t(1, 8, 8), _t(1, 8, 8), True, "Input image should be 4D"),
    (_t(1, 4, 4, 4, 12), _t(1, 8, 8), _t(1, 8, 9, 3), True, "Input and guide size should match"),
    (_t(2, 4, 4, 4, 12), _t(1, 8, 8), _t(1, 8, 8, 3), True, "Batch sizes should match"),
    (_t(1, 4, 4, 4, 13), _t(1, 8, 8), _t(1, 8, 8, 3), True, "with affine offset"),
    (_t(1, 4, 4, 4, 10), _t(1, 8, 8), _t(1, 8, 8, 3), False, "without affine offset"),
])
def test_apply_shape_rules(grid, guide, inp, ho, msg):
    with pytest.raises(ValueError, match=msg):
        ops.bilateral_slice_apply(grid, guide, inp, has_offset=ho)


def test_slice_shape_rules():
    with pytest.raises(ValueError, match="Grid should be 5D"):
        ops.bilateral_slice(_t(1, 4, 4, 4), _t(1, 8, 8))
    with pytest.raises(ValueError, match="Guide image should be 3D"):
        ops.bilateral_slice(_t(1, 4, 4, 4, 2), _t(1, 8, 8, 1))
    with pytest.raises(ValueError, match="Batch sizes"):
        ops.bilateral_slice(_t(2, 4, 4, 4, 2), _t(1, 8, 8))


def test_dtype_rule():
    with pytest.raises(TypeError, match="float32"):
        ops.bilateral_slice(_t(1, 4, 4, 4, 2).double(), _t(1, 8, 8).double())


def test_no_cpu_path():
    """The product has no CPU / eager fallback: well-formed CPU tensors are refused loudly."""
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.bilateral_slice_apply(_t(1, 4, 4, 4, 12), _t(1, 8, 8), _t(1, 8, 8, 3), has_offset=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.bilateral_slice(_t(1, 4, 4, 4, 2), _t(1, 8, 8))


def test_layers_6d_grid_routes_through_shape_checks():
    # 6-D [B,GH,GW,GD,n_out,n_in] is flattened to 5-D before the op (layers.py:139-144): a
    # mismatching n_in must be reported by the op's channel rule.
    with pytest.raises(ValueError, match="with affine offset"):
        layers.bilateral_slice_apply(_t(1, 4, 4, 4, 3, 5), _t(1, 8, 8), _t(1, 8, 8, 3), has_offset=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        layers.bilateral_slice_apply(_t(1, 4, 4, 4, 3, 4), _t(1, 8, 8), _t(1, 8, 8, 3), has_offset=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        layers.bilateral_slice(_t(1, 4, 4, 4, 3, 4), _t(1, 8, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from hdrnet_amd import _lib, build
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(build, "LIB_PATH", str(tmp_path / "nope" / "libhdrnet_amd.so"))
    monkeypatch.setattr(build, "build", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no hipcc")))
    with pytest.raises(_lib.HdrnetLibraryError):
        _lib.load()


def test_fused_ops_shape_rules_and_no_cpu_path():
    """The guide-network / pyramid ops share the slice-apply shape rules and refuse CPU tensors."""
    g, x = _t(1, 4, 4, 4, 12), _t(1, 8, 8, 3)
    c1, c2 = _t(16, 4), _t(17)
    with pytest.raises(ValueError, match=r"guide_conv1 should be \[n, Cin \+ 1\]"):
        ops.bilateral_slice_apply_nnguide(g, x, _t(16, 3), c2)
    with pytest.raises(ValueError, match=r"guide_conv2 should be \[n \+ 1\]"):
        ops.bilateral_slice_apply_nnguide(g, x, c1, _t(16))
    with pytest.raises(ValueError, match="with affine offset"):
        ops.bilateral_slice_apply_nnguide(_t(1, 4, 4, 4, 13), x, c1, c2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.bilateral_slice_apply_nnguide(g, x, c1, c2)
    with pytest.raises(RuntimeError, match="no CPU path"):  # also through autograd
        ops.bilateral_slice_apply_nnguide(g.requires_grad_(True), x, c1, c2)
    with pytest.raises(ValueError, match="either guide or"):
        ops.bilateral_slice_apply_upadd(g, x, _t(1, 4, 4, 3))
    with pytest.raises(ValueError, match="either guide or"):
        ops.bilateral_slice_apply_upadd(g, x, _t(1, 4, 4, 3), guide=_t(1, 8, 8), guide_conv1=c1, guide_conv2=c2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.bilateral_slice_apply_upadd(g.detach(), x, _t(1, 4, 4, 3), guide=_t(1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.resize_bilinear(x, 4, 4)
    with pytest.raises(ValueError, match="4D"):
        ops.resize_bilinear(_t(8, 8, 3), 4, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.input_moments(x)
    with pytest.raises(TypeError, match="float32"):
        ops.input_moments(x.double())


def test_three_instruction_white_level_division_is_ieee_exact():
    """apply_fwd_io.hip::div_white forms v / white_level as q = v * r; e = fma(-q, wl, v); q' = fma(e, r, q)
    with r = RN(1 / wl) from the host, claiming the result of the IEEE division TensorFlow performs
    (hdrnet/data_pipeline.py:202-232, 267-274: tf.to_float(im) / 255, 65535, 32767).  Checked here in exact
    rational arithmetic (each fma rounded once), against numpy's IEEE float32 division."""
    from fractions import Fraction
    import numpy as np
    f32 = np.float32

    def rn32(fr):  # round-to-nearest-even of an exact rational to float32 (normal range)
        if fr == 0:
            return f32(0)
        c = f32(float(fr))
        cands = [c, np.nextafter(c, f32(np.inf)), np.nextafter(c, f32(-np.inf))]
        return min(cands, key=lambda x: (abs(Fraction(float(x)) - fr), int(f32(x).view(np.uint32)) & 1))

    for wl, stride in ((255.0, 1), (65535.0, 13), (32767.0, 11), (1023.0, 97), (100.5, 97), (3.3333, 97)):
        wl32 = f32(wl)
        r = f32(1.0) / wl32
        W, R = Fraction(float(wl32)), Fraction(float(r))
        top = 256 if wl == 255.0 else 65536
        for v in range(0, top, stride):
            q = Fraction(float(rn32(v * R)))
            e = Fraction(float(rn32(v - q * W)))
            assert rn32(q + e * R) == f32(v) / wl32, (wl, v)


# ---- curves guide as a sorted piecewise-linear lookup (apply_fwd_io.hip, round 4) ---------------------------
def _curve_tables(shifts, slopes):
    """numpy twin of curves_build_tables for ONE channel: (tree[16], leaf[16][3]) from <= 16 knots -- rank by
    counting with index tie-break, float64 prefix sums rounded once, Eytzinger tree over the sorted knots 1..15."""
    n = len(shifts)
    s = np.full(16, np.inf, np.float32)
    sl = np.zeros(16, np.float32)
    s[:n], sl[:n] = shifts, slopes
    rank = np.array([sum((s[j] < s[k]) or (s[j] == s[k] and j < k) for j in range(16)) for k in range(16)])
    assert sorted(rank) == list(range(16))
    ss, ll = np.empty(16, np.float32), np.empty(16, np.float32)
    ss[rank], ll[rank] = s, sl
    tree = np.full(16, np.nan, np.float32)
    leaf = np.zeros((16, 3), np.float32)
    for i in range(16):
        if not ss[i] < np.inf:
            leaf[i] = (np.inf, 0, 0)
        else:
            cv = sum(float(ll[r]) * (float(ss[i]) - float(ss[r])) for r in range(i))
            leaf[i] = (ss[i], np.float32(cv), np.float32(sum(float(ll[r]) for r in range(i + 1))))
        if i >= 1:
            t = (i & -i).bit_length() - 1  # ctz
            tree[(1 << (3 - t)) + (i >> (t + 1))] = ss[i]
    assert not np.isnan(tree[1:]).any()
    return tree, leaf


def _curve_lookup(tree, leaf, v):
    v = np.asarray(v, np.float32)
    m = np.ones(v.shape, np.int64)
    for _ in range(4):
        m = 2 * m + (v >= tree[m])
    lf = leaf[m - 16]
    with np.errstate(invalid="ignore"):
        d = np.maximum(v - lf[..., 0], np.float32(0))
    return (lf[..., 2] * d + lf[..., 1]).astype(np.float32)


@pytest.mark.parametrize("case", ["reference-like", "random", "ties", "reversed", "few-knots", "one-knot", "wide"])
def test_curves_lookup_tables_equal_the_knot_scan(case):
    """The table form of the curves guide (sorted knots, value + slope per interval, 4-step search) against the
    reference's knot-by-knot sum (hdrnet/models.py:157-188 as oracle.curves_guide restates it) -- every interval,
    every knot itself, both sides of every knot, far outside; unsorted / tied / fewer knots."""
    import oracle
    rng = np.random.default_rng(sum(map(ord, case)))
    n = {"few-knots": 5, "one-knot": 1}.get(case, 16)
    if case == "reference-like":
        shifts = np.tile(np.linspace(0, 1, n, endpoint=False)[:, None], (1, 3)) + 0.01 * rng.standard_normal((n, 3))
        slopes = 0.3 * rng.standard_normal((n, 3))
        slopes[0] += 1.0
    elif case == "ties":
        shifts = rng.choice([0.1, 0.25, 0.25, 0.5, 0.5, 0.5, 0.9], (n, 3))
        slopes = rng.standard_normal((n, 3))
    elif case == "reversed":
        shifts = np.tile(np.linspace(1, 0, n)[:, None], (1, 3))
        slopes = rng.standard_normal((n, 3))
    elif case == "wide":
        shifts = 10 * rng.standard_normal((n, 3))
        slopes = 3 * rng.standard_normal((n, 3))
    else:
        shifts = rng.random((n, 3))
        slopes = rng.standard_normal((n, 3))
    shifts, slopes = shifts.astype(np.float32), slopes.astype(np.float32)
    ccm = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)  # identity: t = the sample itself
    lo, hi = float(shifts.min()) - 1, float(shifts.max()) + 1
    for c in range(3):
        tree, leaf = _curve_tables(shifts[:, c], slopes[:, c])
        ks = shifts[:, c]
        v = np.concatenate([rng.uniform(lo, hi, 4000).astype(np.float32), ks, np.nextafter(ks, np.float32(-np.inf)),
                            np.nextafter(ks, np.float32(np.inf)), np.float32([lo - 100, hi + 100])])
        got = _curve_lookup(tree, leaf, v)
        # the reference's sum, float32 in knot order (what oracle.curves_guide does per channel) and in float64
        want32 = (slopes[:, c] * np.maximum(v[:, None] - ks, np.float32(0))).sum(-1, dtype=np.float32)
        want64 = (slopes[:, c].astype(np.float64) * np.maximum(v[:, None].astype(np.float64) - ks, 0)).sum(-1)
        scale = np.abs(slopes[:, c].astype(np.float64) * np.maximum(v[:, None].astype(np.float64) - ks, 0)).sum(-1) + 1e-30
        # one rounding of value + slope term against the exact sum; the float32 scan itself is up to ~n ulp away
        assert np.all(np.abs(got - want64) <= 3 * 2.0 ** -24 * np.maximum(scale, np.abs(want64)) + 1e-30), case
        assert np.all(np.abs(got - want64) <= np.abs(want32 - want64) + 3 * 2.0 ** -24 * scale + 1e-30), case
    # and through the whole guide formula, at the GPU test's tolerance
    x = rng.random((2000, 3)).astype(np.float32)
    ccm2 = (ccm + 0.2 * rng.standard_normal((3, 4))).astype(np.float32)
    mix = np.array([0.4, 0.35, 0.25, 0.02], np.float32)
    if case in ("reference-like", "random", "few-knots"):
        t = (x @ ccm2[:, :-1].T + ccm2[:, -1]).astype(np.float32)
        cv = np.stack([_curve_lookup(*_curve_tables(shifts[:, c], slopes[:, c]), t[:, c]) for c in range(3)], -1)
        g = np.clip((cv @ mix[:-1] + mix[-1]).astype(np.float32), 0, 1)
        np.testing.assert_allclose(g, oracle.curves_guide(x, ccm2, shifts, slopes, mix), rtol=0, atol=2e-6)


def test_build_sweeps_what_a_killed_build_left_behind(tmp_path, monkeypatch):
    """A killed build never reaches its `finally`: per-process objects `*.o.<pid>`, `*.tmp` link outputs and
    `.digest.<pid>` files stay in the tree (24 stale objects, 5 MB).  The next build removes them under
    the build lock -- and nothing else."""
    from hdrnet_amd import build as hb
    lib = tmp_path / "lib"
    obj = lib / "obj"
    obj.mkdir(parents=True)
    stale = [obj / "capi.o.4242", obj / "apply_fwd_seg.o.17", lib / "libhdrnet_amd.so.4242.tmp", lib / "libhdrnet_amd.so.digest.4242"]
    keep = [lib / "libhdrnet_amd.so", lib / "libhdrnet_amd.so.digest", obj / "notes.txt"]
    for f in stale + keep:
        f.write_text("x")
    monkeypatch.setattr(hb, "LIB_DIR", str(lib))
    hb._sweep_stale(str(obj))
    assert not any(f.exists() for f in stale)
    assert all(f.exists() for f in keep)
