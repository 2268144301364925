"""TEST INFRASTRUCTURE: float64 evaluation of the per-pixel VJPs of BilateralSliceApply.

dguide and dinput of hdrnet/ops/bilateral_slice_apply.cc:140-259 with every sum and product in float64.  The
coordinates and the tap offsets are formed in FLOAT32 exactly as the reference forms them ((x + .5) * (GW / W),
guide * GD, (float)gz + 0.5f - gzf: bilateral_slice_apply.cc:37-48, 163-186) -- they are part of the op's semantics,
not of its rounding noise.  Cin = Cout = 3 with offset.  Row-chunked, so that a 4K frame fits in memory.

Used by tests/ (and tools/dguide_noise_floor.py) to measure how far a float32 implementation -- the reference's own
arithmetic (the C oracle) or the HIP kernels -- is from the exact value of the reference's formulas.
"""
import numpy as np


def f64_vjps(grid, guide, inp, dout, rows_per_chunk=128):
    f32 = np.float32
    B, GH, GW, GD, _ = grid.shape
    _, H, W = guide.shape
    xs = (np.arange(W, dtype=f32) + f32(0.5)) * (f32(GW) / f32(W))
    ys_all = (np.arange(H, dtype=f32) + f32(0.5)) * (f32(GH) / f32(H))
    gx0 = np.floor(xs - f32(0.5)).astype(np.int64)
    G = grid.astype(np.float64).reshape(B, GH, GW, GD, 3, 4)
    eps = np.float64(np.float32(1e-8))
    dg = np.zeros((B, H, W))
    di = np.zeros((B, H, W, 3))
    for b in range(B):
        for r0 in range(0, H, rows_per_chunk):
            r1 = min(H, r0 + rows_per_chunk)
            ys = ys_all[r0:r1]
            gy0 = np.floor(ys - f32(0.5)).astype(np.int64)
            gzf = (guide[b, r0:r1] * f32(GD)).astype(f32)
            gz0 = np.floor(gzf - f32(0.5)).astype(np.int64)
            d64 = dout[b, r0:r1].astype(np.float64)
            inh = np.concatenate([inp[b, r0:r1].astype(np.float64), np.ones((r1 - r0, W, 1))], -1)
            for dy in (0, 1):
                gy = gy0 + dy
                wy = np.maximum(1 - np.abs((gy.astype(f32) + f32(0.5)) - ys).astype(np.float64), 0)
                gyc = np.clip(gy, 0, GH - 1)
                for dx in (0, 1):
                    gx = gx0 + dx
                    wx = np.maximum(1 - np.abs((gx.astype(f32) + f32(0.5)) - xs).astype(np.float64), 0)
                    gxc = np.clip(gx, 0, GW - 1)
                    w2 = wy[:, None] * wx[None, :]
                    for dz in (0, 1):
                        gz = gz0 + dz
                        d32 = (gz.astype(f32) + f32(0.5)) - gzf
                        d = d32.astype(np.float64)
                        s = np.sqrt(d * d + eps)
                        # numerics.h:116-126: the derivative is CUT to 0 where the smoothed |d| exceeds 1.  That decision
                        # is taken in float32 as the reference takes it (a tap exactly one cell away has
                        # sqrtf(1 + 1e-8f) == 1.0f: not cut, while sqrt(1 + 1e-8) > 1 in float64) -- like the floor of the
                        # coordinates it selects WHICH function is evaluated; the value is then formed in float64
                        cut = np.sqrt((d32 * d32).astype(f32) + f32(1e-8), dtype=f32) > f32(1)
                        dw = np.where(cut, 0.0, d / s) * GD     # x GD (:186)
                        wz = np.maximum(1 - s, 0)                # numerics.h:108-113
                        g = G[b][gyc[:, None], gxc[None, :], np.clip(gz, 0, GD - 1)]  # [rows, W, 3, 4]
                        dg[b, r0:r1] += np.einsum("hwij,hwj,hwi->hw", g, inh, d64) * (w2 * dw)
                        di[b, r0:r1] += np.einsum("hwij,hwi->hwj", g[..., :3], d64) * (w2 * wz)[..., None]
    return dg, di
