// Generic (runtime-shape) kernels of the bilateral-grid path for gfx950.
//
// One thread per output element group, any Cin / Cout / has_offset / grid and
// image extents.  These are the fall-back for shapes the LDS-staged
// specialisations (apply_fwd_rows.hip, ...) do not cover and the bit-exact
// cross-check of the fast kernels: this translation unit is compiled with
// -ffp-contract=off and evaluates every sum in the reference's order, so its
// results equal the reference CPU op's bit for bit
// (hdrnet/ops/bilateral_slice_apply.cc:24-259, bilateral_slice.cc:25-168).
//
// Unlike the reference CUDA kernels (one thread per output CHANNEL, weights and
// 96 sqrt recomputed per term -- bilateral_slice_apply.cu.cc:83-122) a thread
// here owns a whole pixel: the 8 corner weights are formed once (2 sqrt) and
// reused for every (i, j).
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"

namespace hdrnet_amd {
namespace {

constexpr int kThreads = 256;

struct Corners {
  float w[8];   // (wx*wy)*wz in the reference's gy, gx, gz loop order
  int off[8];   // element offset of grid[b, gyc, gxc, gzc, 0] relative to grid[b]
};

// Everything the reference computes per pixel before touching the grid
// (bilateral_slice_apply.cc:41-48, :54-64).  DERIV selects wz' = GD * d(wz)/d(gzf)
// (bilateral_slice_apply.cc:186-187) instead of wz.
template <bool DERIV>
__device__ __forceinline__ Corners make_corners(int x, int y, float g, float scale_x,
                                                float scale_y, int GH, int GW, int GD,
                                                int C) {
  const float gxf = (x + 0.5f) * scale_x;
  const float gyf = (y + 0.5f) * scale_y;
  const float gzf = g * GD;
  const int gx0 = floor_to_int(gxf - 0.5f);
  const int gy0 = floor_to_int(gyf - 0.5f);
  const int gz0 = floor_to_int(gzf - 0.5f);
  Corners c;
  int k = 0;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int gy = gy0 + dy;
    const int gyc = clamp_index(gy, 0, GH - 1);
    const float wy = tent_weight(gy + 0.5f, gyf);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int gx = gx0 + dx;
      const int gxc = clamp_index(gx, 0, GW - 1);
      const float wx = tent_weight(gx + 0.5f, gxf);
#pragma unroll
      for (int dz = 0; dz < 2; ++dz) {
        const int gz = gz0 + dz;
        const int gzc = clamp_index(gz, 0, GD - 1);
        const float wz = DERIV ? GD * smoothed_tent_grad(gz + 0.5f, gzf)
                               : smoothed_tent_weight(gz + 0.5f, gzf);
        c.w[k] = wx * wy * wz;
        c.off[k] = ((gyc * GW + gxc) * GD + gzc) * C;
        ++k;
      }
    }
  }
  return c;
}

__device__ __forceinline__ float sample8(const Corners& c, const float* __restrict__ gb,
                                         int ch) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += c.w[k] * gb[c.off[k] + ch];
  return s;
}

// ---- BilateralSliceApply forward (bilateral_slice_apply.cc:24-82) -----------------
__global__ __launch_bounds__(kThreads) void apply_fwd_generic(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, float* __restrict__ out, long long npix, int H,
    int W, int GH, int GW, int GD, int Cin, int Cout, int Cj, float scale_x,
    float scale_y, int y0) {
  const long long p = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  const int C = Cout * Cj;
  const float* gb = grid + b * ((long long)GH * GW * GD * C);
  // y0: first frame row of a row-split launch (0 for whole frames); scale_y is GH / frame height
  const Corners c = make_corners<false>(x, y + y0, guide[p], scale_x, scale_y, GH, GW, GD, C);
  const float* in = input + p * Cin;
  float* o = out + p * Cout;
  for (int i = 0; i < Cout; ++i) {
    float value = 0.0f;
    for (int j = 0; j < Cj; ++j) {
      const float s = sample8(c, gb, i * Cj + j);
      if (j < Cin) {
        value += s * in[j];
      } else {
        value += s;
      }
    }
    o[i] = value;
  }
}

// ---- BilateralSliceApply guide VJP (bilateral_slice_apply.cc:140-206) --------------
__global__ __launch_bounds__(kThreads) void apply_guide_grad_generic(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, const float* __restrict__ dout,
    float* __restrict__ dguide, long long npix, int H, int W, int GH, int GW, int GD,
    int Cin, int Cout, int Cj, float scale_x, float scale_y) {
  const long long p = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  const int C = Cout * Cj;
  const float* gb = grid + b * ((long long)GH * GW * GD * C);
  const Corners c = make_corners<true>(x, y, guide[p], scale_x, scale_y, GH, GW, GD, C);
  const float* in = input + p * Cin;
  const float* d = dout + p * Cout;
  float vjp = 0.0f;
  for (int i = 0; i < Cout; ++i) {
    float grad_value = 0.0f;
    for (int j = 0; j < Cj; ++j) {
      const float s = sample8(c, gb, i * Cj + j);
      const float input_value = (j < Cin) ? in[j] : 1.0f;
      grad_value += s * input_value;
    }
    vjp += grad_value * d[i];
  }
  dguide[p] = vjp;
}

// ---- BilateralSliceApply input VJP (bilateral_slice_apply.cc:208-259) --------------
__global__ __launch_bounds__(kThreads) void apply_input_grad_generic(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ dout, float* __restrict__ dinput, long long npix, int H,
    int W, int GH, int GW, int GD, int Cin, int Cout, int Cj, float scale_x,
    float scale_y) {
  const long long p = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  const int C = Cout * Cj;
  const float* gb = grid + b * ((long long)GH * GW * GD * C);
  const Corners c = make_corners<false>(x, y, guide[p], scale_x, scale_y, GH, GW, GD, C);
  const float* d = dout + p * Cout;
  float* di = dinput + p * Cin;
  for (int j = 0; j < Cin; ++j) {
    float vjp = 0.0f;
    for (int i = 0; i < Cout; ++i) {
      const float s = sample8(c, gb, i * Cj + j);
      vjp += s * d[i];
    }
    di[j] = vjp;
  }
}

// ---- grid VJP, gather form (bilateral_slice_apply.cc:84-138, bilateral_slice.cc:72-118)
// One thread per grid element, serial mirror-boundary gather over the +-1 cell
// pixel window exactly as the reference CPU code does it: deterministic and
// bit-exact, and as slow as the reference's GridGrad kernel.  Only used when
// HDRNET_KERNEL_GENERIC is forced or no faster variant applies.
// APPLY: element = (j, i), value = wx*wy*wz*in_j * dout_i
// !APPLY: element = c,     value = wz*wx*wy * dout_c   (operand order of bilateral_slice.cc:112)
template <bool APPLY>
__global__ __launch_bounds__(kThreads) void grid_grad_gather_generic(
    const float* __restrict__ guide, const float* __restrict__ input,
    const float* __restrict__ dout, float* __restrict__ dgrid, long long nelem, int H,
    int W, int GH, int GW, int GD, int Cin, int Cout, int Cj, float scale_x,
    float scale_y) {
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= nelem) return;
  const int C = Cout * Cj;  // !APPLY: Cj == 1, Cout == C
  const int ch = (int)(e % C);
  const int gz = (int)((e / C) % GD);
  const int gx = (int)((e / ((long long)C * GD)) % GW);
  const int gy = (int)((e / ((long long)C * GD * GW)) % GH);
  const long long b = e / ((long long)C * GD * GW * GH);
  const int i = ch / Cj;
  const int j = ch % Cj;
  const int x0 = floor_to_int(scale_x * (gx + 0.5f - 1.0f));
  const int x1 = (int)ceilf(scale_x * (gx + 0.5f + 1.0f));
  const int y0 = floor_to_int(scale_y * (gy + 0.5f - 1.0f));
  const int y1 = (int)ceilf(scale_y * (gy + 0.5f + 1.0f));
  const long long pb = b * H * (long long)W;
  float vjp = 0.0f;
  for (int y = y0; y < y1; ++y) {
    const int ym = mirror_index(y, H);
    const float gyf = (y + 0.5f) / scale_y;
    const float wy = tent_weight(gy + 0.5f, gyf);
    for (int x = x0; x < x1; ++x) {
      const int xm = mirror_index(x, W);
      const float gxf = (x + 0.5f) / scale_x;
      const float wx = tent_weight(gx + 0.5f, gxf);
      const long long p = pb + (long long)ym * W + xm;
      const float gzf = guide[p] * GD;
      float wz = smoothed_tent_weight(gz + 0.5f, gzf);
      if ((gz == 0 && gzf < 0.5f) || (gz == GD - 1 && gzf > GD - 0.5f)) wz = 1.0f;
      if (APPLY) {
        const float input_value = (j < Cin) ? input[p * Cin + j] : 1.0f;
        const float grad_value = wx * wy * wz * input_value;
        vjp += grad_value * dout[p * Cout + i];
      } else {
        vjp += wz * wx * wy * dout[p * C + ch];
      }
    }
  }
  dgrid[e] = vjp;
}

// ---- BilateralSlice forward (bilateral_slice.cc:25-70) -----------------------------
__global__ __launch_bounds__(kThreads) void slice_fwd_generic(
    const float* __restrict__ grid, const float* __restrict__ guide,
    float* __restrict__ out, long long npix, int H, int W, int GH, int GW, int GD, int C,
    float scale_x, float scale_y) {
  const long long p = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  const float* gb = grid + b * ((long long)GH * GW * GD * C);
  const Corners c = make_corners<false>(x, y, guide[p], scale_x, scale_y, GH, GW, GD, C);
  float* o = out + p * C;
  for (int ch = 0; ch < C; ++ch) o[ch] = sample8(c, gb, ch);
}

// ---- BilateralSlice guide VJP (bilateral_slice.cc:120-168) -------------------------
__global__ __launch_bounds__(kThreads) void slice_guide_grad_generic(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ dout, float* __restrict__ dguide, long long npix, int H,
    int W, int GH, int GW, int GD, int C, float scale_x, float scale_y) {
  const long long p = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (p >= npix) return;
  const int x = (int)(p % W);
  const int y = (int)((p / W) % H);
  const long long b = p / ((long long)W * H);
  const float* gb = grid + b * ((long long)GH * GW * GD * C);
  const Corners c = make_corners<true>(x, y, guide[p], scale_x, scale_y, GH, GW, GD, C);
  const float* d = dout + p * C;
  float vjp = 0.0f;
  for (int ch = 0; ch < C; ++ch) vjp += sample8(c, gb, ch) * d[ch];
  dguide[p] = vjp;
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + kThreads - 1) / kThreads); }

}  // namespace

// ---- launchers -----------------------------------------------------------------------
hipError_t launch_apply_fwd_generic(const ApplyArgs& a, hipStream_t s) {
  const long long npix = (long long)a.B * a.H * a.W;
  apply_fwd_generic<<<blocks_for(npix), kThreads, 0, s>>>(
      a.grid, a.guide, a.input, a.out, npix, a.H, a.W, a.GH, a.GW, a.GD, a.Cin, a.Cout,
      a.Cj, (float)a.GW / a.W, (float)a.GH / a.frame_rows(), a.y0);
  return hipGetLastError();
}

hipError_t launch_apply_grad_generic(const ApplyGradArgs& a, hipStream_t s) {
  const long long npix = (long long)a.B * a.H * a.W;
  const float sx = (float)a.GW / a.W, sy = (float)a.GH / a.H;
  if (a.dgrid) {
    const long long nelem = (long long)a.B * a.GH * a.GW * a.GD * a.Cout * a.Cj;
    grid_grad_gather_generic<true><<<blocks_for(nelem), kThreads, 0, s>>>(
        a.guide, a.input, a.dout, a.dgrid, nelem, a.H, a.W, a.GH, a.GW, a.GD, a.Cin,
        a.Cout, a.Cj, (float)a.W / a.GW, (float)a.H / a.GH);
  }
  if (a.dguide) {
    apply_guide_grad_generic<<<blocks_for(npix), kThreads, 0, s>>>(
        a.grid, a.guide, a.input, a.dout, a.dguide, npix, a.H, a.W, a.GH, a.GW, a.GD,
        a.Cin, a.Cout, a.Cj, sx, sy);
  }
  if (a.dinput) {
    apply_input_grad_generic<<<blocks_for(npix), kThreads, 0, s>>>(
        a.grid, a.guide, a.dout, a.dinput, npix, a.H, a.W, a.GH, a.GW, a.GD, a.Cin,
        a.Cout, a.Cj, sx, sy);
  }
  return hipGetLastError();
}

hipError_t launch_slice_fwd_generic(const SliceArgs& a, hipStream_t s) {
  const long long npix = (long long)a.B * a.H * a.W;
  slice_fwd_generic<<<blocks_for(npix), kThreads, 0, s>>>(
      a.grid, a.guide, a.out, npix, a.H, a.W, a.GH, a.GW, a.GD, a.C, (float)a.GW / a.W,
      (float)a.GH / a.H);
  return hipGetLastError();
}

hipError_t launch_slice_grad_generic(const SliceGradArgs& a, hipStream_t s) {
  const long long npix = (long long)a.B * a.H * a.W;
  if (a.dgrid) {
    const long long nelem = (long long)a.B * a.GH * a.GW * a.GD * a.C;
    grid_grad_gather_generic<false><<<blocks_for(nelem), kThreads, 0, s>>>(
        a.guide, nullptr, a.dout, a.dgrid, nelem, a.H, a.W, a.GH, a.GW, a.GD, 0, a.C, 1,
        (float)a.W / a.GW, (float)a.H / a.GH);
  }
  if (a.dguide) {
    slice_guide_grad_generic<<<blocks_for(npix), kThreads, 0, s>>>(
        a.grid, a.guide, a.dout, a.dguide, npix, a.H, a.W, a.GH, a.GW, a.GD, a.C,
        (float)a.GW / a.W, (float)a.GH / a.H);
  }
  return hipGetLastError();
}

}  // namespace hdrnet_amd
