// The pixel core of the row-segment forward kernels (apply_fwd_seg.hip, apply_fwd_io.hip): the padded,
// y-pre-lerped LDS coefficient image of a row segment and the per-pixel gather / blend / affine from it.
// See apply_fwd_seg.hip's header comment for the layout and its rationale.
#pragma once

#include <hip/hip_runtime.h>

#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace rows {

// x-only terms of a pixel (bilateral_slice_apply.cc:41,46,53-54,61-62).
struct XTerm {
  float wx0, wx1;
  int xbp;  // byte offset of (column gx0, plane 0 + 1) in the image
};

__device__ __forceinline__ XTerm x_term(float xf, float scale_x, int cmin, int colb, int cb) {
#pragma clang fp contract(off)
  XTerm t;
  const float gxf = mul_rn(xf, scale_x);
  const float fxl = floorf(gxf - 0.5f);
  const float dx0 = (fxl + 0.5f) - gxf;  // in (-1, 0]
  t.wx0 = 1.0f + dx0;
  t.wx1 = -dx0;
  t.xbp = __mul24((int)fxl - cmin, colb) + cb;
  return t;
}

// One pixel: z terms, the four-vector blend from the padded image, the affine
// (bilateral_slice_apply.cc:43-80).
template <int CIN, int COUT, bool OFFSET>
__device__ __forceinline__ void seg_pixel(const float* __restrict__ img, float gd_f, float zhi, int colb,
                                          const XTerm& xt, float g, const float (&in)[CIN > 0 ? CIN : 1],
                                          float (&out)[COUT]) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  f32x2 w0, w1;
  int a0;
  {
#pragma clang fp contract(off)
    const float gzf = mul_rn(g, gd_f);
    const float fzl = floorf(gzf - 0.5f);
    // corner centres as the reference forms them, (float)gz + 0.5f with gz1 = gz0 + 1: identical to
    // fzl + 1.5f while gzf is exact, and the same rounding as the reference once it is not
    const f32x2 cz = {fzl + 0.5f, (fzl + 1.0f) + 0.5f};
    const f32x2 gz2 = {gzf, gzf};
    const f32x2 dz = cz - gz2;  // (gz0 + .5) - gzf, (gz0 + 1.5) - gzf
    const f32x2 eps2 = {kSmoothEps, kSmoothEps};
    const f32x2 q = __builtin_elementwise_fma(dz, dz, eps2);
    const f32x2 s = {__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y)};
    const f32x2 one2 = {1.0f, 1.0f};
    // max(., 0) as the reference (numerics.h:108-113): never binds for a guide whose gzf is exact
    // in f32, but once |guide * GD| reaches 2^23 the rounding of (gz0 + 1.5) - gzf can make a corner
    // offset 2 and its un-clamped weight -1
    const f32x2 zero2 = {0.0f, 0.0f};
    const f32x2 wz = __builtin_elementwise_max(one2 - s, zero2);
    const f32x2 wx0 = {xt.wx0, xt.wx0}, wx1 = {xt.wx1, xt.wx1};
    w0 = wx0 * wz;
    w1 = wx1 * wz;
    // plane of z index iz is iz + 1; the clamp to [-1, GD-1] only guards wild guides (the
    // padded planes already hold the reference's clamped reads; v_med3 of a NaN yields -1).
    const int iz = (int)__builtin_amdgcn_fmed3f(fzl, -1.0f, zhi);
    a0 = __mul24(iz, CB) + xt.xbp;
  }
  CoefVec<C> coef;
  accum_vec<C, true>(coef, img, a0, w0.x);
  accum_vec<C, false>(coef, img, a0 + CB, w0.y);
  accum_vec<C, false>(coef, img, a0 + colb, w1.x);
  accum_vec<C, false>(coef, img, a0 + colb + CB, w1.y);
#pragma unroll
  for (int i = 0; i < COUT; ++i) {
    float v = OFFSET ? coef.get(i * CJ + CIN) : 0.0f;
#pragma unroll
    for (int j = 0; j < CIN; ++j) v = fmaf(coef.get(i * CJ + j), in[j], v);
    out[i] = v;
  }
}

// ---- the LEAN pixel phase (round 3) ---------------------------------------------------------------------
// The same gather / blend / affine with the per-pixel bookkeeping cut down (the pixel phase,
// not the memory system, is what follows the shader clock in a slow episode):
//   * x: wx1 = fract(gxf - .5) (one v_fract_f32) instead of floor / +.5 / subtract; equal to the
//     reference's -((gx0 + .5) - gxf) exactly for gxf >= .5 and to 1 ulp (6e-8) in the first half cell,
//     where gxf - .5 itself rounds.  wx0 is never formed: w(x0, z) = wz - wz * wx1 (one fma).
//   * byte addresses are carried as FLOATS (exact: < 2^24): xbp = fma(floor, colb, base), a0 =
//     cvt(fma(clamp(floor z), CB, xbp)) -- no int multiplies, one conversion per pixel instead of two.
//   * max(1 - s, 0) is the clamp modifier of the subtraction (1 - s <= 1 always).
// Everything else -- the rounded products gxf / gzf, the reference's corner-centre expressions for z,
// v_sqrt_f32 -- is as in seg_pixel above; weights differ from it by <= 1 ulp.
struct XTermLean {
  float wx1;   // weight of column gx0 + 1; column gx0 gets 1 - wx1
  float xbpf;  // byte offset of (column gx0, plane 0 + 1) in the image, as a float
};

__device__ __forceinline__ XTermLean x_term_lean(float xf, float scale_x, float colb_f, float xbase_f) {
#pragma clang fp contract(off)
  XTermLean t;
  const float gxf = mul_rn(xf, scale_x);
  const float tx = gxf - 0.5f;
  t.wx1 = __builtin_amdgcn_fractf(tx);
  t.xbpf = __builtin_fmaf(floorf(tx), colb_f, xbase_f);
  return t;
}

// PK: blend on 2-wide vectors (v_pk_fma_f32) or on scalars (v_fma_f32).  PKZ: the same choice for the two z taps.
template <int CIN, int COUT, bool OFFSET, bool PK, bool PKZ = true>
__device__ __forceinline__ void seg_pixel_lean(const float* __restrict__ img, float gd_f, float zhi, int colb,
                                               const XTermLean& xt, float g, const float (&in)[CIN > 0 ? CIN : 1],
                                               float (&out)[COUT]) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  float w00, w01, w10, w11;
  int a0;
  {
#pragma clang fp contract(off)
    const float gzf = mul_rn(g, gd_f);
    const float fzl = floorf(gzf - 0.5f);
    // corner centres as the reference forms them: (float)gz + 0.5f with gz1 = gz0 + 1
    const float c0 = fzl + 0.5f, c1 = (fzl + 1.0f) + 0.5f;
    if constexpr (PKZ) {  // the two taps as a 2-wide vector
      const f32x2 cz = {c0, c1};
      const f32x2 gz2 = {gzf, gzf};
      const f32x2 dz = cz - gz2;
      const f32x2 eps2 = {kSmoothEps, kSmoothEps};
      const f32x2 q = __builtin_elementwise_fma(dz, dz, eps2);
      const float s0 = __builtin_amdgcn_sqrtf(q.x), s1 = __builtin_amdgcn_sqrtf(q.y);
      // max(1 - s, 0) (numerics.h:108-113) == clamp(1 - s) to [0, 1] since s > 0: folds into the subtraction
      const f32x2 wz = {__builtin_amdgcn_fmed3f(1.0f - s0, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(1.0f - s1, 0.0f, 1.0f)};
      const f32x2 wx1 = {xt.wx1, xt.wx1};
      const f32x2 w1 = wz * wx1;
      const f32x2 w0 = __builtin_elementwise_fma(-wz, wx1, wz);  // wz * (1 - wx1), one rounding
      w00 = w0.x; w01 = w0.y; w10 = w1.x; w11 = w1.y;
    } else {
      const float dz0 = c0 - gzf, dz1 = c1 - gzf;
      const float s0 = __builtin_amdgcn_sqrtf(__builtin_fmaf(dz0, dz0, kSmoothEps));
      const float s1 = __builtin_amdgcn_sqrtf(__builtin_fmaf(dz1, dz1, kSmoothEps));
      const float wz0 = __builtin_amdgcn_fmed3f(1.0f - s0, 0.0f, 1.0f);
      const float wz1 = __builtin_amdgcn_fmed3f(1.0f - s1, 0.0f, 1.0f);
      w10 = wz0 * xt.wx1;
      w11 = wz1 * xt.wx1;
      w00 = __builtin_fmaf(-wz0, xt.wx1, wz0);  // wz0 * (1 - wx1), one rounding
      w01 = __builtin_fmaf(-wz1, xt.wx1, wz1);
    }
    const float izf = __builtin_amdgcn_fmed3f(fzl, -1.0f, zhi);  // wild guides only (NaN -> -1)
    a0 = (int)__builtin_fmaf(izf, (float)CB, xt.xbpf);
  }
  float o[COUT];
  if constexpr (PK) {
    CoefVec<C> coef;
    accum_vec<C, true>(coef, img, a0, w00);
    accum_vec<C, false>(coef, img, a0 + CB, w01);
    accum_vec<C, false>(coef, img, a0 + colb, w10);
    accum_vec<C, false>(coef, img, a0 + colb + CB, w11);
#pragma unroll
    for (int i = 0; i < COUT; ++i) {
      float v = OFFSET ? coef.get(i * CJ + CIN) : 0.0f;
#pragma unroll
      for (int j = 0; j < CIN; ++j) v = fmaf(coef.get(i * CJ + j), in[j], v);
      o[i] = v;
    }
  } else {
    float coef[C];
    const char* base = reinterpret_cast<const char*>(img) + a0;
    const float w[4] = {w00, w01, w10, w11};
    const int off[4] = {0, CB, colb, colb + CB};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if constexpr (C % 4 == 0) {
        const f32x4* p4 = reinterpret_cast<const f32x4*>(base + off[v]);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) {
          const f32x4 t = p4[q];
#pragma unroll
          for (int e = 0; e < 4; ++e) coef[4 * q + e] = v == 0 ? w[0] * t[e] : fmaf(w[v], t[e], coef[4 * q + e]);
        }
      } else {
        const float* p1 = reinterpret_cast<const float*>(base + off[v]);
#pragma unroll
        for (int q = 0; q < C; ++q) coef[q] = v == 0 ? w[0] * p1[q] : fmaf(w[v], p1[q], coef[q]);
      }
    }
#pragma unroll
    for (int i = 0; i < COUT; ++i) {
      float v = OFFSET ? coef[i * CJ + CIN] : 0.0f;
#pragma unroll
      for (int j = 0; j < CIN; ++j) v = fmaf(coef[i * CJ + j], in[j], v);
      o[i] = v;
    }
  }
#pragma unroll
  for (int i = 0; i < COUT; ++i) out[i] = o[i];
}

// Blend the two grid rows image row y needs into the padded LDS image (see the header comment):
//   img[j][p][c] = wy0 * grid[gy0c][clamp(cmin + j)][clamp(p - 1)][c] + wy1 * grid[gy1c][...]
// Work item = one VEC-float element of a source (column, plane) vector (one per thread at 4K; a
// rolled loop keeps the kernel at <= 64 VGPRs, i.e. 8 waves per SIMD, for every load flavour).
// IN_SCALE (wire formats, apply_fwd_io.hip): the image is staged with the coefficients that MULTIPLY AN INPUT CHANNEL
// (columns j < 3 of every 3 x 4 affine; C = 12, one float4 = one output row) scaled by `in_scale` = 1 / white level, so
// that a pixel enters the affine as its raw integer sample: coef * (1 / wl) * v instead of coef * (v / wl), one
// multiply per staged element instead of a division per sample.
template <int C, bool IN_SCALE = false>
__device__ __forceinline__ void stage_image(float* __restrict__ img, const float* __restrict__ grid_b,
                                            int y, int cmin, int ncols, int GH, int GW, int GD,
                                            float scale_y, float inv_col, int tid, int nthreads, float in_scale = 1.0f) {
  static_assert(!IN_SCALE || C == 12, "IN_SCALE: 3 -> 3 with offset, a float4 per output row");
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  constexpr int CV = C / VEC;
  typedef float elem_t __attribute__((ext_vector_type(VEC)));
  // Wave-uniform y terms (bilateral_slice_apply.cc:42,47,55-56).
  const float gyf = mul_rn(y + 0.5f, scale_y);
  const int gy0 = floor_to_int(gyf - 0.5f);
  const float wy0 = tent_weight(gy0 + 0.5f, gyf);
  const float wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
  const int gy0c = clamp_index(gy0, 0, GH - 1);
  const int gy1c = clamp_index(gy0 + 1, 0, GH - 1);
  const unsigned row_floats = (unsigned)(GW * GD * C);  // one image of the grid is < 2^31 floats
  const elem_t* r0 = reinterpret_cast<const elem_t*>(grid_b + (unsigned)gy0c * row_floats);
  const elem_t* r1 = reinterpret_cast<const elem_t*>(grid_b + (unsigned)gy1c * row_floats);
  elem_t* d = reinterpret_cast<elem_t*>(img);
  const int per_col = GD * CV;
  const int n = ncols * per_col;
  auto locate = [&](int e, int& j, int& rem) {
    j = (int)(((float)e + 0.5f) * inv_col);  // e / per_col, exact for e < 2^20
    rem = e - j * per_col;
    return min(max(cmin + j, 0), GW - 1) * per_col + rem;
  };
  auto put = [&](int e, int j, int rem, elem_t v) {
    if constexpr (IN_SCALE) {
      v[0] *= in_scale;
      v[1] *= in_scale;
      v[2] *= in_scale;
    }
    const int dst = e + CV * (2 * j + 1);  // column j has GD + 2 planes; source plane z is plane z + 1
    d[dst] = v;
    if (rem < CV) d[dst - CV] = v;             // z = 0      -> also plane 0
    if (rem >= per_col - CV) d[dst + CV] = v;  // z = GD - 1 -> also plane GD + 1
  };
  if (n <= nthreads) {  // uniform.  One element per thread: every frame of BASELINE.json at luma_bins <= 8
    for (int e = tid; e < n; e += nthreads) {
      int j, rem;
      const int src = locate(e, j, rem);
      put(e, j, rem, wy0 * r0[src] + wy1 * r1[src]);
    }
  } else {
    // Deeper grids (luma_bins = 16: 288 elements for 192 threads at 4K): two elements per trip, all four L2 reads in
    // flight together -- the rolled loop above pays the L2 latency once per trip, and the staging sits on every
    // workgroup's critical path ahead of the barrier (round 6: 41.8 -> ?? us at 4K, GD = 16).
    for (int e = tid; e < n; e += 2 * nthreads) {
      const int e1 = min(e + nthreads, n - 1);  // (a clamped second element re-stages the last one: same value)
      int j0, rem0, j1, rem1;
      const int s0 = locate(e, j0, rem0), s1 = locate(e1, j1, rem1);
      const elem_t a0 = r0[s0], b0 = r1[s0], a1 = r0[s1], b1 = r1[s1];
      put(e, j0, rem0, wy0 * a0 + wy1 * b0);
      put(e1, j1, rem1, wy0 * a1 + wy1 * b1);
    }
  }
}


// Grid columns a segment [xs, xe) touches, unclamped: gx0 of the first pixel .. gx0 + 1 of the last.
struct SegCols {
  int cmin, ncols;
};

__device__ __forceinline__ SegCols seg_cols(int xs, int xe, float scale_x) {
  const int cmin = floor_to_int(mul_rn(xs + 0.5f, scale_x) - 0.5f);
  const int cmax = floor_to_int(mul_rn(xe - 1 + 0.5f, scale_x) - 0.5f) + 1;
  return SegCols{cmin, cmax - cmin + 1};
}

// The same two numbers for every segment of a row, computed ONCE on the host and passed in the kernel
// arguments (they depend on the segment only; on the device they are ~14 wave-uniform VALU instructions
// per wave): packed (cmin + 1) | ncols << 16.  Rows of more than kSegTab segments compute them on the device.
constexpr int kSegTab = 8;
struct SegTab {
  unsigned packed[kSegTab];
  int n;  // 0: no table
};

inline SegTab make_seg_tab(int W, int seg, int nseg, float scale_x) {
  SegTab t{};
  if (nseg > kSegTab) return t;
  for (int i = 0; i < nseg; ++i) {
    const int xs = i * seg, xe = (xs + seg < W) ? xs + seg : W;
    // the device's expressions, each product rounded to float before use (volatile: no contraction)
    volatile float p0 = ((float)xs + 0.5f) * scale_x;
    volatile float p1 = ((float)(xe - 1) + 0.5f) * scale_x;
    const int cmin = (int)floorf(p0 - 0.5f), cmax = (int)floorf(p1 - 0.5f) + 1;
    t.packed[i] = (unsigned)(cmin + 1) | ((unsigned)(cmax - cmin + 1) << 16);
  }
  t.n = nseg;
  return t;
}

__device__ __forceinline__ SegCols seg_cols_tab(const SegTab& t, int seg_index, int xs, int xe, float scale_x) {
  if (t.n > 0) {  // uniform
    const unsigned v = t.packed[seg_index];
    return SegCols{(int)(v & 0xffffu) - 1, (int)(v >> 16)};
  }
  return seg_cols(xs, xe, scale_x);
}

}  // namespace rows
}  // namespace hdrnet_amd
