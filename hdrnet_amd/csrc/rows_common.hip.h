// Shared device code of the LDS-staged "row" kernels (forward: apply_fwd_rows.hip,
// backward: apply_bwd_rows.hip): the y-pre-lerped column image, its staging, the
// per-pixel slicing terms and the packed-FMA blend.  See apply_fwd_rows.hip for the
// design notes and the reference line numbers.
#pragma once

#include <hip/hip_runtime.h>

#include "numerics.hip.h"

namespace hdrnet_amd {
namespace rows {

constexpr int kPxPerThread = 4;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int C>
struct CoefVec {
  static constexpr int kPairs = (C + 1) / 2;
  f32x2 v[kPairs];  // coefficient c lives in v[c / 2][c % 2]
  __device__ __forceinline__ float get(int c) const { return v[c >> 1][c & 1]; }
};

// coef (+)= w * vec, vec = the C floats at byte offset `off` of the LDS image.
// Explicit 2-wide vectors: the blend maps onto v_pk_fma_f32 / v_pk_mul_f32 with the
// weight broadcast through op_sel and the ds_read_b128 result consumed in place.
template <int C, bool FIRST>
__device__ __forceinline__ void accum_vec(CoefVec<C>& coef, const float* __restrict__ colY,
                                          int off, float w) {
  const f32x2 w2 = {w, w};
  const char* base = reinterpret_cast<const char*>(colY) + off;
  if constexpr (C % 4 == 0) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base);
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const f32x4 t = p[q];
      if constexpr (FIRST) {
        coef.v[2 * q + 0] = w2 * t.xy;
        coef.v[2 * q + 1] = w2 * t.zw;
      } else {
        coef.v[2 * q + 0] = __builtin_elementwise_fma(w2, t.xy, coef.v[2 * q + 0]);
        coef.v[2 * q + 1] = __builtin_elementwise_fma(w2, t.zw, coef.v[2 * q + 1]);
      }
    }
  } else {
    const float* p = reinterpret_cast<const float*>(base);
#pragma unroll
    for (int q = 0; q < C; ++q)
      coef.v[q >> 1][q & 1] = FIRST ? w * p[q] : fmaf(w, p[q], coef.v[q >> 1][q & 1]);
  }
}

struct RowCtx {
  const float* colY;  // LDS image, [ncol][GD][C]
  float scale_x, gd_f;
  // Byte-space addressing of the LDS image: column stride and the clamp windows of
  // (gx - gxlo) * col_bytes and gz * vec_bytes.
  int gxlo, col_bytes, x_lo_b, x_hi_b, z_hi_b;
};

// Blend the two grid rows this image row needs into LDS; returns the row context.
// WAVE = false: the whole workgroup fills one image and meets at a barrier.
// WAVE = true : each wavefront fills its own private image; LDS operations of one
//               wave complete in order, so no s_barrier is needed -- waves never wait
//               for each other.
template <int C, bool WAVE>
__device__ __forceinline__ RowCtx stage_row(float* __restrict__ colY,
                                            const float* __restrict__ grid_b, int y, int xs,
                                            int xe, int GH, int GW, int GD, float scale_x,
                                            float scale_y) {
  // Wave-uniform y terms (bilateral_slice_apply.cc:42,47,55-56).
  const float gyf = mul_rn(y + 0.5f, scale_y);  // rounded product, as the reference
  const int gy0 = floor_to_int(gyf - 0.5f);
  const float wy0 = tent_weight(gy0 + 0.5f, gyf);
  const float wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
  const int gy0c = clamp_index(gy0, 0, GH - 1);
  const int gy1c = clamp_index(gy0 + 1, 0, GH - 1);
  // Grid columns touched by pixels [xs, xe).
  const int gxlo = clamp_index(floor_to_int(mul_rn(xs + 0.5f, scale_x) - 0.5f), 0, GW - 1);
  const int gxhi =
      clamp_index(floor_to_int(mul_rn(xe - 1 + 0.5f, scale_x) - 0.5f) + 1, 0, GW - 1);
  const int n = (gxhi - gxlo + 1) * GD * C;  // floats; contiguous in the grid row
  const float* r0 = grid_b + ((size_t)(gy0c * GW + gxlo) * GD) * C;
  const float* r1 = grid_b + ((size_t)(gy1c * GW + gxlo) * GD) * C;
  if constexpr (C % 4 == 0) {
    const float4* a4 = reinterpret_cast<const float4*>(r0);
    const float4* b4 = reinterpret_cast<const float4*>(r1);
    float4* d4 = reinterpret_cast<float4*>(colY);
    const int e0 = WAVE ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    const int estep = WAVE ? 64 : (int)blockDim.x;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int e = e0; e < n / 4; e += estep) {
      const float4 a = a4[e], b = b4[e];
      d4[e] = make_float4(wy0 * a.x + wy1 * b.x, wy0 * a.y + wy1 * b.y,
                          wy0 * a.z + wy1 * b.z, wy0 * a.w + wy1 * b.w);
    }
  } else {
    const int e0 = WAVE ? (int)(threadIdx.x & 63u) : (int)threadIdx.x;
    const int estep = WAVE ? 64 : (int)blockDim.x;
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
    for (int e = e0; e < n; e += estep) colY[e] = wy0 * r0[e] + wy1 * r1[e];
  }
  if constexpr (WAVE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
  const int col_bytes = GD * C * (int)sizeof(float);
  return RowCtx{colY,
                scale_x,
                (float)GD,
                gxlo,
                col_bytes,
                (0 - gxlo) * col_bytes,
                (GW - 1 - gxlo) * col_bytes,
                (GD - 1) * C * (int)sizeof(float)};
}

// Per-pixel slicing terms: the two x / z corner weights and the byte offsets of the four
// (gx, gz) coefficient vectors in the LDS image.
//
// Coordinates and weights follow bilateral_slice_apply.cc:41-64:
//   gxf = (x + .5) * scale_x, gx0 = floor(gxf - .5), dx0 = (gx0 + .5) - gxf  in (-1, 0]
//   gzf = guide * GD,         gz0 = floor(gzf - .5), dz0 = (gz0 + .5) - gzf  in (-1, 0]
//   wx0 = 1 - |dx0| = 1 + dx0,   wx1 = 1 - |dx0 + 1| = -dx0          (tent, numerics.h:53)
//   wz0 = 1 - sqrt(dz0^2 + eps), wz1 = 1 - sqrt((dz0 + 1)^2 + eps)  (smoothed, :108)
// floor() puts both corners within one cell of the sample, so the reference's
// max(., 0) never binds and is dropped; weights still come from the UNclamped
// corner and only the indices are clamped (in byte space).
// EXACT_SQRT = false: v_sqrt_f32 (1 ulp) stands in for the correctly-rounded
// expansion -- the argument lies in [1e-8, 1 + 1e-8], nowhere near a denormal, and one
// ulp of a weight is 6e-8 (forward kernel).  EXACT_SQRT = true: IEEE sqrtf; the guide
// VJP divides by it and compares it with 1 (numerics.h:116-126).
struct SliceTerms {
  float wx0, wx1, wz0, wz1;
  float dz0, dz1, sz0, sz1;  // z offsets of the two corners and their smoothed |.|
  int a00, a01, a10, a11;    // byte offsets: a<x><z>
};

template <int C, bool EXACT_SQRT>
__device__ __forceinline__ SliceTerms slice_terms(const RowCtx& r, float xf, float g) {
// No FMA contraction in the coordinate arithmetic: the reference rounds gxf / gzf to f32
// before using them (__fmul_rn alone does not stop the contraction pass).
#pragma clang fp contract(off)
  constexpr int kVecBytes = C * (int)sizeof(float);
  SliceTerms t;
  // __fmul_rn: the products must be ROUNDED to f32 before use, as in the reference; letting
  // the compiler contract them into the consumers (fma(xf, sx, -0.5), fma(-xf, sx, gx0+.5))
  // shifts the weights by up to ulp(gxf)/2 -- 1e-5 at GW = 256.
  const float gxf = mul_rn(xf, r.scale_x);  // xf = x + 0.5f, exact
  const float fxl = floorf(gxf - 0.5f);
  const float dx0 = (fxl + 0.5f) - gxf;
  t.wx0 = 1.0f + dx0;
  t.wx1 = -dx0;
  const float gzf = mul_rn(g, r.gd_f);
  const float fzl = floorf(gzf - 0.5f);
  t.dz0 = (fzl + 0.5f) - gzf;
  t.dz1 = ((fzl + 1.0f) + 0.5f) - gzf;  // the reference's (float)(gz0 + 1) + 0.5f - gzf.  NOT dz0 + 1: near a bin centre that loses the low bits of dz1,
                               // and the guide VJP's d wz/d gz has slope 1e4 there
  const float q0 = fmaf(t.dz0, t.dz0, kSmoothEps), q1 = fmaf(t.dz1, t.dz1, kSmoothEps);
  t.sz0 = EXACT_SQRT ? sqrtf(q0) : __builtin_amdgcn_sqrtf(q0);
  t.sz1 = EXACT_SQRT ? sqrtf(q1) : __builtin_amdgcn_sqrtf(q1);
  // max(., 0) as the reference (numerics.h:108-113): binds only for wild guides (|guide * GD| >= 2^23,
  // where the rounding of (gz0 + 1.5) - gzf can make a corner offset 2 and its weight -1)
  t.wz0 = fmaxf(1.0f - t.sz0, 0.0f);
  t.wz1 = fmaxf(1.0f - t.sz1, 0.0f);
  // floor of a wild guide is clamped in float before the int conversion so that the
  // byte-space arithmetic below cannot overflow (v_med3_f32).
  const int iz = (int)__builtin_amdgcn_fmed3f(fzl, -2.0f, r.gd_f + 1.0f);
  const int zb = __mul24(iz, kVecBytes);  // |iz| <= GD + 1: 24-bit multiply is exact
  const int zb0 = min(max(zb, 0), r.z_hi_b);
  const int zb1 = min(max(zb + kVecBytes, 0), r.z_hi_b);
  // x needs no guard: gxf in (0, GW) by construction, so gx0 in [-1, GW - 1].
  const int xb = __mul24((int)fxl - r.gxlo, r.col_bytes);
  const int xb0 = max(xb, r.x_lo_b);
  const int xb1 = min(xb + r.col_bytes, r.x_hi_b);
  t.a00 = xb0 + zb0;
  t.a01 = xb0 + zb1;
  t.a10 = xb1 + zb0;
  t.a11 = xb1 + zb1;
  return t;
}

// One pixel: slice the y-pre-lerped columns at (x, guide) and apply the affine
// (bilateral_slice_apply.cc:50-80).
template <int CIN, int COUT, bool OFFSET>
__device__ __forceinline__ void slice_apply_pixel(const RowCtx& r, float xf, float g,
                                                  const float (&in)[CIN],
                                                  float (&out)[COUT]) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  const SliceTerms t = slice_terms<C, false>(r, xf, g);
  CoefVec<C> coef;
  accum_vec<C, true>(coef, r.colY, t.a00, t.wx0 * t.wz0);
  accum_vec<C, false>(coef, r.colY, t.a01, t.wx0 * t.wz1);
  accum_vec<C, false>(coef, r.colY, t.a10, t.wx1 * t.wz0);
  accum_vec<C, false>(coef, r.colY, t.a11, t.wx1 * t.wz1);
  // :72-80 -- per-pixel (Cout x Cj) . [in; 1]
#pragma unroll
  for (int i = 0; i < COUT; ++i) {
    float v = OFFSET ? coef.get(i * CJ + CIN) : 0.0f;
#pragma unroll
    for (int j = 0; j < CIN; ++j) v = fmaf(coef.get(i * CJ + j), in[j], v);
    out[i] = v;
  }
}

// ---- per-pixel VJP core shared by apply_bwd_rows.hip and the fused backward (grid_grad_mfma.hip) ----
// Gather the four vectors once, accumulate with two weight sets.
template <int C, bool FIRST, bool WA, bool WB>
__device__ __forceinline__ void accum_vec2(CoefVec<C>& ca, CoefVec<C>& cb,
                                           const float* __restrict__ colY, int off, float wa,
                                           float wb) {
  const f32x2 wa2 = {wa, wa}, wb2 = {wb, wb};
  const char* base = reinterpret_cast<const char*>(colY) + off;
  if constexpr (C % 4 == 0) {
    const f32x4* p = reinterpret_cast<const f32x4*>(base);
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      const f32x4 t = p[q];
      if constexpr (WA) {
        ca.v[2 * q + 0] = FIRST ? wa2 * t.xy : __builtin_elementwise_fma(wa2, t.xy, ca.v[2 * q + 0]);
        ca.v[2 * q + 1] = FIRST ? wa2 * t.zw : __builtin_elementwise_fma(wa2, t.zw, ca.v[2 * q + 1]);
      }
      if constexpr (WB) {
        cb.v[2 * q + 0] = FIRST ? wb2 * t.xy : __builtin_elementwise_fma(wb2, t.xy, cb.v[2 * q + 0]);
        cb.v[2 * q + 1] = FIRST ? wb2 * t.zw : __builtin_elementwise_fma(wb2, t.zw, cb.v[2 * q + 1]);
      }
    }
  } else {
    const float* p = reinterpret_cast<const float*>(base);
#pragma unroll
    for (int q = 0; q < C; ++q) {
      const float t = p[q];
      if constexpr (WA) ca.v[q >> 1][q & 1] = FIRST ? wa * t : fmaf(wa, t, ca.v[q >> 1][q & 1]);
      if constexpr (WB) cb.v[q >> 1][q & 1] = FIRST ? wb * t : fmaf(wb, t, cb.v[q >> 1][q & 1]);
    }
  }
}

// The four (x corner, z tap) coefficient vectors at byte offsets a<x><z> of an LDS image, blended
// once with the tent weights (-> sliced coefficients A_ij, dinput_j = sum_i dout_i A_ij) and once
// with the z tent's derivative dw = GD * d wz / d gz (-> dA_ij, dguide = sum_i dout_i (sum_j dA_ij
// in_j + dA_i,offset)).  bilateral_slice_apply.cc:140-259; BilateralSlice: CIN = 0, CJ = 1.
template <int CIN, int COUT, bool OFFSET, bool WANT_GUIDE, bool WANT_INPUT>
__device__ __forceinline__ void vjp_blend(const float* __restrict__ img, int a00, int a01, int a10, int a11,
                                          float wx0, float wx1, float wz0, float wz1, float dw0, float dw1,
                                          const float* __restrict__ in,  // [CIN]
                                          const float* __restrict__ d,   // [COUT]
                                          float& dguide, float* __restrict__ dinput) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  CoefVec<C> A, dA;
  accum_vec2<C, true, WANT_INPUT, WANT_GUIDE>(A, dA, img, a00, wx0 * wz0, wx0 * dw0);
  accum_vec2<C, false, WANT_INPUT, WANT_GUIDE>(A, dA, img, a01, wx0 * wz1, wx0 * dw1);
  accum_vec2<C, false, WANT_INPUT, WANT_GUIDE>(A, dA, img, a10, wx1 * wz0, wx1 * dw0);
  accum_vec2<C, false, WANT_INPUT, WANT_GUIDE>(A, dA, img, a11, wx1 * wz1, wx1 * dw1);
  if constexpr (WANT_GUIDE) {
    float vjp = 0.0f;
#pragma unroll
    for (int i = 0; i < COUT; ++i) {
      float gv = OFFSET ? dA.get(i * CJ + CIN) : 0.0f;
#pragma unroll
      for (int j = 0; j < CIN; ++j) gv = fmaf(dA.get(i * CJ + j), in[j], gv);
      vjp = fmaf(gv, d[i], vjp);
    }
    dguide = vjp;
  }
  if constexpr (WANT_INPUT) {
#pragma unroll
    for (int j = 0; j < CIN; ++j) {
      float v = 0.0f;
#pragma unroll
      for (int i = 0; i < COUT; ++i) v = fmaf(A.get(i * CJ + j), d[i], v);
      dinput[j] = v;
    }
  }
}

// ---- buffer stores with a cache policy -------------------------------------------------------------
// Output streams are written exactly once and never re-read by the kernel.  Plain stores leave up
// to an L2's worth of dirty lines for the end-of-kernel write-back; `nt` (streaming) and `sc0 sc1`
// (write-through) stores drain during the kernel: worth 1.5 us per 4K frame and 0.7 us per 1080p
// frame on the forward (profiles/r02/exp2_store_policy_*.txt).  The two are equal where a
// workgroup's run is a whole number of 128-B lines (4K: 768 px x 12 B, 1080p: 960 px x 12 B), but
// write-through pays for every PARTIAL line at a run boundary -- 4000-px rows cut into 1000-px
// segments (12 000 B): 64.8 us write-through vs 58.7 us nt vs 63.7 us plain (profiles/r02/
// exp10_hdrp_flavours.txt) -- so `nt` is the policy the product kernels store with (the forward
// alone picks write-through per launch where every segment is whole lines, apply_fwd_seg.hip).  For
// ~2 ms after a kernel that left plain-store dirty lines behind, both run slower (write-through 45-47
// us, nt 42-43 us at 4K) and then settle (profiles/r02/exp25).  A raw buffer descriptor over exactly
// the run being written also drops out-of-range lanes in hardware (no predicate).
typedef int v4i32 __attribute__((ext_vector_type(4)));
constexpr int kAuxPlain = 0, kAuxNt = 2, kAuxSc1 = 16, kAuxSc0Sc1 = 17;
constexpr int kAuxStream = kAuxNt;  // the policy the product kernels store with

// Raw buffer descriptor over `bytes` bytes at `base` (gfx9 family: dword 3 = 0x00020000).  `base` and `bytes` are
// WAVE-UNIFORM at every call site (a row segment, a wave's run), and the instruction wants the descriptor in SGPRs -- but a
// 64-bit `row * W + x` is multiplied on the VALU (v_mad_u64_u32), so the compiler finds the descriptor in VGPRs and wraps
// every buffer store in a WATERFALL LOOP (4 v_readfirstlane + 2 v_cmp_eq_u64 + exec bookkeeping + a branch per store, three
// stores per wave in the forward: ~30 instructions in front of the stores; found in round 5 in the ISA of every kernel
// that stores through a descriptor).  Reading the three words through readfirstlane here says what the call sites
// guarantee: one v_readfirstlane per word, no loop; for a base the compiler already holds in SGPRs it folds away.
// The name says the contract: a per-lane `base` / `bytes` (or a call inside divergent control flow with differing
// values) would silently take the FIRST ACTIVE LANE's.  The tools build voids the descriptor on it, so that a violation fails
// the parity suite's tools-library tests instead of corrupting an output.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  int n = __builtin_amdgcn_readfirstlane((int)bytes);
#ifdef HDRNET_TOOLS_BUILD
  // a lane that disagrees with the first one voids the descriptor (zero records: loads return 0, stores are dropped), so
  // the kernel's output is visibly wrong instead of subtly.  No branch and no trap: a trap block behind every store took
  // the fused gradient kernels of the tools build into scratch (192 B) and 20 % off the product's time, which voided the
  // A/B timings this library exists for.
  if (__builtin_amdgcn_ballot_w64(((unsigned)a ^ lo) | ((unsigned)(a >> 32) ^ hi) | (bytes ^ (unsigned)n)) != 0ull) n = 0;
#endif
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

template <int AUX>
__device__ __forceinline__ void buf_store16(float4 v, __amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const v4i32 d = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, AUX);
}

// Streaming 16-B load of pixel data that is read exactly once (guide / input / dout): the `nt`
// policy.  Measured with the forward's byte volume and launch geometry and no compute
// (tools/debug/ubench/stream_cache_policy.hip): plain loads + plain stores 41.2 us, nontemporal
// loads + plain stores 36.2 us (6.4 TB/s = 80 % of 8 TB/s), nontemporal stores on top 36.9 us.  The
// builtin needs an ext-vector pointee to stay ONE global_load_dwordx4 (through a float4 struct it
// is scalarised into four dword loads -- the variant round 1 first measured as 8 % slower).
__device__ __forceinline__ float4 load_stream4(const float* __restrict__ p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Upper bound of grid columns `npx` consecutive pixels can touch: floor differences of gx0
// over npx-1 pixels (<= floor(d)+1), +1 for the upper neighbour, +1 for the count, +1
// slack for float rounding of the coordinates; never more than GW.
inline int max_cols_for(int npx, int GW, int W) {
  const long long cols = ((long long)(npx - 1) * GW) / W + 4;
  return (int)(cols < GW ? cols : GW);
}

// ---- fused guide network / pyramid up-add (shared by apply_fwd_seg.hip and apply_fwd_rows.hip) ----
// GUIDE_NN: the guide is not read from memory but computed per pixel from the input by the
// reference's point-wise guide network with batch-norm folded (HDRNetPointwiseNNGuide._guide,
// hdrnet/models.py:203-210; parameters in the layout hdrnet/bin/freeze_graph.py:170-184 exports):
//   guide = sigmoid(conv2[n] + sum_k conv2[k] * relu(conv1[k][CIN] + sum_j conv1[k][j] * in_j))
// -- the fusion the reference's own GL renderer performs (benchmark/assets/gpyrnn.frag:42-63).
// The guide never touches HBM (24 instead of 28 B/px) and the 16-channel full-resolution
// intermediate of the un-fused graph disappears.
//
struct GuideNN {
  const float* conv1;  // [n][CIN + 1]: weights then bias of feature k
  const float* conv2;  // [n + 1]: mixing weights then bias
  float* guide_out;    // optional [B][H][W] copy of the guide (null: not written)
  int n;
  // The sigmoid: false = tf.nn.sigmoid's form, expf + an IEEE divide (the reference's bits to ~1 ulp; what a
  // backward pass wants: the guide's VJP is steep near bin centres, and a 1-ulp guide moves dinput by 3e-4 of its
  // scale there); true = v_exp_f32 + v_rcp_f32 (<= 2 ulp of the guide, 1e-6 of the output's scale; 10 instead of
  // ~24 instructions per pixel, worth 9-11 % of these VALU-bound kernels).  Chosen by the CALLER
  // (HDRNET_GUIDE_SIGMOID_FAST in the flags of the ..._ex entry points) -- until round 5 it followed
  // `guide_out == NULL`, an implicit numerics switch.
  bool fast_sigmoid = false;
  // PRESCALED parameters (HDRNET_GUIDE_RELU_PRESCALED; CIN = 3; written by hdrnet_guide_nn_prescale_f32, guide_grad.hip):
  //   conv1 = [n][4] {w0, b, w1, w2} * 2^-e_k,  conv2 = [n + 1] {m_k * 2^e_k ..., bias},  2^e_k >= 2 (|b| + x_max sum_j |w_j|)
  // so that the hidden activation of feature k, scaled by the power of two, cannot exceed 1 for any input with
  // |x_j| <= x_max, and relu(h) 2^-e == clamp(h 2^-e, 0, 1) -- the CLAMP output modifier of the feature's last
  // v_pk_fma_f32 instead of two v_max_f32 per pixel pair.  Powers of two commute with every rounding of the chain
  // (outside the denormal range: |h| < 2^(e - 126)), so the guide is the un-scaled evaluation's bit for bit; with the
  // feature's {w0, b} in ONE aligned SGPR pair the chain's first FMA takes weight and bias from the same scalar operand
  // (one constant-bus read), which also removes the v_mov of the bias: 4 packed instructions per feature and pixel pair
  // instead of 4 packed + 2 v_max + 1 v_mov (208 -> 128 VALU per wave of 256 pixels).  Inputs beyond x_max saturate the
  // activation at 2^e_k instead of overflowing the bound: the caller's contract (include/hdrnet_amd.h).
  bool prescaled = false;
};

// One feature of a PRESCALED guide network (GuideNN::prescaled) on both pixel pairs of a lane: hv = w0 x0 + b (weight =
// the SGPR pair's low half, bias = its high half: one constant-bus operand), + w1 x1, + w2 x2 clamped to [0, 1];
// acc += m hv with m the low (HI = false) or high half of its SGPR pair.
template <bool HI>
__device__ __forceinline__ void nn_feature_prescaled(f32x2 wa, f32x2 wb, f32x2 mm, const f32x2 (&in2)[2][3], f32x2 (&acc2)[2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x2 hv;
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(hv) : "s"(wa), "v"(in2[h][0]));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(hv) : "s"(wb), "v"(in2[h][1]));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] clamp" : "+v"(hv) : "s"(wb), "v"(in2[h][2]));
    if constexpr (HI)
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc2[h]) : "s"(mm), "v"(hv));
    else
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc2[h]) : "s"(mm), "v"(hv));
  }
}

// Evaluated for a lane's 4 consecutive pixels at once (inf = [pixel][CIN] floats): one pass over the
// features, the weights read ONCE per feature through the constant address space (wave-uniform s_load; a plain global
// pointer next to the kernel's stores is not provably invariant and compiles to per-lane vector loads).
template <int CIN>
__device__ __forceinline__ void guide_nn_quad(const GuideNN& gn, const float* inf, float (&g)[kPxPerThread]) {
  typedef __attribute__((address_space(4))) const float cfloat;
  cfloat* c1 = (cfloat*)gn.conv1;
  cfloat* c2 = (cfloat*)gn.conv2;
  // pixels in pairs: the hidden layer is v_pk_fma_f32 on (pixel 0, 1) and (pixel 2, 3) with the weight
  // broadcast from an SGPR -- 6 packed FMAs per feature instead of 12 scalar ones (same fmaf chain per
  // pixel, so the results are those of the scalar form bit for bit)
  static_assert(kPxPerThread == 4, "two pixel pairs");
  const float bias = c2[gn.n];
  f32x2 in2[2][CIN];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < CIN; ++j) in2[h][j] = f32x2{inf[(2 * h) * CIN + j], inf[(2 * h + 1) * CIN + j]};
  }
  f32x2 acc2[2] = {f32x2{bias, bias}, f32x2{bias, bias}};
  bool done = false;
  if constexpr (CIN == 3) {
    if (gn.prescaled) {  // wave-uniform (GuideNN::prescaled)
      done = true;
      typedef __attribute__((address_space(4))) const f32x4 cfloat4;
      cfloat4* q1 = (cfloat4*)gn.conv1;
      int k = 0;
      for (; k + 4 <= gn.n; k += 4) {
        const f32x4 m4 = *(cfloat4*)(c2 + k);  // (conv2 is 16-B aligned: checked by the entry points)
        const f32x4 wq[4] = {q1[k], q1[k + 1], q1[k + 2], q1[k + 3]};
        nn_feature_prescaled<false>(f32x2{wq[0].x, wq[0].y}, f32x2{wq[0].z, wq[0].w}, f32x2{m4.x, m4.y}, in2, acc2);
        nn_feature_prescaled<true>(f32x2{wq[1].x, wq[1].y}, f32x2{wq[1].z, wq[1].w}, f32x2{m4.x, m4.y}, in2, acc2);
        nn_feature_prescaled<false>(f32x2{wq[2].x, wq[2].y}, f32x2{wq[2].z, wq[2].w}, f32x2{m4.z, m4.w}, in2, acc2);
        nn_feature_prescaled<true>(f32x2{wq[3].x, wq[3].y}, f32x2{wq[3].z, wq[3].w}, f32x2{m4.z, m4.w}, in2, acc2);
      }
      for (; k < gn.n; ++k) {
        const f32x4 w = q1[k];
        const float m = c2[k];
        nn_feature_prescaled<false>(f32x2{w.x, w.y}, f32x2{w.z, w.w}, f32x2{m, m}, in2, acc2);
      }
    }
  }
  if (!done) {
#pragma unroll 4
    for (int k = 0; k < gn.n; ++k) {
      float w[CIN + 1];
#pragma unroll
      for (int j = 0; j <= CIN; ++j) w[j] = c1[k * (CIN + 1) + j];
      const float m = c2[k];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2 hv = {w[CIN], w[CIN]};
#pragma unroll
        for (int j = 0; j < CIN; ++j) hv = __builtin_elementwise_fma(f32x2{w[j], w[j]}, in2[h][j], hv);
        const f32x2 r = {fmaxf(hv.x, 0.0f), fmaxf(hv.y, 0.0f)};
        acc2[h] = __builtin_elementwise_fma(f32x2{m, m}, r, acc2[h]);
      }
    }
  }
  const float acc[kPxPerThread] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y};
  if (!gn.fast_sigmoid) {  // wave-uniform: the reference's sigmoid (GuideNN::fast_sigmoid)
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q) g[q] = 1.0f / (1.0f + expf(-acc[q]));
  } else {
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q)
      g[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * acc[q]));
  }
}

// UPADD: out += the coarser pyramid level's output, bilinearly up-sampled with align_corners = True
// -- the `tf.image.resize_images(current, sz, BILINEAR, align_corners=True)` + `tf.add` of
// HDRNetGaussianPyrNN._output (hdrnet/models.py:283-287).  TensorFlow (requirements.txt:
// tensorflow_gpu==2.12.0; not vendored) computes, per axis, scale = (in - 1) / float(out - 1),
// src = i * scale, lower = floor(src), upper = min(ceil(src), in - 1), lerp = src - lower, and
// top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp
// (tensorflow/core/kernels/image/resize_bilinear_op.cc, legacy non-half-pixel path).
struct UpAdd {
  const float* coarse;  // [B][Hc][Wc][COUT]
  int Hc, Wc;
  float sh, sw;  // (Hc - 1) / float(H - 1), (Wc - 1) / float(W - 1)   (in / out when out == 1)
};

inline float resize_scale(int in, int out) {  // TF CalculateResizeScale, align_corners = true
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : (float)in / (float)out;
}

// of[k * COUT + i] += up-sampled coarse level at pixel (x + k, y), k = 0 .. 3.  Row terms are
// workgroup-uniform; the 2 x 2 x COUT gathers hit the (small, cache-resident) coarse level.
template <int COUT>
__device__ __forceinline__ void upadd_quad(const UpAdd& up, int b, int y, int x, float* of) {
  const float sy = mul_rn((float)y, up.sh);
  const float fy = floorf(sy);
  const float ly = sy - fy;
  const int y0 = (int)fy, y1 = min((int)ceilf(sy), up.Hc - 1);
  const float* r0 = up.coarse + ((size_t)b * up.Hc + y0) * up.Wc * COUT;
  const float* r1 = up.coarse + ((size_t)b * up.Hc + y1) * up.Wc * COUT;
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const float sxf = mul_rn((float)(x + k), up.sw);
    const float fx = floorf(sxf);
    const float lx = sxf - fx;
    const int x0 = (int)fx * COUT, x1 = min((int)ceilf(sxf), up.Wc - 1) * COUT;
#pragma unroll
    for (int i = 0; i < COUT; ++i) {
      const float tl = r0[x0 + i], tr = r0[x1 + i], bl = r1[x0 + i], br = r1[x1 + i];
      const float top = tl + (tr - tl) * lx;
      const float bot = bl + (br - bl) * lx;
      of[k * COUT + i] += top + (bot - top) * ly;
    }
  }
}


// How a row is cut into workgroup segments: `threads` lanes x 4 pixels per segment, the
// segment width balanced over the row (e.g. W = 3840 -> 5 segments of 768 px / 192 threads;
// W = 1920 -> 2 x 960 px / 256 threads with 16 idle lanes).
struct Plan {
  int threads, nseg, seg, max_cols;
  bool vec4;  // 16-B accesses usable: W % 4 == 0 and 16-B aligned buffers
};

inline Plan make_row_plan(int W, int GW, bool aligned16) {
  Plan best{};
  long long best_waste = -1;
  const int cands[3] = {256, 192, 128};
  for (int T : cands) {
    const int span = T * kPxPerThread;
    const int nseg = (W + span - 1) / span;
    const long long waste = (long long)nseg * span - W;
    if (best_waste < 0 || waste < best_waste) {
      best_waste = waste;
      best.threads = T;
      best.nseg = nseg;
    }
  }
  best.vec4 = aligned16 && (W % 4 == 0);
  best.seg = round_up((W + best.nseg - 1) / best.nseg, 4);
  // (Round 4: rounding the segments to 32-px multiples -- W = 4000 as 1024 / 1024 / 1024 / 928 instead of 4 x 1000,
  // so that every output run is whole 128-B lines and the forward may store write-through -- measured 61.3 vs
  // 60.7 us at 4000 x 3000, interleaved, the memory skeleton at 61.0: no gain, not kept.  profiles/r04/fwd_launch_shape.md)
  best.threads = round_up((best.seg + kPxPerThread - 1) / kPxPerThread, 64);
  if (best.threads > 256) best.threads = 256;
  best.max_cols = max_cols_for(best.seg, GW, W);
  return best;
}

// Compute units of the current device (cached per device ordinal; benign race).
inline int num_cus() {
  static int cache[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cache[dev] = n;
  }
  return cache[dev];
}

}  // namespace rows
}  // namespace hdrnet_amd
