// Device-side numerics of the bilateral-grid path (gfx950).
//
// Semantics follow hdrnet/ops/numerics.h of the reference:
//   tent_weight           <- LerpWeight             numerics.h:53-57
//   mirror_index          <- MirrorBoundary         numerics.h:72-80
//   smoothed_tent_weight  <- SmoothedLerpWeight     numerics.h:108-113 (eps 1e-8:
//                            peak is 1-sqrt(1e-8) = 0.9999, NOT 1)
//   smoothed_tent_grad    <- SmoothedLerpWeightGrad numerics.h:116-126
//
// sqrtf and '/' are IEEE correctly rounded here (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt, no fast-math), f32 denormals are
// kept (hipcc default), so with FP contraction off these return bit-for-bit
// what the reference's host code returns.
#pragma once

#include <hip/hip_runtime.h>

namespace hdrnet_amd {

constexpr float kSmoothEps = 1.0e-8f;

// std::max(a, b) semantics (returns a when a is NaN), as the reference uses.
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

__device__ __forceinline__ int clamp_index(int v, int lo, int hi) {
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}

__device__ __forceinline__ float tent_weight(float x, float xs) {
  const float dx = x - xs;
  return std_max(1.0f - fabsf(dx), 0.0f);
}

__device__ __forceinline__ int mirror_index(int x, int extent) {
  if (x < 0) return -x - 1;
  if (x >= extent) return 2 * extent - 1 - x;
  return x;
}

__device__ __forceinline__ float smoothed_abs(float x) {
  return sqrtf(x * x + kSmoothEps);
}

__device__ __forceinline__ float smoothed_tent_weight(float x, float xs) {
  const float dx = x - xs;
  return std_max(1.0f - smoothed_abs(dx), 0.0f);
}

__device__ __forceinline__ float smoothed_tent_grad(float x, float xs) {
  const float dx = x - xs;
  const float a = smoothed_abs(dx);
  return (a > 1.0f) ? 0.0f : dx / a;
}

// a * b rounded to f32 and NOT available for FMA contraction into its consumers.  The
// reference forms gxf = (x+.5)*scale_x, gyf, gzf as f32 values first; contracting them into
// fma(x+.5, scale, -.5) or fma(-(x+.5), scale, g0+.5) shifts a weight by up to ulp(gxf)/2
// (measured: 1.2e-5 output error at GH = 64).  Note __fmul_rn() does NOT stop the
// contraction pass; the pragma (honoured under hipcc's -ffp-contract=fast-honor-pragmas)
// drops the `contract` flag from this multiply.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  const float p = a * b;
  return p;
}

// floor(v) as int with the conversion saturating (v_cvt_i32_f32 saturates and
// maps NaN to 0), so a wild guide value cannot index out of bounds once clamped.
__device__ __forceinline__ int floor_to_int(float v) { return (int)floorf(v); }

}  // namespace hdrnet_amd
