// BilateralSliceApply forward with the product's WIRE FORMATS fused in.
//
// Around the op the reference converts images on the host / in separate TF ops:
//   input  uint8 / uint16  ->  tf.to_float(im) / white_level      hdrnet/data_pipeline.py:202-232
//                                                                  (255 or 65535), :267-274 (HDR+:
//                                                                  32767); run.py:157-164
//   output float -> tf.cast(255 * clip(out, 0, 1), uint8)          hdrnet/bin/run.py:95
// Here both conversions happen in registers: a 4K RGB frame moves 3 (or 6) + 3 bytes per pixel
// instead of 12 + 12 (+ the conversion kernels' own traffic), and, with the guide network fused
// as well (GUIDE_NN, apply_fwd_rows.hip), nothing but the image itself touches HBM.
//
// Geometry and slicing code are those of apply_fwd_rows_vec4 (rows_common.hip.h).  A thread's 4
// pixels are 12 contiguous bytes of uint8 RGB -- per-lane dwordx3 accesses that are contiguous
// across the wave -- so the quantised paths need no LDS transpose; a float output still goes
// through it.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <cstdio>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

// Guide sources: 0 = a [B][H][W] map in memory; 1 = the folded point-wise guide network
// (HDRNetPointwiseNNGuide._guide, hdrnet/models.py:203-210); 2 = the curves guide of the standard
// model (HDRNetCurves._guide, hdrnet/models.py:145-190: 3x4 colour matrix, npts-knot ReLU curves per
// channel, channel mixing, clip) -- the network the reference's standard GL shader evaluates in its
// slicing pass (benchmark/assets/std.frag:36-45), in the parameter layout
// hdrnet/bin/freeze_graph.py:107-127 exports (guide_ccm_f32_3x4.bin, guide_shifts_f32_16x3.bin,
// guide_slopes_f32_16x3.bin, guide_mix_matrix_f32_1x4.bin).
constexpr int kGuideMap = 0, kGuideNN = 1, kGuideCurves = 2;

struct GuideNet {
  const float* conv1;  // NN: [n][CIN + 1]            curves: ccm [CIN][CIN + 1] (row = output channel)
  const float* conv2;  // NN: [n + 1]                 curves: mix [CIN + 1]
  const float* shifts;  //                            curves: [n][CIN]
  const float* slopes;  //                            curves: [n][CIN]
  float* guide_out;    // optional
  int n;               // NN: features                curves: knots per channel
};

template <int CIN>
__device__ __forceinline__ float guide_net_pixel(const GuideNet& gn, const float (&in)[CIN]) {
  float acc = gn.conv2[gn.n];
#pragma unroll 4
  for (int k = 0; k < gn.n; ++k) {
    const float* w = gn.conv1 + k * (CIN + 1);
    float h = w[CIN];
#pragma unroll
    for (int j = 0; j < CIN; ++j) h = fmaf(w[j], in[j], h);
    acc = fmaf(gn.conv2[k], fmaxf(h, 0.0f), acc);
  }
  return 1.0f / (1.0f + expf(-acc));
}

// guide = clip(mix[CIN] + sum_c mix[c] * sum_k slopes[k][c] * relu(t_c - shifts[k][c]), 0, 1),
// t_c = ccm[c][CIN] + sum_j ccm[c][j] * in_j     (models.py:157-188)
template <int CIN>
__device__ __forceinline__ float guide_curves_pixel(const GuideNet& gn, const float (&in)[CIN]) {
  float t[CIN], cv[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    const float* w = gn.conv1 + c * (CIN + 1);  // wave-uniform -> scalar loads
    float h = w[CIN];
#pragma unroll
    for (int j = 0; j < CIN; ++j) h = fmaf(w[j], in[j], h);
    t[c] = h;
    cv[c] = 0.0f;
  }
#pragma unroll 4
  for (int k = 0; k < gn.n; ++k) {
#pragma unroll
    for (int c = 0; c < CIN; ++c)
      cv[c] = fmaf(gn.slopes[k * CIN + c], fmaxf(t[c] - gn.shifts[k * CIN + c], 0.0f), cv[c]);
  }
  float g = gn.conv2[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) g = fmaf(gn.conv2[c], cv[c], g);
  return fminf(fmaxf(g, 0.0f), 1.0f);  // tf.clip_by_value(guidemap, 0, 1)
}

// Load 4 pixels x CIN channels of TI starting at element index e0, as floats / white level.
template <typename TI, int N>
__device__ __forceinline__ void load_pixels(const TI* __restrict__ src, size_t e0, float wl,
                                            float (&dst)[N]) {
  if constexpr (sizeof(TI) == 4) {
#pragma unroll
    for (int q = 0; q < N; ++q) dst[q] = reinterpret_cast<const float*>(src)[e0 + q];
  } else {
    static_assert((N * sizeof(TI)) % 4 == 0, "whole dwords per thread");
    constexpr int ND = N * sizeof(TI) / 4;
    uint32_t w[ND];
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + e0);
#pragma unroll
    for (int q = 0; q < ND; ++q) w[q] = p[q];
#pragma unroll
    for (int q = 0; q < N; ++q) {
      uint32_t v;
      if constexpr (sizeof(TI) == 1) v = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
      else v = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
      dst[q] = (float)v / wl;  // tf.to_float(im) / white_level, IEEE division as TF
    }
  }
}

template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
__global__ __launch_bounds__(256) void apply_fwd_io_rows(
    const float* __restrict__ grid, const float* __restrict__ guide, const TI* __restrict__ input,
    TO* __restrict__ out, int H, int W, int GH, int GW, int GD, int nseg, int seg,
    int slab_offset_floats, float scale_x, float scale_y, float white_level, GuideNet gn) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int NI = CIN * kPxPerThread, NO = COUT * kPxPerThread;
  extern __shared__ __attribute__((aligned(16))) float colY[];
  const int bid = blockIdx.x;
  const int segi = bid % nseg;
  const int row = bid / nseg;  // = b * H + y
  const int y = row % H;
  const int b = row / H;
  const int xs = segi * seg;
  const int xe = min(xs + seg, W);
  const float* grid_b = grid + (size_t)b * GH * GW * GD * C;
  const int x = xs + kPxPerThread * threadIdx.x;
  const bool active = x < xe;
  const size_t p = (size_t)row * W + x;

  float gs[kPxPerThread] = {0.f, 0.f, 0.f, 0.f};
  float inf[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) inf[q] = 0.0f;
  if (active) {
    if constexpr (GUIDE == kGuideMap) {
      const float4 g4 = *reinterpret_cast<const float4*>(guide + p);
      gs[0] = g4.x; gs[1] = g4.y; gs[2] = g4.z; gs[3] = g4.w;
    }
    load_pixels<TI, NI>(input, p * CIN, white_level, inf);
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);

  float of[NO];
  if (active) {
    if constexpr (GUIDE != kGuideMap) {
#pragma unroll
      for (int k = 0; k < kPxPerThread; ++k) {
        float in[CIN];
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
        gs[k] = GUIDE == kGuideNN ? guide_net_pixel<CIN>(gn, in) : guide_curves_pixel<CIN>(gn, in);
      }
      if (gn.guide_out) *reinterpret_cast<float4*>(gn.guide_out + p) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
    const float xf0 = (float)x + 0.5f;
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      const SliceTerms t = slice_terms<C, false>(r, xf0 + (float)k, gs[k]);
      CoefVec<C> coef;
      accum_vec<C, true>(coef, r.colY, t.a00, t.wx0 * t.wz0);
      accum_vec<C, false>(coef, r.colY, t.a01, t.wx0 * t.wz1);
      accum_vec<C, false>(coef, r.colY, t.a10, t.wx1 * t.wz0);
      accum_vec<C, false>(coef, r.colY, t.a11, t.wx1 * t.wz1);
      constexpr int CJ = CIN + (OFFSET ? 1 : 0);
#pragma unroll
      for (int i = 0; i < COUT; ++i) {
        float v = OFFSET ? coef.get(i * CJ + CIN) : 0.0f;
#pragma unroll
        for (int j = 0; j < CIN; ++j) v = fmaf(coef.get(i * CJ + j), inf[k * CIN + j], v);
        of[k * COUT + i] = v;
      }
    }
  }

  if constexpr (sizeof(TO) == 1) {
    // tf.cast(255 * clip(out, 0, 1), uint8): truncation.  12 bytes per lane, contiguous across
    // the wave: plain per-lane stores are already dense.
    static_assert(NO % 4 == 0, "whole dwords per thread");
    if (active) {
      uint32_t w[NO / 4];
#pragma unroll
      for (int q = 0; q < NO / 4; ++q) w[q] = 0;
#pragma unroll
      for (int q = 0; q < NO; ++q) {
        const float c = fminf(fmaxf(of[q], 0.0f), 1.0f);
        w[q >> 2] |= ((uint32_t)(255.0f * c)) << (8 * (q & 3));
      }
      uint32_t* op = reinterpret_cast<uint32_t*>(out + p * COUT);
#pragma unroll
      for (int q = 0; q < NO / 4; ++q) op[q] = w[q];
    }
  } else {
    // float output: lane-contiguous stores through the per-wave LDS slab (apply_fwd_rows.hip)
    float4* slab = reinterpret_cast<float4*>(colY + slab_offset_floats) + (threadIdx.x >> 6) * (64 * COUT);
    const int lane = threadIdx.x & 63;
    if (active) {
#pragma unroll
      for (int q = 0; q < COUT; ++q)
        slab[lane * COUT + q] = make_float4(of[4 * q], of[4 * q + 1], of[4 * q + 2], of[4 * q + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int wave_x0 = xs + kPxPerThread * (int)(threadIdx.x & ~63u);
    const int nvalid = (min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0) * COUT / 4;
    float4* gp = reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + ((size_t)row * W + wave_x0) * COUT);
#pragma unroll
    for (int k = 0; k < COUT; ++k) {
      const int e = lane + 64 * k;
      if (e < nvalid) gp[e] = slab[e];
    }
  }
}

template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
hipError_t launch_io(const ApplyIoArgs& a, const Plan& pl, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const int slab_off = round_up(pl.max_cols * a.GD * C, 4);
  const size_t lds = ((size_t)slab_off + (size_t)(pl.threads / 64) * 64 * kPxPerThread * COUT) * sizeof(float);
  const long long nblocks = (long long)a.B * a.H * pl.nseg;
  const GuideNet gn{a.guide_conv1, a.guide_conv2, a.guide_shifts, a.guide_slopes, a.guide_out, a.n_feats};
  apply_fwd_io_rows<CIN, COUT, OFFSET, GUIDE, TI, TO><<<(unsigned)nblocks, pl.threads, lds, s>>>(
      a.grid, a.guide, static_cast<const TI*>(a.input), static_cast<TO*>(a.out), a.H, a.W, a.GH, a.GW,
      a.GD, pl.nseg, pl.seg, slab_off, (float)a.GW / a.W, (float)a.GH / a.H, a.white_level, gn);
  return hipGetLastError();
}

template <int GUIDE, typename TI, typename TO>
hipError_t dispatch_shape(const ApplyIoArgs& a, const Plan& pl, hipStream_t s) {
  if (a.Cin == 3 && a.Cout == 3 && a.has_offset) return launch_io<3, 3, true, GUIDE, TI, TO>(a, pl, s);
  return hipErrorInvalidValue;
}

template <int GUIDE>
hipError_t dispatch_types(const ApplyIoArgs& a, const Plan& pl, hipStream_t s) {
  const int in = a.input_dtype, out = a.output_dtype;
  if (in == 1 && out == 1) return dispatch_shape<GUIDE, uint8_t, uint8_t>(a, pl, s);
  if (in == 1 && out == 0) return dispatch_shape<GUIDE, uint8_t, float>(a, pl, s);
  if (in == 2 && out == 1) return dispatch_shape<GUIDE, uint16_t, uint8_t>(a, pl, s);
  if (in == 2 && out == 0) return dispatch_shape<GUIDE, uint16_t, float>(a, pl, s);
  if (in == 0 && out == 1) return dispatch_shape<GUIDE, float, uint8_t>(a, pl, s);
  if (in == 0 && out == 0) return dispatch_shape<GUIDE, float, float>(a, pl, s);
  return hipErrorInvalidValue;
}

bool plan_io(const ApplyIoArgs& a, Plan* pl) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return false;
  if (a.W % 4 != 0) return false;
  const uintptr_t bits = (uintptr_t)a.grid | (uintptr_t)a.guide | (uintptr_t)a.guide_out |
                         (a.output_dtype == 0 ? (uintptr_t)a.out : 0) |
                         (a.input_dtype == 0 ? (uintptr_t)a.input : 0);
  if (bits & 15u) return false;
  if (((uintptr_t)a.input | (uintptr_t)a.out) & 3u) return false;
  *pl = make_row_plan(a.W, a.GW, true);
  if ((long long)a.B * a.H * pl->nseg > 0x7fffffffLL) return false;
  const size_t lds = ((size_t)pl->max_cols * a.GD * 12 + 4 + (size_t)(pl->threads / 64) * 64 * kPxPerThread * 3) * sizeof(float);
  return lds <= 64 * 1024;
}

}  // namespace

bool apply_fwd_io_supported(const ApplyIoArgs& a) {
  Plan pl;
  return plan_io(a, &pl);
}

hipError_t launch_apply_fwd_io(const ApplyIoArgs& a, hipStream_t s, const char** name) {
  Plan pl;
  if (!plan_io(a, &pl)) return hipErrorInvalidValue;
  static const char* const io[3][2] = {{"f32->f32", "f32->u8"}, {"u8->f32", "u8->u8"}, {"u16->f32", "u16->u8"}};
  static const char* const suffix[3] = {"", "+nnguide", "+curvesguide"};
  static thread_local char label[64];
  const int kind = a.guide ? kGuideMap : (a.guide_shifts ? kGuideCurves : kGuideNN);
  snprintf(label, sizeof label, "apply_fwd_io/%s%s", io[a.input_dtype][a.output_dtype], suffix[kind]);
  *name = label;
  if (kind == kGuideCurves) return dispatch_types<kGuideCurves>(a, pl, s);
  return kind == kGuideNN ? dispatch_types<kGuideNN>(a, pl, s) : dispatch_types<kGuideMap>(a, pl, s);
}

}  // namespace hdrnet_amd
