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
#include <cstring>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
#include "seg_common.hip.h"

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

typedef __attribute__((address_space(4))) const float cfloat;  // wave-uniform parameters: s_load

// guide = clip(mix[CIN] + sum_c mix[c] * sum_k slopes[k][c] * relu(t_c - shifts[k][c]), 0, 1),
// t_c = ccm[c][CIN] + sum_j ccm[c][j] * in_j     (models.py:157-188), for a lane's 4 pixels at once: one
// pass over the knots, every parameter read once through the constant address space.
template <int CIN>
__device__ __forceinline__ void guide_curves_quad(const GuideNet& gn, const float* inf, float (&g)[kPxPerThread]) {
  cfloat* ccm = (cfloat*)gn.conv1;
  cfloat* mix = (cfloat*)gn.conv2;
  cfloat* shifts = (cfloat*)gn.shifts;
  cfloat* slopes = (cfloat*)gn.slopes;
  // pixels in pairs on v_pk_fma_f32 / v_pk_add_f32 (the same per-pixel operation chains as the scalar form)
  static_assert(kPxPerThread == 4, "two pixel pairs");
  f32x2 t[2][CIN], cv[2][CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    float w[CIN + 1];
#pragma unroll
    for (int j = 0; j <= CIN; ++j) w[j] = ccm[c * (CIN + 1) + j];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 hv = {w[CIN], w[CIN]};
#pragma unroll
      for (int j = 0; j < CIN; ++j)
        hv = __builtin_elementwise_fma(f32x2{w[j], w[j]}, f32x2{inf[(2 * h) * CIN + j], inf[(2 * h + 1) * CIN + j]}, hv);
      t[h][c] = hv;
      cv[h][c] = f32x2{0.0f, 0.0f};
    }
  }
#pragma unroll 4
  for (int k = 0; k < gn.n; ++k) {
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float sl = slopes[k * CIN + c], sh = shifts[k * CIN + c];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 d = t[h][c] - f32x2{sh, sh};
        const f32x2 r = {fmaxf(d.x, 0.0f), fmaxf(d.y, 0.0f)};
        cv[h][c] = __builtin_elementwise_fma(f32x2{sl, sl}, r, cv[h][c]);
      }
    }
  }
  float m[CIN + 1];
#pragma unroll
  for (int c = 0; c <= CIN; ++c) m[c] = mix[c];
#pragma unroll
  for (int q = 0; q < kPxPerThread; ++q) {
    float v = m[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) v = fmaf(m[c], (q & 1) ? cv[q >> 1][c].y : cv[q >> 1][c].x, v);
    g[q] = fminf(fmaxf(v, 0.0f), 1.0f);  // tf.clip_by_value(guidemap, 0, 1)
  }
}

// v / wl for an integer sample v, correctly rounded like the IEEE division TF performs
// (tf.to_float(im) / white_level), in three instructions instead of the ~11 of a general IEEE divide:
//   q = v * r;  e = fma(-q, wl, v);  q' = fma(e, r, q)      with r = RN(1 / wl) from the host.
// With a correctly rounded reciprocal, one exact-remainder correction yields RN(v / wl) for every v
// unless wl's significand is all ones (Markstein, "Computation of elementary functions on the IBM RISC
// System/6000 processor", 1990, theorem on division by a correctly rounded reciprocal); v <= 65535 and
// wl in [2^-40, 2^40] keep every intermediate normal.  The host (io_fast_div) checks those conditions and
// otherwise selects the plain divide; tests/test_gpu_parity.py compares both forms exhaustively over
// all 65536 sample values for the white levels of hdrnet/data_pipeline.py:202-232,267-274.
struct WhiteLevel {
  float wl, rcp;  // rcp = 0: use the IEEE divide
};

__device__ __forceinline__ float div_white(float v, const WhiteLevel& w) {
  if (w.rcp == 0.0f) return v / w.wl;  // uniform
  const float q = v * w.rcp;
  const float e = __builtin_fmaf(-q, w.wl, v);
  return __builtin_fmaf(e, w.rcp, q);
}

// Load 4 pixels x CIN channels of TI starting at element index e0, as floats / white level.
template <typename TI, int N>
__device__ __forceinline__ void load_pixels(const TI* __restrict__ src, size_t e0, const WhiteLevel& wl,
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
      dst[q] = div_white((float)v, wl);  // tf.to_float(im) / white_level, rounded as TF's IEEE division
    }
  }
}

struct IoParams {
  const float* grid;
  const float* guide;
  const void* input;
  void* out;
  int H, W, GH, GW, GD;
  int seg, slab_off;
  float scale_x, scale_y, inv_col;
  WhiteLevel white;
  int grid_image;  // floats per image of the grid
  SegTab tab;      // (cmin, ncols) per segment, from the host (seg_common.hip.h)
  GuideNet gn;
};

// Geometry, LDS image and pixel core are apply_fwd_seg.hip's (seg_common.hip.h): a workgroup owns a row
// segment, 3-D launch grid (segment, row, image), padded y-pre-lerped coefficient image.
template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
__global__ __launch_bounds__(256) void apply_fwd_io_rows(const IoParams p) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int CB = C * (int)sizeof(float);
  constexpr int NI = CIN * kPxPerThread, NO = COUT * kPxPerThread;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xs = blockIdx.x * p.seg;
  const int xe = min(xs + p.seg, p.W);
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  const float* grid_b = p.grid + (size_t)b * (unsigned)p.grid_image;
  const int x = xs + kPxPerThread * tid;
  const bool active = x < xe;
  const size_t row = (unsigned)b * (unsigned)p.H + (unsigned)y;  // B, H <= 65535 (plan_io)
  const size_t px = row * p.W + x;
  const TI* input = static_cast<const TI*>(p.input);

  float gs[kPxPerThread] = {0.f, 0.f, 0.f, 0.f};
  float inf[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) inf[q] = 0.0f;
  if (active) {
    if constexpr (GUIDE == kGuideMap) {
      const float4 g4 = *reinterpret_cast<const float4*>(p.guide + px);
      gs[0] = g4.x; gs[1] = g4.y; gs[2] = g4.z; gs[3] = g4.w;
    }
    load_pixels<TI, NI>(input, px * CIN, p.white, inf);
  }

  const SegCols sc = seg_cols_tab(p.tab, blockIdx.x, xs, xe, p.scale_x);
  const int colb = (p.GD + 2) * CB;
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);
  stage_image<C>(lds, grid_b, y, sc.cmin, sc.ncols, p.GH, p.GW, p.GD, p.scale_y, p.inv_col, tid, (int)blockDim.x);
  // the lean pixel phase of the product forward (seg_common.hip.h)
  XTermLean xt[kPxPerThread];
  const float xf0 = (float)x + 0.5f;
  const float colb_f = (float)colb, xbase_f = (float)(CB - sc.cmin * colb);
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) xt[k] = x_term_lean(xf0 + (float)k, p.scale_x, colb_f, xbase_f);
  __syncthreads();

  float of[NO];
  if (active) {
    if constexpr (GUIDE != kGuideMap) {
      if constexpr (GUIDE == kGuideNN)
        guide_nn_quad<CIN>(GuideNN{p.gn.conv1, p.gn.conv2, p.gn.guide_out, p.gn.n}, inf, gs);  // (writes nothing itself)
      else
        guide_curves_quad<CIN>(p.gn, inf, gs);
      if (p.gn.guide_out) *reinterpret_cast<float4*>(p.gn.guide_out + px) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      float in[CIN], o[COUT];
#pragma unroll
      for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
      seg_pixel_lean<CIN, COUT, OFFSET, true>(lds, gd_f, zhi, colb, xt[k], gs[k], in, o);
#pragma unroll
      for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
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
        const float c = __builtin_amdgcn_fmed3f(of[q], 0.0f, 1.0f);  // clip: folds into the affine's last fma (clamp)
        w[q >> 2] |= ((uint32_t)(255.0f * c)) << (8 * (q & 3));
      }
      uint32_t* op = reinterpret_cast<uint32_t*>(static_cast<TO*>(p.out) + px * COUT);
#pragma unroll
      for (int q = 0; q < NO / 4; ++q) op[q] = w[q];
    }
  } else {
    // float output: lane-contiguous nontemporal buffer stores through the per-wave LDS slab
    // (apply_fwd_seg.hip); the descriptor covers exactly the row segment
    float4* slab = reinterpret_cast<float4*>(lds + p.slab_off) + wave * (64 * COUT);
    if (active) {
#pragma unroll
      for (int q = 0; q < COUT; ++q)
        slab[lane * COUT + q] = make_float4(of[4 * q], of[4 * q + 1], of[4 * q + 2], of[4 * q + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned wpx = kPxPerThread * 64u * (unsigned)wave;
    float* oseg = static_cast<float*>(p.out) + (row * p.W + xs) * COUT;
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(oseg, (unsigned)(xe - xs) * COUT * 4u);
#pragma unroll
    for (int k = 0; k < COUT; ++k)
      buf_store16<kAuxStream>(slab[lane + 64 * k], orsrc, (wpx * COUT + 4u * (unsigned)(lane + 64 * k)) * 4u);
  }
}

// Host side of div_white: the reciprocal if the three-instruction form is exact for this white level.
WhiteLevel io_white_level(float wl) {
  unsigned bits;
  memcpy(&bits, &wl, sizeof bits);
  const bool all_ones = (bits & 0x7fffffu) == 0x7fffffu;
  const bool in_range = wl >= 0x1p-40f && wl <= 0x1p40f;
  volatile float r = 1.0f / wl;  // IEEE, correctly rounded
  return WhiteLevel{wl, (all_ones || !in_range) ? 0.0f : (float)r};
}

struct IoGeom {
  Plan pl;
  int slab_off;
  size_t lds;
};

IoGeom io_geom(int W, int GW, int GD, int C, int Cout) {
  IoGeom g;
  g.pl = make_row_plan(W, GW, true);
  const int max_cols = (int)(((long long)(g.pl.seg - 1) * GW) / W + 4);
  g.slab_off = round_up(max_cols * (GD + 2) * C, 4);
  g.lds = ((size_t)g.slab_off + (size_t)(g.pl.threads / 64) * 64 * kPxPerThread * Cout) * sizeof(float);
  return g;
}

template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
hipError_t launch_io(const ApplyIoArgs& a, const Plan&, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const IoGeom g = io_geom(a.W, a.GW, a.GD, C, COUT);
  IoParams p;
  p.grid = a.grid;
  p.guide = a.guide;
  p.input = a.input;
  p.out = a.out;
  p.H = a.H; p.W = a.W; p.GH = a.GH; p.GW = a.GW; p.GD = a.GD;
  p.seg = g.pl.seg;
  p.slab_off = g.slab_off;
  p.scale_x = (float)a.GW / a.W;
  p.scale_y = (float)a.GH / a.H;
  p.inv_col = 1.0f / (float)(a.GD * (C / 4));
  p.white = io_white_level(a.white_level);
  p.grid_image = a.GH * a.GW * a.GD * C;
  p.tab = make_seg_tab(a.W, g.pl.seg, g.pl.nseg, p.scale_x);
  p.gn = GuideNet{a.guide_conv1, a.guide_conv2, a.guide_shifts, a.guide_slopes, a.guide_out, a.n_feats};
  const dim3 grid3((unsigned)g.pl.nseg, (unsigned)a.H, (unsigned)a.B);
  apply_fwd_io_rows<CIN, COUT, OFFSET, GUIDE, TI, TO><<<grid3, g.pl.threads, g.lds, s>>>(p);
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
  const IoGeom g = io_geom(a.W, a.GW, a.GD, 12, 3);
  *pl = g.pl;
  if (a.B > 65535 || a.H > 65535 || (long long)a.W * a.Cout * 4 >= (1LL << 31)) return false;
  if ((long long)(g.slab_off) >= (1 << 20)) return false;
  return g.lds <= 64 * 1024;
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
