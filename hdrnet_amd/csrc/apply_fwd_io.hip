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
constexpr int kGuideCurvesScan = 3;  // the curves guide evaluated knot by knot: more than kCurveMaxKnots knots per channel
constexpr int kGuideCurvesCells = 4;  // the curves guide from the PREPARED uniform cell tables (CurveCells below)

struct GuideNet {
  const float* conv1;  // NN: [n][CIN + 1]            curves: ccm [CIN][CIN + 1] (row = output channel)
  const float* conv2;  // NN: [n + 1]                 curves: mix [CIN + 1]
  const float* shifts;  //                            curves: [n][CIN]
  const float* slopes;  //                            curves: [n][CIN]
  float* guide_out;    // optional
  int n;               // NN: features                curves: knots per channel
  bool fast_sigmoid;   // NN: GuideNN::fast_sigmoid (rows_common.hip.h)
  bool prescaled;      // NN: GuideNN::prescaled
  const float* prepared;  // curves: the uniform cell tables of hdrnet_curves_guide_prepare_f32 (CurveCells), or null
};

typedef __attribute__((address_space(4))) const float cfloat;  // wave-uniform parameters: s_load

// guide = clip(mix[CIN] + sum_c mix[c] * sum_k slopes[k][c] * relu(t_c - shifts[k][c]), 0, 1),
// t_c = ccm[c][CIN] + sum_j ccm[c][j] * in_j     (models.py:157-188), for a lane's 4 pixels at once: one
// pass over the knots, every parameter read once through the constant address space.  (The SCAN form: 3 operations
// per knot, channel and pixel -- 144 per pixel for the reference's 16 knots.  Kept for more than kCurveMaxKnots
// knots; the product path is the table form below.)
template <int CIN>
__device__ __forceinline__ void guide_curves_scan_quad(const GuideNet& gn, const float* inf, float (&g)[kPxPerThread]) {
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

// ---- the curves guide as a sorted piecewise-linear lookup (round 4) --------------------------------------------------
// sum_k slope_k * relu(t - shift_k) is piecewise linear in t with its breaks at the shifts.  Per workgroup, wave 0
// sorts each channel's knots (rank by counting; ties by index) and tabulates, for the interval that starts at the
// i-th smallest knot s_(i):  the knot, the curve's VALUE there  C_i = sum_{r<i} slope_(r) (s_(i) - s_(r))  and its SLOPE
// from there on  A_i = sum_{r<=i} slope_(r)  (both summed in float64, rounded once).  A pixel then finds its interval
// with a 4-step binary search over the sorted knots (an implicit tree in LDS, Eytzinger order: node e's children are
// 2e and 2e + 1) and evaluates  C_i + A_i * max(t - s_(i), 0)  -- anchored at the interval's own knot, so that the
// result carries one rounding of the curve's value instead of the scan's 16.  15 instead of 48 operations per channel
// and pixel, 4 ds_read_b32 + 1 ds_read_b128 (conflict-free: <= 16 consecutive dwords / 16-B entries per channel).
// Left of the smallest knot every relu is zero: leaf 0 holds C = 0 and the max() clamps the distance.
// Up to kCurveMaxKnots knots per channel (the reference has 16, hdrnet/models.py:150); unused slots are +inf knots
// with slope 0.  The tables sit at the START of dynamic LDS (compile-time addresses); the coefficient image follows.
constexpr int kCurveMaxKnots = 16;
template <int CIN>
struct CurveTab {
  static constexpr int kTree = 0;                    // [CIN][16]: node e of channel c at c * 16 + e (e = 1 .. 15)
  static constexpr int kLeaf = 64;                   // [CIN][16][4]: (knot, value at it, slope after it, 0); >= 64 floats in,
                                                     // so that (leaf base - 256 B) stays a non-negative immediate offset
  static constexpr int kRawS = kLeaf + CIN * 64;     // scratch while building: knots / slopes as given,
  static constexpr int kRawL = kRawS + CIN * 16;
  static constexpr int kSortS = kRawL + CIN * 16;    //   ... and sorted
  static constexpr int kSortL = kSortS + CIN * 16;
  static constexpr int kFloats = kSortL + CIN * 16;  // 16-B multiple
};

// Run by the first CIN * 16 lanes of wave 0 (lane = channel * 16 + knot); the caller's workgroup barrier publishes
// the tables.  CIN * 16 <= 64.
template <int CIN>
__device__ __forceinline__ void curves_build_tables(float* __restrict__ tab, const GuideNet& gn, int lane) {
  static_assert(CIN * 16 <= 64 && CIN * 16 <= CurveTab<CIN>::kLeaf, "one wave builds the tables; the tree fits below the leaves");
  typedef CurveTab<CIN> T;
  const int c = lane >> 4, k = lane & 15;
  const bool mine = lane < CIN * 16;
  float s = __builtin_inff(), sl = 0.0f;
  if (mine && k < gn.n) {
    s = gn.shifts[k * CIN + c];
    sl = gn.slopes[k * CIN + c];
  }
  if (mine) {
    tab[T::kRawS + lane] = s;
    tab[T::kRawL + lane] = sl;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (mine) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float sj = tab[T::kRawS + c * 16 + j];
      rank += (sj < s || (sj == s && j < k)) ? 1 : 0;
    }
    tab[T::kSortS + c * 16 + rank] = s;
    tab[T::kSortL + c * 16 + rank] = sl;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (mine) {
    const int i = k;
    const float si = tab[T::kSortS + c * 16 + i];
    double A = 0.0, Cv = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sr = tab[T::kSortS + c * 16 + r], lr = tab[T::kSortL + c * 16 + r];
      if (r < i) Cv += (double)lr * ((double)si - (double)sr);
      if (r <= i) A += (double)lr;
    }
    const bool dead = !(si < __builtin_inff());  // an unused slot (or an infinite / NaN knot): never reached
    float4 leaf = make_float4(si, (float)Cv, (float)A, 0.0f);
    if (dead) leaf = make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    *reinterpret_cast<float4*>(tab + T::kLeaf + (c * 16 + i) * 4) = leaf;
    if (i >= 1) {
      // sorted key j = i - 1 of the 15 search keys s_(1) .. s_(15) -> its Eytzinger node: with t = ctz(j + 1) the node sits
      // on level 3 - t at position (j + 1) >> (t + 1)
      const int j1 = i;  // j + 1
      const int t = __builtin_ctz(j1);
      const int e = (1 << (3 - t)) + (j1 >> (t + 1));
      tab[T::kTree + c * 16 + e] = si;
    }
  }
}

template <int CIN>
__device__ __forceinline__ float curve_lookup(const float* __restrict__ tab, int c, float v) {
  typedef CurveTab<CIN> T;
  // the walk carries the node's BYTE offset a = 4 * node: a' = 2 a + (v >= key ? 4 : 0) -- compare, select, shift-add --
  // and the ds_read takes it as it is (the channel's base is the instruction's immediate offset)
  const char* tree = reinterpret_cast<const char*>(tab + T::kTree + c * 16);
  unsigned a = 4u;
#pragma unroll
  for (int d = 0; d < 4; ++d) a = (a << 1) + ((v >= *reinterpret_cast<const float*>(tree + a)) ? 4u : 0u);
  // leaf = node - 16, 16 bytes each: byte offset 4 a - 256
  const char* leaves = reinterpret_cast<const char*>(tab + T::kLeaf + c * 64) - 256;
  const f32x4 leaf = *reinterpret_cast<const f32x4*>(leaves + (a << 2));
  return __builtin_fmaf(leaf.z, fmaxf(v - leaf.x, 0.0f), leaf.y);
}

// The guide of a lane's 4 pixels from the tables (same colour matrix / mixing / clip as the scan form).
template <int CIN>
__device__ __forceinline__ void guide_curves_quad(const float* __restrict__ tab, const GuideNet& gn, const float* inf,
                                                  float (&g)[kPxPerThread]) {
  cfloat* ccm = (cfloat*)gn.conv1;
  cfloat* mix = (cfloat*)gn.conv2;
  float w[CIN][CIN + 1], m[CIN + 1];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
#pragma unroll
    for (int j = 0; j <= CIN; ++j) w[c][j] = ccm[c * (CIN + 1) + j];
  }
#pragma unroll
  for (int c = 0; c <= CIN; ++c) m[c] = mix[c];
  float cv[kPxPerThread][CIN];
#pragma unroll
  for (int q = 0; q < kPxPerThread; ++q) {
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      float t = w[c][CIN];
#pragma unroll
      for (int j = 0; j < CIN; ++j) t = fmaf(w[c][j], inf[q * CIN + j], t);
      cv[q][c] = curve_lookup<CIN>(tab, c, t);
    }
  }
#pragma unroll
  for (int q = 0; q < kPxPerThread; ++q) {
    float v = m[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) v = fmaf(m[c], cv[q][c], v);
    g[q] = fminf(fmaxf(v, 0.0f), 1.0f);  // tf.clip_by_value(guidemap, 0, 1)
  }
}

// ---- the curves guide through UNIFORM CELL TABLES prepared once per parameter set (round 5) ---------------------------
// The sorted tables above are rebuilt by wave 0 of EVERY workgroup (~150 instructions + two wave barriers before the
// workgroup's barrier) and searched with four dependent LDS reads per channel and pixel.  Both costs go with a table that
// depends on the parameters only and is indexed by arithmetic: hdrnet_curves_guide_prepare_f32 cuts each channel's knot
// range [s_min, s_max] into kCurveCells uniform cells by the MONOTONE map
//     cell(t) = uint(clamp(fma(t, k16, o16), 0, 16 kCurveCells - 1)) >> 4,   k16 = 16 (kCurveCells - 1) / (s_max - s_min)
// (the same fp32 instructions for a knot at prepare time and for a pixel here: t1 <= t2 => cell(t1) <= cell(t2), so every
// knot of an earlier cell is <= t and every knot of a later cell is > t -- exactly, whatever the rounding) and stores per
// cell one 16-byte entry (s, C, A_lo, A_hi): the knot inside the cell, the curve's value at it, the slopes before and
// after it -- or, for a cell without a knot, the last knot to its left with A_lo = A_hi.  A pixel evaluates
//     C + (t >= s ? A_hi : A_lo) (t - s)
// -- one ds_read_b128 and 8 VALU instructions per channel instead of 4 + 1 dependent reads and 16; anchored at a knot at
// most one cell away, with the float64-summed C and A of the sorted tables (curves_build_tables, run once by the prepare
// kernel).  A cell can hold one knot only: if two knots of a channel share a cell (closer than 1/63 of the knot range, or
// equal) the prepare kernel clears the table's `ok` word: the table is NOT USABLE, hdrnet_curves_guide_prepare_f32 reads
// the word back and says so, and its caller passes no prepared buffer -- the sorted tables above, the same results.  (The
// two forms are two kernel instantiations chosen on the host: one kernel with both behind a run-time switch took 70-74
// VGPRs instead of 54-64.)  Layout of the prepared buffer (floats): [CIN][kCurveCells][4] entries, then per channel
// (k16, o16, 0, 0), then (ok, 0, 0, 0).
constexpr int kCurveCells = 64;
template <int CIN>
struct CurveCells {
  static constexpr int kEntries = 0;
  static constexpr int kScale = CIN * kCurveCells * 4;  // [CIN][4]
  static constexpr int kOk = kScale + CIN * 4;          // [4]
  static constexpr int kFloats = kOk + 4;
};

__device__ __forceinline__ unsigned curve_cell_byte(float t, float k16, float o16) {
  const float u = __builtin_amdgcn_fmed3f(__builtin_fmaf(t, k16, o16), 0.0f, (float)(16 * kCurveCells - 1));
  return (unsigned)u & ~15u;
}

template <int CIN>
__global__ __launch_bounds__(256) void curves_prepare_kernel(const GuideNet gn, float* __restrict__ out) {
  typedef CurveTab<CIN> T;
  typedef CurveCells<CIN> L;
  __shared__ __attribute__((aligned(16))) float tab[T::kFloats];
  __shared__ int cell_of[CIN * 16];
  __shared__ float kk[CIN], oo[CIN];
  __shared__ int bad;
  const int tid = threadIdx.x;
  if (tid == 0) bad = 0;
  if (tid < 64) curves_build_tables<CIN>(tab, gn, tid);  // sorted leaves (knot, value at it, slope after it, 0); dead: +inf
  __syncthreads();
  const float inf = __builtin_inff();
  if (tid < CIN) {
    const float* lf = tab + T::kLeaf + tid * 64;
    const float smin = lf[0];
    float smax = smin;
    for (int i = 1; i < 16; ++i) {
      const float s = lf[4 * i];
      if (s < inf) smax = s;
    }
    float k16 = 0.0f, o16 = 0.0f;
    if (smin < inf && smax > smin) {
      k16 = (float)(16 * (kCurveCells - 1)) / (smax - smin);
      o16 = -smin * k16;
      if (!(k16 < inf) || !(fabsf(o16) < inf)) k16 = o16 = 0.0f;  // a range too narrow to scale: one cell (ok only for one knot)
    }
    kk[tid] = k16;
    oo[tid] = o16;
  }
  __syncthreads();
  if (tid < CIN * 16) {
    const float s = tab[T::kLeaf + tid * 4];
    cell_of[tid] = (s < inf) ? (int)(curve_cell_byte(s, kk[tid >> 4], oo[tid >> 4]) >> 4) : 0x7fffffff;  // dead knots: no cell
  }
  __syncthreads();
  for (int e = tid; e < CIN * kCurveCells; e += (int)blockDim.x) {
    const int c = e / kCurveCells, j = e - c * kCurveCells;
    int here = -1, count = 0, last = -1;
    for (int i = 0; i < 16; ++i) {  // sorted knots + a monotone cell map: cell_of is non-decreasing in i
      const int ci = cell_of[c * 16 + i];
      if (ci == j) {
        here = i;
        ++count;
      }
      if (ci < j) last = i;
    }
    if (count > 1) bad = 1;  // (every writer writes 1)
    float4 ent = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // left of every knot: the curve is 0
    const float4* leaves = reinterpret_cast<const float4*>(tab + T::kLeaf + c * 64);
    if (here >= 0) {
      const float4 lf = leaves[here];
      ent = make_float4(lf.x, lf.y, here > 0 ? leaves[here - 1].z : 0.0f, lf.z);
    } else if (last >= 0) {
      const float4 lf = leaves[last];
      ent = make_float4(lf.x, lf.y, lf.z, lf.z);
    }
    reinterpret_cast<float4*>(out + L::kEntries)[e] = ent;
  }
  __syncthreads();
  if (tid < CIN) reinterpret_cast<float4*>(out + L::kScale)[tid] = make_float4(kk[tid], oo[tid], 0.0f, 0.0f);
  if (tid == 0) reinterpret_cast<float4*>(out + L::kOk)[0] = make_float4(bad ? 0.0f : 1.0f, 0.0f, 0.0f, 0.0f);
}

// The guide of a lane's 4 pixels from the cell tables (`cells`: the entries, copied into LDS by the workgroup).  The colour
// matrix and the cell coordinate run on pixel pairs (v_pk_fma_f32, parameters from SGPRs); mixing and clip as above.
template <int CIN>
__device__ __forceinline__ void guide_curves_quad_cells(const float* __restrict__ cells, const GuideNet& gn, const float* inf,
                                                        float (&g)[kPxPerThread]) {
  typedef CurveCells<CIN> L;
  cfloat* ccm = (cfloat*)gn.conv1;
  cfloat* mix = (cfloat*)gn.conv2;
  cfloat* sc = (cfloat*)(gn.prepared + L::kScale);
  static_assert(kPxPerThread == 4, "two pixel pairs");
  float cv[kPxPerThread][CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    float w[CIN + 1];
#pragma unroll
    for (int j = 0; j <= CIN; ++j) w[j] = ccm[c * (CIN + 1) + j];
    const float k16 = sc[4 * c], o16 = sc[4 * c + 1];
    const char* base = reinterpret_cast<const char*>(cells + c * kCurveCells * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 t = {w[CIN], w[CIN]};
#pragma unroll
      for (int j = 0; j < CIN; ++j)
        t = __builtin_elementwise_fma(f32x2{w[j], w[j]}, f32x2{inf[(2 * h) * CIN + j], inf[(2 * h + 1) * CIN + j]}, t);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float ti = i ? t.y : t.x;
        const f32x4 e = *reinterpret_cast<const f32x4*>(base + curve_cell_byte(ti, k16, o16));
        const float d = ti - e.x;
        cv[2 * h + i][c] = __builtin_fmaf(d >= 0.0f ? e.w : e.z, d, e.y);
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
    for (int c = 0; c < CIN; ++c) v = fmaf(m[c], cv[q][c], v);
    g[q] = fminf(fmaxf(v, 0.0f), 1.0f);  // tf.clip_by_value(guidemap, 0, 1)
  }
}

// ---- TOOLS BUILD ONLY (knob 5): the point-wise guide network's hidden layer on the bf16 matrix cores, uint8 input ----
// Round 4 experiment, bit-level parity green (guide within 1e-6 of the oracle), REJECTED on time:
// u8 -> guide network -> u8 at 4K 39.5 us on the VALU, 47.2 us with 48 of these matrix instructions per wave (max + fma
// on the VALU), 53.4 us with 60 (the |h| form below) -- a v_mfma_f32_4x4x4_16B_bf16 costs its SIMD ~25 cycles for 1024
// multiply-adds, and an f32-accurate product needs three of them (hi / mid / lo of the weight): no faster than the 3
// VALU FMAs it replaces.  The 16 x 16 x 16 form does 96 useful multiply-adds per cycle, / 3 parts = the VALU's own 32, and
// needs every pixel's bytes in four lanes (an LDS round trip).  A K = 4 contraction is too small for the matrix cores at
// f32 accuracy.  profiles/r04/guide_nn_mfma.md.  The product evaluates the network with guide_nn_quad (rows_common.hip.h).
// h[f] = conv1[f][3] + sum_j conv1[f][j] * (v_j / white) is a 16 x 4 by 4 x pixels contraction.  For UINT8 samples the
// pixel side is exact in bf16 (integers <= 255; the 1 of the bias column too), so the only thing to split is the
// weight side: w' = w / white (f32) = hi + mid + lo, three bf16 that carry its 24 significant bits, every product exact in
// the matrix core's f32 accumulator.  v_mfma_f32_4x4x4_16B_bf16 multiplies sixteen independent 4 x 4 blocks: in block b
// (lanes 4b .. 4b + 3) lane 4b + i supplies row i of A (feature 4 fg + i: its three weights and its bias, one of the three
// parts), lane 4b + j supplies column j of B -- ITS OWN pixel's (r, g, b, 1) -- and receives column j of D: the four
// features of its own pixel.  No cross-lane traffic on the pixel side at all; 3 parts x (n / 4) feature groups
// matrix instructions per pixel quad-slot replace 3 n VALU FMAs per pixel, and the bf16 pipe runs beside the VALU
// (the f32 matrix instructions do not: profiles/r01/f_ubench_mfma_valu_overlap.txt).  ReLU, the mixing layer and the
// sigmoid stay on the VALU.  A operands: prepared once per workgroup by wave 0 in LDS (NnTab), re-read per feature group.
// Requires n % 4 == 0, n <= 16; other sizes / input types take guide_nn_quad.
#ifdef HDRNET_TOOLS_BUILD
constexpr bool kToolsBuild = true;
#else
constexpr bool kToolsBuild = false;
#endif
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
struct NnTab {
  // [feature group 0 .. 3, 4 = the linear row][part][row i][2 dwords = 4 bf16]
  static constexpr int kWords = 5 * 3 * 4 * 2 + 4;  // + 4 floats of scratch for the linear row; 16-B multiple
};

__device__ __forceinline__ unsigned bf16_rne_bits(float x) {  // round-to-nearest-even truncation to bf16 (finite x)
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// w[0 .. 3] -> its three bf16 parts, packed as the A operand's two dwords, at tab[((group * 3 + part) * 4 + row) * 2].
__device__ __forceinline__ void nn_store_row(unsigned* __restrict__ tab, int group, int row, const float (&w)[4]) {
  unsigned part[3][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float r = w[k];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned b = bf16_rne_bits(r);
      part[q][k] = b;
      r -= __uint_as_float(b << 16);  // exact: the remainder of a bf16 rounding fits a float
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    unsigned* d = tab + ((group * 3 + q) * 4 + row) * 2;
    d[0] = part[q][0] | (part[q][1] << 16);
    d[1] = part[q][2] | (part[q][3] << 16);
  }
}

// relu(h) = (h + |h|) / 2:  bias2 + sum_f m_f relu(h_f) = [bias2 + sum_f (m_f / 2) h_f] + sum_f (m_f / 2) |h_f|.  The
// bracket is affine in the pixel -- ONE more row for the matrix cores (group 4, row 0: L_j = sum_f (m_f / 2) w'_fj,
// L_3 = bias2 + sum_f (m_f / 2) b_f, summed in float64) -- and the rest is one FMA with the |.| source modifier per
// feature and pixel: no max, no second accumulation.
// Run by lanes 0 .. 19 of wave 0: lane < 16 = a feature row, lanes 16 .. 19 = the linear row's four entries.
__device__ __forceinline__ void nn_build_tables(unsigned* __restrict__ tab, float* __restrict__ lin, const GuideNet& gn,
                                                float white, int lane) {
  if (lane < 16) {
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < gn.n) {
#pragma unroll
      for (int j = 0; j < 3; ++j) w[j] = gn.conv1[lane * 4 + j] / white;  // (w / white) * v == w * (v / white) to an ulp
      w[3] = gn.conv1[lane * 4 + 3];
    }
    nn_store_row(tab, lane >> 2, lane & 3, w);
    if (lane >= 1 && lane < 4) {  // rows 1 .. 3 of the linear group: zero
      const float z[4] = {0.f, 0.f, 0.f, 0.f};
      nn_store_row(tab, 4, lane, z);
    }
  } else if (lane < 20) {
    const int j = lane - 16;
    double acc = j == 3 ? (double)gn.conv2[gn.n] : 0.0;
    for (int f = 0; f < gn.n; ++f) {
      const float w = j == 3 ? gn.conv1[f * 4 + 3] : gn.conv1[f * 4 + j] / white;
      acc += 0.5 * (double)gn.conv2[f] * (double)w;
    }
    lin[j] = (float)acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane == 0) {
    const float w[4] = {lin[0], lin[1], lin[2], lin[3]};
    nn_store_row(tab, 4, 0, w);
  }
}

// raw: the lane's 12 bytes (4 RGB pixels).  g[q] = the guide of pixel q.
__device__ __forceinline__ void guide_nn_quad_mfma_u8(const unsigned* __restrict__ tab, const GuideNet& gn,
                                                      const uint32_t (&raw)[3], int lane, float (&g)[kPxPerThread]) {
  cfloat* c2 = (cfloat*)gn.conv2;
  // B operands: bf16(v) is the upper half of float(v) for an integer v <= 255
  bf16x4_t bq[kPxPerThread];
#pragma unroll
  for (int q = 0; q < kPxPerThread; ++q) {
    float ch[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int byte = 3 * q + c;
      ch[c] = (float)((raw[byte >> 2] >> (8 * (byte & 3))) & 0xffu);  // v_cvt_f32_ubyteN
    }
    const unsigned lo = (__float_as_uint(ch[0]) >> 16) | (__float_as_uint(ch[1]) & 0xffff0000u);
    const unsigned hi = (__float_as_uint(ch[2]) >> 16) | 0x3f800000u;  // (b, 1.0)
    bq[q] = __builtin_bit_cast(bf16x4_t, uint2{lo, hi});
  }
  const unsigned* arow = tab + (lane & 3) * 2;
  float acc[kPxPerThread];
  {  // the affine part: group 4, row 0
    bf16x4_t a[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x4_t, *reinterpret_cast<const uint2*>(arow + (4 * 3 + q) * 8));
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[2], bq[q], h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[1], bq[q], h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[0], bq[q], h, 0, 0, 0);
      acc[q] = h[0];
    }
  }
  const int ngroups = gn.n >> 2;
#pragma unroll 1
  for (int fg = 0; fg < ngroups; ++fg) {
    bf16x4_t a[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x4_t, *reinterpret_cast<const uint2*>(arow + (fg * 3 + q) * 8));
    float m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = 0.5f * c2[fg * 4 + i];
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
      // small terms first: lo, mid, hi
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[2], bq[q], h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[1], bq[q], h, 0, 0, 0);
      h = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[0], bq[q], h, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[q] = fmaf(m[i], __builtin_fabsf(h[i]), acc[q]);
    }
  }
  if (!gn.fast_sigmoid) {  // the reference's sigmoid (rows_common.hip.h: GuideNN::fast_sigmoid)
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q) g[q] = 1.0f / (1.0f + expf(-acc[q]));
  } else {
#pragma unroll
    for (int q = 0; q < kPxPerThread; ++q)
      g[q] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * acc[q]));
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
  float inv;      // RN(1 / wl), always: the factor folded into the coefficient image where the input feeds the affine only
};

__device__ __forceinline__ float div_white(float v, const WhiteLevel& w) {
  if (w.rcp == 0.0f) return v / w.wl;  // uniform
  const float q = v * w.rcp;
  const float e = __builtin_fmaf(-q, w.wl, v);
  return __builtin_fmaf(e, w.rcp, q);
}

// Load 4 pixels x CIN channels of TI starting at element index e0, as floats / white level.
// UNSCALED: the samples as they are, (float)v -- the white level then sits in the coefficient image (stage_image IN_SCALE).
template <typename TI, int N, bool UNSCALED = false>
__device__ __forceinline__ void load_pixels(const TI* __restrict__ src, size_t e0, const WhiteLevel& wl,
                                            float (&dst)[N], uint32_t* raw = nullptr) {
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
    if (raw) {
#pragma unroll
      for (int q = 0; q < ND; ++q) raw[q] = w[q];
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      uint32_t v;
      if constexpr (sizeof(TI) == 1) v = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
      else v = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
      if constexpr (UNSCALED) dst[q] = (float)v;
      else dst[q] = div_white((float)v, wl);  // tf.to_float(im) / white_level, rounded as TF's IEEE division
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
  int nn_mfma;     // tools build, knob 5: the guide network's hidden layer on the matrix cores (uint8 input)
};

// Geometry, LDS image and pixel core are apply_fwd_seg.hip's (seg_common.hip.h): a workgroup owns a row
// segment, 3-D launch grid (segment, row, image), padded y-pre-lerped coefficient image.
template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
__global__ __launch_bounds__(256) void apply_fwd_io_rows(const IoParams p) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int CB = C * (int)sizeof(float);
  constexpr int NI = CIN * kPxPerThread, NO = COUT * kPxPerThread;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  // curves guide: its lookup tables first (compile-time LDS addresses), the coefficient image behind them
  // Integer samples with a guide MAP: the input is used by the affine only, so the white level goes into the staged
  // coefficients (coef / wl) * v, and the 12 divisions per lane (3 instructions each) disappear: u8 -> u8 26.7 -> ... us.
  // (With a guide NETWORK the guide's own input stays v / wl, IEEE-rounded as TensorFlow's division.)
  constexpr bool FOLD_WL = GUIDE == kGuideMap && sizeof(TI) < 4 && C == 12;
#ifdef HDRNET_TOOLS_BUILD
  constexpr bool NN_MFMA = GUIDE == kGuideNN && sizeof(TI) == 1 && CIN == 3;  // experiment: hidden layer on the bf16 matrix cores
#else
  constexpr bool NN_MFMA = false;
#endif
  // The curves guide's lookup tables are STATIC LDS: their addresses are compile-time constants, so a table walk's
  // ds_read takes the walked byte offset as its address register and the table's base as its immediate (behind the
  // dynamic array's link-time base every step paid a v_add).  The coefficient image and the slabs stay dynamic.
  constexpr int kTabFloats = NN_MFMA ? NnTab::kWords : 0;
  float* const lds = lds_all + kTabFloats;
  // (static LDS costs resident workgroups here: 3 KB more measured 8 %, and a kernel carrying BOTH table forms behind a
  //  run-time switch took 70-74 VGPRs instead of 54-64 -- profiles/r05/f2_prepared_guides.md -- so the two forms are two
  //  instantiations, chosen on the host by whether the caller passed a prepared buffer)
  [[maybe_unused]] float* ctab = nullptr;
  if constexpr (GUIDE == kGuideCurves) {
    __shared__ __attribute__((aligned(16))) float curve_tables[CurveTab<CIN>::kFloats];
    ctab = curve_tables;
  }
  if constexpr (GUIDE == kGuideCurvesCells) {
    __shared__ __attribute__((aligned(16))) float curve_cells[CIN * kCurveCells * 4];
    ctab = curve_cells;
  }
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
  [[maybe_unused]] uint32_t raw[3] = {};
#pragma unroll
  for (int q = 0; q < NI; ++q) inf[q] = 0.0f;
  if (active) {
    if constexpr (GUIDE == kGuideMap) {
      const float4 g4 = *reinterpret_cast<const float4*>(p.guide + px);
      gs[0] = g4.x; gs[1] = g4.y; gs[2] = g4.z; gs[3] = g4.w;
    }
    if constexpr (NN_MFMA) load_pixels<TI, NI>(input, px * CIN, p.white, inf, raw);
    else load_pixels<TI, NI, FOLD_WL>(input, px * CIN, p.white, inf);
  }
  const bool nn_mfma = NN_MFMA && p.nn_mfma && p.gn.n <= 16 && (p.gn.n & 3) == 0;  // uniform
  if constexpr (NN_MFMA) {
    if (nn_mfma && wave == 0)
      nn_build_tables(reinterpret_cast<unsigned*>(lds_all), lds_all + NnTab::kWords - 4, p.gn, p.white.wl, lane);
  }

  if constexpr (GUIDE == kGuideCurves) {
    if (wave == 0) curves_build_tables<CIN>(ctab, p.gn, lane);  // published by the barrier below
  }
  if constexpr (GUIDE == kGuideCurvesCells) {  // the prepared cell tables: 3 KB copied by the whole workgroup (barrier below)
    const f32x4* src = reinterpret_cast<const f32x4*>(p.gn.prepared);
    for (int e = tid; e < CIN * kCurveCells; e += (int)blockDim.x) reinterpret_cast<f32x4*>(ctab)[e] = src[e];
  }
  const SegCols sc = seg_cols_tab(p.tab, blockIdx.x, xs, xe, p.scale_x);
  const int colb = (p.GD + 2) * CB;
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);
  stage_image<C, FOLD_WL>(lds, grid_b, y, sc.cmin, sc.ncols, p.GH, p.GW, p.GD, p.scale_y, p.inv_col, tid, (int)blockDim.x,
                          p.white.inv);
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
      if constexpr (GUIDE == kGuideNN) {
        bool done = false;
        if constexpr (NN_MFMA) {
          if (nn_mfma) {
            guide_nn_quad_mfma_u8(reinterpret_cast<const unsigned*>(lds_all), p.gn, raw, lane, gs);
            done = true;
          }
        }
        if (!done) guide_nn_quad<CIN>(GuideNN{p.gn.conv1, p.gn.conv2, p.gn.guide_out, p.gn.n, p.gn.fast_sigmoid, p.gn.prescaled}, inf, gs);  // (writes nothing itself)
      }
      else if constexpr (GUIDE == kGuideCurves)
        guide_curves_quad<CIN>(ctab, p.gn, inf, gs);
      else if constexpr (GUIDE == kGuideCurvesCells)
        guide_curves_quad_cells<CIN>(ctab, p.gn, inf, gs);
      else
        guide_curves_scan_quad<CIN>(p.gn, inf, gs);
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
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc_uniform(oseg, (unsigned)(xe - xs) * COUT * 4u);
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
  return WhiteLevel{wl, (all_ones || !in_range) ? 0.0f : (float)r, (float)r};
}

struct IoGeom {
  Plan pl;
  int slab_off;
  size_t lds;
};

IoGeom io_geom(int W, int GW, int GD, int C, int Cout, int tab_floats = 0) {
  IoGeom g;
  g.pl = make_row_plan(W, GW, true);
  const int max_cols = (int)(((long long)(g.pl.seg - 1) * GW) / W + 4);
  g.slab_off = round_up(max_cols * (GD + 2) * C, 4);
  g.lds = ((size_t)tab_floats + (size_t)g.slab_off + (size_t)(g.pl.threads / 64) * 64 * kPxPerThread * Cout) * sizeof(float);
  return g;
}

template <int CIN, int COUT, bool OFFSET, int GUIDE, typename TI, typename TO>
hipError_t launch_io(const ApplyIoArgs& a, const Plan&, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const IoGeom g = io_geom(a.W, a.GW, a.GD, C, COUT, (kToolsBuild && GUIDE == kGuideNN) ? NnTab::kWords : 0);
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
  p.gn = GuideNet{a.guide_conv1, a.guide_conv2, a.guide_shifts, a.guide_slopes, a.guide_out, a.n_feats, a.fast_sigmoid, a.guide_prescaled,
                     (a.guide_shifts && a.n_feats <= kCurveMaxKnots) ? a.guide_prepared : nullptr};
#ifdef HDRNET_TOOLS_BUILD
  p.nn_mfma = tools_knob(5);
#else
  p.nn_mfma = 0;
#endif
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
  const IoGeom g = io_geom(a.W, a.GW, a.GD, 12, 3, CurveTab<3>::kFloats > 3 * kCurveCells * 4 ? CurveTab<3>::kFloats : 3 * kCurveCells * 4);  // + the curves kernel's static tables: the largest of the kernels
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

size_t curves_guide_prepared_bytes(int Cin) { return Cin == 3 ? CurveCells<3>::kFloats * sizeof(float) : 0; }
size_t curves_guide_prepared_ok_offset(int Cin) { return Cin == 3 ? (size_t)CurveCells<3>::kOk : 0; }

hipError_t launch_curves_guide_prepare(const float* shifts, const float* slopes, int npts, int Cin, float* prepared,
                                       hipStream_t s) {
  if (Cin != 3 || npts > kCurveMaxKnots) return hipErrorInvalidValue;
  const GuideNet gn{nullptr, nullptr, shifts, slopes, nullptr, npts, false, false, nullptr};
  curves_prepare_kernel<3><<<1, 256, 0, s>>>(gn, prepared);
  return hipGetLastError();
}

hipError_t launch_apply_fwd_io(const ApplyIoArgs& a, hipStream_t s, const char** name) {
  Plan pl;
  if (!plan_io(a, &pl)) return hipErrorInvalidValue;
  static const char* const io[3][2] = {{"f32->f32", "f32->u8"}, {"u8->f32", "u8->u8"}, {"u16->f32", "u16->u8"}};
  static const char* const suffix[3] = {"", "+nnguide", "+curvesguide"};
  static thread_local char label[64];
  const int kind = a.guide ? kGuideMap : (a.guide_shifts ? kGuideCurves : kGuideNN);
  const bool cells = kind == kGuideCurves && a.guide_prepared && a.n_feats <= kCurveMaxKnots;
  snprintf(label, sizeof label, "apply_fwd_io/%s%s%s", io[a.input_dtype][a.output_dtype], suffix[kind], cells ? "/cells" : "");
  *name = label;
  if (kind == kGuideCurves) {
    if (a.n_feats > kCurveMaxKnots) return dispatch_types<kGuideCurvesScan>(a, pl, s);
    return cells ? dispatch_types<kGuideCurvesCells>(a, pl, s) : dispatch_types<kGuideCurves>(a, pl, s);
  }
  return kind == kGuideNN ? dispatch_types<kGuideNN>(a, pl, s) : dispatch_types<kGuideMap>(a, pl, s);
}

}  // namespace hdrnet_amd
