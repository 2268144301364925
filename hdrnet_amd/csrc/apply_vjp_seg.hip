// BilateralSliceApply per-pixel VJPs (dguide, dinput) WITHOUT the grid gradient, on the product forward's
// core (apply_fwd_seg.hip / seg_common.hip.h).
//
// Reference semantics: BilateralSliceApplyGuideGrad / InputGrad, hdrnet/ops/bilateral_slice_apply.cc:140-259
// (one kernel per VJP in the CUDA twin, each re-gathering the 8 grid corners).  When dgrid is requested too
// the fused pass of grid_grad_mfma.hip produces all three; this kernel serves the calls that want only the
// per-pixel gradients (and the shapes the fused pass does not cover).  It is the forward's design applied to
// a 44 B/px stream (28 in, 16 out): a workgroup owns a row segment, the padded y-pre-lerped coefficient
// image in LDS, guide / input / dout streamed in by nontemporal LDS-DMA (7 KiB per wave), a pixel's four
// coefficient vectors at one address + immediate offsets feeding both blends (weights and their guide
// derivatives, rows_common.hip.h: vjp_blend), dguide stored lane-contiguously as it is and dinput through
// the per-wave LDS transpose, both as nontemporal buffer stores.  The smoothed tent and its derivative use
// v_sqrt_f32 / v_rcp_f32 exactly as the fused pass does (grid_grad_mfma.hip; the reference's
// `abs_dx > 1 ? 0 : dx / abs_dx` is kept for wild guides).
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
#include "seg_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

struct VjpSegParams {
  const float *grid, *guide, *input, *dout;
  float *dguide, *dinput;
  int H, W, GH, GW, GD;
  int seg, slab_off;
  float scale_x, scale_y, inv_col;
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16_nt(const float* src, float* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst_wave_base, 16, 0, 2);
}

template <int CIN, int COUT, bool OFFSET, bool WANT_GUIDE, bool WANT_INPUT>
__global__ __launch_bounds__(256) void apply_vjp_seg(const VjpSegParams p) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  constexpr int RUN = 64 * kPxPerThread;            // pixels of a wave's run
  constexpr int SLAB = RUN * (1 + CIN + COUT);      // floats per wave: guide | input (-> dinput) | dout
  constexpr bool kZDiff = C % 4 == 0 && CIN >= 1;   // coefficient vectors of whole float4s: the z-difference form
  constexpr bool kRowVec = CJ == 4;                 // ... and one float4 = one output row (packed pairs); else by element
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xs = blockIdx.x * p.seg;
  const int xe = min(xs + p.seg, p.W);
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  const float* grid_b = p.grid + (size_t)b * p.GH * p.GW * p.GD * C;
  const int x = xs + kPxPerThread * tid;
  const bool active = x < xe;
  const int wave_x0 = xs + RUN * wave;
  const int wave_px = min(xe, wave_x0 + RUN) - wave_x0;  // <= 0: idle wave
  float4* gslab = reinterpret_cast<float4*>(lds + p.slab_off + wave * SLAB);
  float4* islab = gslab + RUN / 4;
  float4* dslab = islab + RUN * CIN / 4;

  const size_t row = (size_t)b * p.H + y;
  const unsigned wpx = (unsigned)(RUN * wave);
  const float* gseg = p.guide + (row * p.W + xs);
  const float* iseg = p.input + (row * p.W + xs) * CIN;
  const float* dseg = p.dout + (row * p.W + xs) * COUT;
  if (wave_px > 0) {  // every lane issues; lanes past the run re-read its last float4
    dma16_nt(gseg + (wpx + 4u * (unsigned)min(lane, wave_px / 4 - 1)), reinterpret_cast<float*>(gslab));
    const int ilast = wave_px * CIN / 4 - 1, dlast = wave_px * COUT / 4 - 1;
#pragma unroll
    for (int k = 0; k < CIN; ++k)
      dma16_nt(iseg + (wpx * CIN + 4u * (unsigned)min(lane + 64 * k, ilast)), reinterpret_cast<float*>(islab + 64 * k));
#pragma unroll
    for (int k = 0; k < COUT; ++k)
      dma16_nt(dseg + (wpx * COUT + 4u * (unsigned)min(lane + 64 * k, dlast)), reinterpret_cast<float*>(dslab + 64 * k));
  }

  const SegCols sc = seg_cols(xs, xe, p.scale_x);
  const int colb = (p.GD + 2) * CB;
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);
  stage_image<C>(lds, grid_b, y, sc.cmin, sc.ncols, p.GH, p.GW, p.GD, p.scale_y, p.inv_col, tid, (int)blockDim.x);
  XTerm xt[kPxPerThread];
  const float xf0 = (float)x + 0.5f;
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) xt[k] = x_term(xf0 + (float)k, p.scale_x, sc.cmin, colb, CB);
  __syncthreads();  // image complete; the compiler drains the LDS-DMA (vmcnt) here too

  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 iv[CIN], dv[COUT];
  if (active) {
    g4 = gslab[lane];
#pragma unroll
    for (int q = 0; q < CIN; ++q) iv[q] = islab[lane * CIN + q];
#pragma unroll
    for (int q = 0; q < COUT; ++q) dv[q] = dslab[lane * COUT + q];
  }
  const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
  const float* inf = reinterpret_cast<const float*>(iv);
  const float* df = reinterpret_cast<const float*>(dv);
  float4 dgv = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 div[CIN];
  float* dgf = reinterpret_cast<float*>(&dgv);
  float* dif = reinterpret_cast<float*>(div);
  if (active) {
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      float in[CIN], d[COUT], di[CIN];
#pragma unroll
      for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
#pragma unroll
      for (int i = 0; i < COUT; ++i) d[i] = df[k * COUT + i];
      float dgk = 0.0f;
      if constexpr (kZDiff) {
        // The fused pass's evaluation (grid_grad_mfma.hip, "dguide from the z difference"), streamed one float4 of the
        // four corner vectors at a time: with X_t = the x-blended vector of z tap t (A_t + wx1 (B_t - A_t)) and
        // U = dout_i [in; 1],
        //   dguide = dw1 <X1 - X0, U> + (dw0 + dw1) <X0, U>,  dw0 + dw1 = GD eps (D_b - D_a) / (D_a D_b), D = s (s + |dz|)
        //   dinput = (wz0 + wz1) T(X0) + wz1 T(X1 - X0),      T(G)_j = sum_i dout_i G[i, j]
        // -- no subtraction of two derivative weights near -+GD and none of two contracted taps of magnitude ~10: the
        // cancellation happens per coefficient before the contraction.  (Round 5 wrote the same algebra on whole
        // CoefVec taps: 205 VGPRs.  Here a tap never exists as a whole: 4 reads -> 2 vectors -> 4 + 4 scalars.)
        float wz0, wz1, dw1 = 0.0f, dwsum = 0.0f;
        int a0;
        {
#pragma clang fp contract(off)
          const float gzf = mul_rn(gs[k], gd_f);
          const float fzl = floorf(gzf - 0.5f);
          const float dza = (fzl + 0.5f) - gzf, dzb = ((fzl + 1.0f) + 0.5f) - gzf;
          const float qza = __builtin_fmaf(dza, dza, kSmoothEps), qzb = __builtin_fmaf(dzb, dzb, kSmoothEps);
          float sza, szb;
          if constexpr (WANT_GUIDE) {
            // smoothed |dz| and its reciprocal from ONE v_rsq_f32 per tap: s = q rsq(q) (1.5 ulp; rsq(1.0f) is exact)
            const float rza = __builtin_amdgcn_rsqf(qza), rzb = __builtin_amdgcn_rsqf(qzb);
            sza = qza * rza;
            szb = qzb * rzb;
            // GD * SmoothedLerpWeightGrad (:186-187); the s > 1 branch (decided on q = s^2: grid_grad_mfma.hip's note)
            // binds only for wild guides -- |guide * GD| beyond 2^23, where a tap's offset rounds to 2 -- and is taken
            // per lane with selects: a branch on a ballot inside this unrolled loop costs the kernel 120 registers
            dw1 = (qzb > 1.0f) ? 0.0f : gd_f * (dzb * rzb);
            const float dw0 = (qza > 1.0f) ? 0.0f : gd_f * (dza * rza);
            const float Da = sza * (sza + fabsf(dza)), Db = szb * (szb + fabsf(dzb));
            dwsum = (gd_f * kSmoothEps) * ((Db - Da) * __builtin_amdgcn_rcpf(Da * Db));
            dwsum = (qza > 1.0f || qzb > 1.0f) ? dw0 + dw1 : dwsum;
          } else {
            sza = __builtin_amdgcn_sqrtf(qza);
            szb = __builtin_amdgcn_sqrtf(qzb);
          }
          wz0 = __builtin_amdgcn_fmed3f(1.0f - sza, 0.0f, 1.0f);  // max(1 - s, 0): s > 0
          wz1 = __builtin_amdgcn_fmed3f(1.0f - szb, 0.0f, 1.0f);
          const int iz = (int)__builtin_amdgcn_fmed3f(fzl, -1.0f, zhi);
          a0 = __mul24(iz, CB) + xt[k].xbp;
        }
        constexpr int NQ = C / 4;  // float4 q = row i of [COUT][CJ = 4]
        const f32x2 i01 = {in[0], CIN > 1 ? in[CIN > 1 ? 1 : 0] : 1.0f};
        const f32x2 i23 = {CIN > 2 ? in[CIN > 2 ? 2 : 0] : 1.0f, CIN > 3 ? in[CIN > 3 ? 3 : 0] : 1.0f};
        const f32x4 wx1 = {xt[k].wx1, xt[k].wx1, xt[k].wx1, xt[k].wx1};
        const char* ibase = reinterpret_cast<const char*>(lds) + a0;
        f32x2 acc0 = {0.0f, 0.0f}, accd = {0.0f, 0.0f};                       // <X0, U>, <X1 - X0, U> two columns at a time
        f32x2 t0a = {0.0f, 0.0f}, t0b = {0.0f, 0.0f}, tda = {0.0f, 0.0f}, tdb = {0.0f, 0.0f};  // T(X0), T(X1 - X0)
        [[maybe_unused]] float t0s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, tds[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the same, by element
        f32x4 nA0 = *reinterpret_cast<const f32x4*>(ibase), nA1 = *reinterpret_cast<const f32x4*>(ibase + CB);
        f32x4 nB0 = *reinterpret_cast<const f32x4*>(ibase + colb), nB1 = *reinterpret_cast<const f32x4*>(ibase + colb + CB);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 X0 = __builtin_elementwise_fma(wx1, nB0 - nA0, nA0);
          const f32x4 X1 = __builtin_elementwise_fma(wx1, nB1 - nA1, nA1);
          const f32x4 Xd = X1 - X0;
          if (q + 1 < NQ) {
            const int o = (q + 1) * 16;
            nA0 = *reinterpret_cast<const f32x4*>(ibase + o);
            nA1 = *reinterpret_cast<const f32x4*>(ibase + CB + o);
            nB0 = *reinterpret_cast<const f32x4*>(ibase + colb + o);
            nB1 = *reinterpret_cast<const f32x4*>(ibase + colb + CB + o);
          }
          if constexpr (!kRowVec) {
            // (4 -> 4 with offset, C = 20: a float4 straddles output rows) element c = 4 q + e is coefficient (i, j) =
            // (c / CJ, c % CJ), known at compile time after unrolling
            const float x0[4] = {X0.x, X0.y, X0.z, X0.w}, xd[4] = {Xd.x, Xd.y, Xd.z, Xd.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = 4 * q + e, i = c / CJ, j = c % CJ;
              if constexpr (WANT_GUIDE) {
                const float u = j < CIN ? d[i] * in[j < CIN ? j : 0] : d[i];
                if (e & 1) {
                  acc0.y = fmaf(x0[e], u, acc0.y);
                  accd.y = fmaf(xd[e], u, accd.y);
                } else {
                  acc0.x = fmaf(x0[e], u, acc0.x);
                  accd.x = fmaf(xd[e], u, accd.x);
                }
              }
              if constexpr (WANT_INPUT) {
                if (j < CIN) {
                  t0s[j < 4 ? j : 0] = fmaf(x0[e], d[i], t0s[j < 4 ? j : 0]);
                  tds[j < 4 ? j : 0] = fmaf(xd[e], d[i], tds[j < 4 ? j : 0]);
                }
              }
            }
          } else {
            // one float4 = output row q: its two column pairs against U = dout_q [in; 1], packed
            const float dqs = d[q < COUT ? q : 0];
            const f32x2 dq = {dqs, dqs};
            if constexpr (WANT_GUIDE) {
              const f32x2 U01 = i01 * dq, U23 = i23 * dq;
              acc0 = __builtin_elementwise_fma(f32x2{X0.x, X0.y}, U01, acc0);
              acc0 = __builtin_elementwise_fma(f32x2{X0.z, X0.w}, U23, acc0);
              accd = __builtin_elementwise_fma(f32x2{Xd.x, Xd.y}, U01, accd);
              accd = __builtin_elementwise_fma(f32x2{Xd.z, Xd.w}, U23, accd);
            }
            if constexpr (WANT_INPUT) {
              t0a = __builtin_elementwise_fma(f32x2{X0.x, X0.y}, dq, t0a);
              tda = __builtin_elementwise_fma(f32x2{Xd.x, Xd.y}, dq, tda);
              if constexpr (CIN > 3) {
                t0b = __builtin_elementwise_fma(f32x2{X0.z, X0.w}, dq, t0b);
                tdb = __builtin_elementwise_fma(f32x2{Xd.z, Xd.w}, dq, tdb);
              } else if constexpr (CIN > 2) {
                t0b.x = fmaf(X0.z, dqs, t0b.x);
                tdb.x = fmaf(Xd.z, dqs, tdb.x);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // one vector of read-ahead: a whole tap in registers costs occupancy
        }
        if constexpr (WANT_GUIDE) dgk = fmaf(dw1, accd.x + accd.y, dwsum * (acc0.x + acc0.y));
        if constexpr (WANT_INPUT) {
          const float wzs = wz0 + wz1;
          const float t0[4] = {t0a.x, t0a.y, t0b.x, t0b.y}, td[4] = {tda.x, tda.y, tdb.x, tdb.y};
#pragma unroll
          for (int j = 0; j < CIN; ++j)
            di[j] = kRowVec ? fmaf(wz1, td[j], wzs * t0[j]) : fmaf(wz1, tds[j < 4 ? j : 0], wzs * t0s[j < 4 ? j : 0]);
        }
      } else {
      // z terms as the forward forms them (seg_common.hip.h: seg_pixel) + the tent's derivative
      float wz0, wz1, dw0, dw1;
      int a0;
      {
#pragma clang fp contract(off)
        const float gzf = mul_rn(gs[k], gd_f);
        const float fzl = floorf(gzf - 0.5f);
        const float dza = (fzl + 0.5f) - gzf, dzb = ((fzl + 1.0f) + 0.5f) - gzf;
        const float sza = __builtin_amdgcn_sqrtf(fmaf(dza, dza, kSmoothEps));
        const float szb = __builtin_amdgcn_sqrtf(fmaf(dzb, dzb, kSmoothEps));
        wz0 = std_max(1.0f - sza, 0.0f);
        wz1 = std_max(1.0f - szb, 0.0f);
        // GD * SmoothedLerpWeightGrad (:186-187); the s > 1 branch binds only for wild guides
        dw0 = (sza > 1.0f) ? 0.0f : gd_f * (dza * __builtin_amdgcn_rcpf(sza));
        dw1 = (szb > 1.0f) ? 0.0f : gd_f * (dzb * __builtin_amdgcn_rcpf(szb));
        const int iz = (int)__builtin_amdgcn_fmed3f(fzl, -1.0f, zhi);
        a0 = __mul24(iz, CB) + xt[k].xbp;
      }
      vjp_blend<CIN, COUT, OFFSET, WANT_GUIDE, WANT_INPUT>(lds, a0, a0 + CB, a0 + colb, a0 + colb + CB, xt[k].wx0,
                                                          xt[k].wx1, wz0, wz1, dw0, dw1, in, d, dgk, di);
      }
      dgf[k] = dgk;
      if constexpr (WANT_INPUT) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) dif[k * CIN + j] = di[j];
      }
    }
    if constexpr (WANT_GUIDE)  // descriptor over the row segment (wave-uniform base), lane offset 16 B * tid
      buf_store16<kAuxStream>(dgv, make_rsrc_uniform(p.dguide + (row * p.W + xs), (unsigned)(xe - xs) * 4u), 16u * (unsigned)tid);
    if constexpr (WANT_INPUT) {
      // in place: a lane overwrites only ITS input entries of the slab (read above)
#pragma unroll
      for (int q = 0; q < CIN; ++q) islab[lane * CIN + q] = div[q];
    }
  }
  if constexpr (WANT_INPUT) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc_uniform(p.dinput + (row * p.W + xs) * CIN, (unsigned)(xe - xs) * CIN * 4u);
#pragma unroll
    for (int k = 0; k < CIN; ++k)
      buf_store16<kAuxStream>(islab[lane + 64 * k], orsrc, (wpx * CIN + 4u * (unsigned)(lane + 64 * k)) * 4u);
  }
}

constexpr size_t kMaxLdsBytes = 64 * 1024;

struct VjpGeom {
  Plan pl;
  int slab_off;
  size_t lds;
  bool ok;
};

VjpGeom vjp_geom(const ApplyGradArgs& a) {
  const int C = a.Cout * a.Cj;
  VjpGeom g{};
  const bool aligned = (((uintptr_t)a.guide | (uintptr_t)a.input | (uintptr_t)a.dout | (uintptr_t)a.grid |
                         (uintptr_t)a.dguide | (uintptr_t)a.dinput) & 15u) == 0;
  g.pl = make_row_plan(a.W, a.GW, aligned);
  const int max_cols = (int)(((long long)(g.pl.seg - 1) * a.GW) / a.W + 4);
  g.slab_off = round_up(max_cols * (a.GD + 2) * C, 4);
  g.lds = ((size_t)g.slab_off + (size_t)(g.pl.threads / 64) * 64 * kPxPerThread * (1 + a.Cin + a.Cout)) * sizeof(float);
  g.ok = g.pl.vec4 && g.lds <= kMaxLdsBytes && (long long)max_cols * a.GD * C < (1 << 20) && a.B <= 65535 &&
         a.H <= 65535 && (long long)a.W * (a.Cin > a.Cout ? a.Cin : a.Cout) * 4 < (1LL << 31);
  return g;
}

template <int CIN, int COUT, bool OFFSET, bool WG, bool WI>
hipError_t launch_t(const ApplyGradArgs& a, const VjpGeom& g, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  VjpSegParams p{a.grid, a.guide, a.input, a.dout, a.dguide, a.dinput, a.H, a.W, a.GH, a.GW, a.GD,
                 g.pl.seg, g.slab_off, (float)a.GW / a.W, (float)a.GH / a.H, 1.0f / (float)(a.GD * (C / VEC))};
  const dim3 grid3((unsigned)g.pl.nseg, (unsigned)a.H, (unsigned)a.B);
  apply_vjp_seg<CIN, COUT, OFFSET, WG, WI><<<grid3, g.pl.threads, g.lds, s>>>(p);
  return hipGetLastError();
}

template <int CIN, int COUT, bool OFFSET>
hipError_t launch_want(const ApplyGradArgs& a, const VjpGeom& g, hipStream_t s) {
  const bool wg = a.dguide != nullptr, wi = a.dinput != nullptr;
  if (wg && wi) return launch_t<CIN, COUT, OFFSET, true, true>(a, g, s);
  if (wg) return launch_t<CIN, COUT, OFFSET, true, false>(a, g, s);
  if (wi) return launch_t<CIN, COUT, OFFSET, false, true>(a, g, s);
  return hipSuccess;
}

}  // namespace

bool apply_vjp_seg_supported(const ApplyGradArgs& a) {
  const bool shape = apply_fast_shape(a.Cin, a.Cout, a.has_offset);
  if (!shape || !a.guide || !a.input || !a.dout) return false;
  if ((a.Cout * a.Cj) % 4 == 0 && ((uintptr_t)a.grid & 15u)) return false;  // stage_image reads float4
  return vjp_geom(a).ok;
}

hipError_t launch_apply_vjp_seg(const ApplyGradArgs& a, hipStream_t s, const char** name) {
  const VjpGeom g = vjp_geom(a);
  if (!g.ok) return hipErrorNotSupported;
  *name = "apply_vjp_seg/vec4";
#define HDRNET_CASE(CI, CO, OFF) \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) return launch_want<CI, CO, OFF>(a, g, s);
  HDRNET_APPLY_FAST_SHAPES(HDRNET_CASE)
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

}  // namespace hdrnet_amd
