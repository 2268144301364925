// BilateralSliceApply / BilateralSlice per-pixel VJPs for gfx950: guide and input
// gradients in ONE pass over the pixels.
//
// Reference semantics (the CPU code; the CUDA twins carry two indexing bugs):
//   BilateralSliceApplyGuideGrad  hdrnet/ops/bilateral_slice_apply.cc:140-206
//   BilateralSliceApplyInputGrad  hdrnet/ops/bilateral_slice_apply.cc:208-259
//   BilateralSliceGuideGrad       hdrnet/ops/bilateral_slice.cc:120-168
// The reference launches one kernel per VJP (bilateral_slice_apply.cu.cc:384-417), each
// re-gathering the same 8 grid corners (96 + 24 scattered loads per pixel).  Both VJPs need
// exactly the four (gx, gz) vectors the forward kernel gathers from its y-pre-lerped LDS
// image, once with the z-tent weights (-> sliced coefficients A_ij, dinput_j = sum_i
// dout_i A_ij) and once with the tent's derivative GD * d wz / d gz (-> dA_ij,
// dguide = sum_i dout_i (sum_j dA_ij in_j + dA_i,offset)).  So one kernel, same workgroup
// geometry and LDS image as the forward (rows_common.hip.h), reads guide/input/dout once
// (28 B/px) and writes dguide + dinput (16 B/px); each ds_read_b128 feeds two v_pk_fma_f32.
//
// The smoothed |.| uses IEEE sqrtf here: its value is compared with 1 and divided by
// (numerics.h:116-126), and the reference's branch `abs_dx > 1 ? 0 : dx / abs_dx` is kept
// verbatim so that a guide sitting exactly on a bin centre picks the same side.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

// One pixel.  APPLY: dout has COUT channels, grid channel c = i*CJ + j.
// !APPLY (BilateralSlice): CIN = 0, CJ = 1, COUT = C, dguide = sum_c dout_c dA_c.
template <int CIN, int COUT, bool OFFSET, bool WANT_GUIDE, bool WANT_INPUT>
__device__ __forceinline__ void vjp_pixel(const RowCtx& r, float xf, float g,
                                          const float* __restrict__ in,   // [CIN]
                                          const float* __restrict__ d,    // [COUT]
                                          float& dguide, float* __restrict__ dinput) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  const SliceTerms t = slice_terms<C, true>(r, xf, g);
  // GD * SmoothedLerpWeightGrad(gz + .5, gzf)   (bilateral_slice_apply.cc:186-187)
  const float dw0 = r.gd_f * ((t.sz0 > 1.0f) ? 0.0f : t.dz0 / t.sz0);
  const float dw1 = r.gd_f * ((t.sz1 > 1.0f) ? 0.0f : t.dz1 / t.sz1);
  vjp_blend<CIN, COUT, OFFSET, WANT_GUIDE, WANT_INPUT>(r.colY, t.a00, t.a01, t.a10, t.a11, t.wx0, t.wx1,
                                                       t.wz0, t.wz1, dw0, dw1, in, d, dguide, dinput);
}

// Same geometry as apply_fwd_rows_vec4: a workgroup owns a segment of one image row, a
// thread 4 consecutive pixels; loads are per-lane 16-B vectors issued before the staging
// pass; dinput leaves through the per-wave LDS transpose (lane-contiguous stores), dguide
// is lane-contiguous as it is.
template <int CIN, int COUT, bool OFFSET, bool WANT_GUIDE, bool WANT_INPUT>
__global__ __launch_bounds__(256) void apply_vjp_rows_vec4(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, const float* __restrict__ dout,
    float* __restrict__ dguide, float* __restrict__ dinput, int H, int W, int GH, int GW,
    int GD, int nseg, int seg, int slab_offset_floats, float scale_x, float scale_y) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr bool kNeedIn = WANT_GUIDE && CIN > 0;
  constexpr int CIN_Q = CIN > 0 ? CIN : 1;
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

  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 iv[CIN_Q];
  float4 dv[COUT];
  if (active) {
    g4 = *reinterpret_cast<const float4*>(guide + p);
    if constexpr (kNeedIn) {
      const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
      for (int q = 0; q < CIN; ++q) iv[q] = ip[q];
    }
    const float4* dp = reinterpret_cast<const float4*>(dout + p * COUT);
#pragma unroll
    for (int q = 0; q < COUT; ++q) dv[q] = dp[q];
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);

  const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
  const float xf0 = (float)x + 0.5f;
  const float* inf = reinterpret_cast<const float*>(iv);
  const float* df = reinterpret_cast<const float*>(dv);
  float4 dgv = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 div[CIN_Q];
  float* dgf = reinterpret_cast<float*>(&dgv);
  float* dif = reinterpret_cast<float*>(div);
  if (active) {
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      float in[CIN_Q], d[COUT], di[CIN_Q];
      if constexpr (kNeedIn) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
      }
#pragma unroll
      for (int i = 0; i < COUT; ++i) d[i] = df[k * COUT + i];
      float dgk = 0.0f;
      vjp_pixel<CIN, COUT, OFFSET, WANT_GUIDE, WANT_INPUT>(r, xf0 + (float)k, gs[k], in, d, dgk, di);
      dgf[k] = dgk;
      if constexpr (WANT_INPUT) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) dif[k * CIN + j] = di[j];
      }
    }
    if constexpr (WANT_GUIDE)  // descriptor over the row segment (wave-uniform base), lane offset 16 B * tid
      buf_store16<kAuxStream>(dgv, make_rsrc_uniform(dguide + ((size_t)row * W + xs), (unsigned)(xe - xs) * 4u),
                              16u * threadIdx.x);
  }
  if constexpr (WANT_INPUT) {
    // lane-contiguous stores through the per-wave LDS slab (see apply_fwd_rows.hip)
    float4* slab = reinterpret_cast<float4*>(colY + slab_offset_floats) + (threadIdx.x >> 6) * (64 * CIN);
    const int lane = threadIdx.x & 63;
    if (active) {
#pragma unroll
      for (int q = 0; q < CIN; ++q) slab[lane * CIN + q] = div[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int wave_x0 = xs + kPxPerThread * (int)(threadIdx.x & ~63u);
    const int nvalid = (min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0) * CIN / 4;
    // nontemporal buffer stores on a descriptor over exactly this wave's run (rows_common.hip.h)
    const __amdgpu_buffer_rsrc_t orsrc =
        make_rsrc_uniform(dinput + ((size_t)row * W + wave_x0) * CIN, nvalid > 0 ? (unsigned)nvalid * 16u : 0u);
#pragma unroll
    for (int k = 0; k < CIN; ++k) buf_store16<kAuxStream>(slab[lane + 64 * k], orsrc, (unsigned)(lane + 64 * k) * 16u);
  }
}

constexpr size_t kMaxLdsBytes = 64 * 1024;

struct VjpShape {
  const float *grid, *guide, *input, *dout;
  float *dguide, *dinput;
  int B, H, W, GH, GW, GD, Cin, Cout, Cj;
};

template <int CIN, int COUT, bool OFFSET, bool WG, bool WI>
hipError_t launch_vjp_t(const VjpShape& a, const Plan& pl, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const int slab_off = round_up(pl.max_cols * a.GD * C, 4);
  const size_t lds =
      ((size_t)slab_off + (WI ? (size_t)(pl.threads / 64) * 64 * kPxPerThread * CIN : 0)) * sizeof(float);
  const long long nblocks = (long long)a.B * a.H * pl.nseg;
  apply_vjp_rows_vec4<CIN, COUT, OFFSET, WG, WI><<<(unsigned)nblocks, pl.threads, lds, s>>>(
      a.grid, a.guide, a.input, a.dout, a.dguide, a.dinput, a.H, a.W, a.GH, a.GW, a.GD, pl.nseg,
      pl.seg, slab_off, (float)a.GW / a.W, (float)a.GH / a.H);
  return hipGetLastError();
}

template <int CIN, int COUT, bool OFFSET>
hipError_t launch_vjp_want(const VjpShape& a, const Plan& pl, hipStream_t s) {
  const bool wg = a.dguide != nullptr, wi = (a.dinput != nullptr) && CIN > 0;
  if (wg && wi) return launch_vjp_t<CIN, COUT, OFFSET, true, (CIN > 0)>(a, pl, s);
  if (wg) return launch_vjp_t<CIN, COUT, OFFSET, true, false>(a, pl, s);
  if (wi) return launch_vjp_t<CIN, COUT, OFFSET, false, (CIN > 0)>(a, pl, s);
  return hipSuccess;
}

bool vjp_plan(const VjpShape& a, Plan* pl) {
  const bool aligned = (((uintptr_t)a.guide | (uintptr_t)a.input | (uintptr_t)a.dout |
                         (uintptr_t)a.grid | (uintptr_t)a.dguide | (uintptr_t)a.dinput) & 15u) == 0;
  *pl = make_row_plan(a.W, a.GW, aligned);
  if (!pl->vec4) return false;
  if ((long long)a.B * a.H * pl->nseg > 0x7fffffffLL) return false;
  const size_t lds = ((size_t)pl->max_cols * a.GD * a.Cout * a.Cj + 4 +
                      (size_t)(pl->threads / 64) * 64 * kPxPerThread * (a.Cin > 0 ? a.Cin : 1)) *
                     sizeof(float);
  return lds <= kMaxLdsBytes;
}

}  // namespace

// ---- BilateralSliceApply: dguide / dinput -------------------------------------------------
bool apply_vjp_rows_supported(const ApplyGradArgs& a) {
  const bool shape = apply_fast_shape(a.Cin, a.Cout, a.has_offset);
  if (!shape) return false;
  VjpShape v{a.grid, a.guide, a.input, a.dout, a.dguide, a.dinput, a.B, a.H, a.W,
             a.GH, a.GW, a.GD, a.Cin, a.Cout, a.Cj};
  Plan pl;
  return vjp_plan(v, &pl);
}

hipError_t launch_apply_vjp_rows(const ApplyGradArgs& a, hipStream_t s, const char** name) {
#ifdef HDRNET_TOOLS_BUILD
  const bool round1_kernel = a.variant == 11;  // A/B: the round-1 kernel below
#else
  const bool round1_kernel = false;
#endif
  if (!round1_kernel && apply_vjp_seg_supported(a)) {
    const hipError_t e = launch_apply_vjp_seg(a, s, name);
    if (e != hipErrorNotSupported) return e;
  }
  VjpShape v{a.grid, a.guide, a.input, a.dout, a.dguide, a.dinput, a.B, a.H, a.W,
             a.GH, a.GW, a.GD, a.Cin, a.Cout, a.Cj};
  Plan pl;
  if (!vjp_plan(v, &pl)) return hipErrorInvalidValue;
  *name = "apply_vjp_rows/vec4";
#define HDRNET_CASE(CI, CO, OFF) \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) return launch_vjp_want<CI, CO, OFF>(v, pl, s);
  HDRNET_APPLY_FAST_SHAPES(HDRNET_CASE)
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

// ---- BilateralSlice: dguide (CIN = 0, one "offset" column per channel) ------------------------
bool slice_vjp_rows_supported(const SliceGradArgs& a) {
  if (!(a.C == 1 || a.C == 2 || a.C == 4 || a.C == 8 || a.C == 12 || a.C == 16)) return false;
  VjpShape v{a.grid, a.guide, nullptr, a.dout, a.dguide, nullptr, a.B, a.H, a.W,
             a.GH, a.GW, a.GD, 0, a.C, 1};
  Plan pl;
  return vjp_plan(v, &pl);
}

hipError_t launch_slice_vjp_rows(const SliceGradArgs& a, hipStream_t s, const char** name) {
  VjpShape v{a.grid, a.guide, nullptr, a.dout, a.dguide, nullptr, a.B, a.H, a.W,
             a.GH, a.GW, a.GD, 0, a.C, 1};
  Plan pl;
  if (!vjp_plan(v, &pl)) return hipErrorInvalidValue;
  *name = "slice_vjp_rows/vec4";
#define HDRNET_CASE(CC) \
  if (a.C == CC) return launch_vjp_t<0, CC, true, true, false>(v, pl, s)
  HDRNET_CASE(1);
  HDRNET_CASE(2);
  HDRNET_CASE(4);
  HDRNET_CASE(8);
  HDRNET_CASE(12);
  HDRNET_CASE(16);
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

}  // namespace hdrnet_amd
