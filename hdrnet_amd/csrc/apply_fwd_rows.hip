// BilateralSliceApply forward for gfx950 -- the north-star kernel.
//
// Reference semantics: hdrnet/ops/bilateral_slice_apply.cc:24-82 (the CUDA twin,
// bilateral_slice_apply.cu.cc:36-126, assigns one thread per output CHANNEL and
// re-derives all weights for each of its 32 scattered grid loads).
//
// Design: the op is a pure HBM stream -- 4 B guide + 4*Cin B input
// in, 4*Cout B out per pixel -- next to a 96 KiB grid that never leaves L2.  So:
//   * a workgroup owns one SEGMENT OF ONE IMAGE ROW.  For a row, gy0/gy1 and the
//     two y-weights are wave-uniform scalars, so the workgroup first blends the two
//     grid rows it needs into LDS ("y-pre-lerp": colY[gx][gz][c] =
//     wy0*grid[gy0c][gx] + wy1*grid[gy1c][gx], only the gx columns the segment
//     touches).  A pixel then gathers 2(x) x 2(z) coefficient vectors instead of 8.
//   * the LDS image keeps the grid's own [gx][gz][c] order: one (gx,gz) vector is
//     C contiguous floats (48 B for C=12), read as ds_read_b128.  With a 48-B
//     stride the eight gz vectors of a column start at dword banks
//     {0,12,24,36,48,60,8,20} mod 64 -- disjoint 4-bank slots -- so the
//     data-dependent gz gather is conflict-free across a 16-lane b128 group.
//   * a thread owns 4 CONSECUTIVE pixels: guide is one 16-B load, input and output
//     are three 16-B loads / stores each (global_load/store_dwordx4), all issued
//     before the arithmetic; rows are contiguous so every wave streams a dense
//     3 KiB span.  (Scalar variant for W % 4 != 0 or unaligned pointers.)
//   * per pixel: 2 sqrt (the reference: 96), 4 weights, 4*C FMAs for the blend,
//     Cout*Cj FMAs for the affine.  No MFMA: this is a gather/interpolate with 8
//     non-zeros per pixel, not a dense contraction.
//
// Numerics: same coordinate / weight expressions as the reference, evaluated in
// the same order; the only re-association is wy folded into the LDS image, i.e.
// (wy0*g0 + wy1*g1)*(wx*wz) instead of sum((wx*wy)*wz*g).  Differences stay at the
// 1e-7 relative level (tests/test_gpu_parity.py holds rtol=atol=1e-5 and reports
// the reference's own 1e-6 bar).
//
// Benchmark-only alternatives (one wave per tile, persistent stream, memory skeletons) live
// in apply_fwd_variants.hip; DESIGN.md section 4 records why they lost.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
namespace hdrnet_amd {
namespace {

using namespace rows;

// ---- 4 consecutive pixels per thread, 16-byte global accesses -----------------------
// Requires W % 4 == 0, seg % 4 == 0 and 16-B aligned guide / input / out.
//
// LDS_STORES = false keeps the naive per-lane stores (16 B at a 16*COUT-byte lane stride); it
// exists only to document the 6.5 us per 4K frame they cost (tools/ab_bench.py, variant 7).
template <int CIN, int COUT, bool OFFSET, bool GUIDE_NN = false, bool LDS_STORES = true, bool UPADD = false,
          bool NT_LOADS = false>
__global__ __launch_bounds__(256) void apply_fwd_rows_vec4(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, float* __restrict__ out, int H, int W, int GH, int GW,
    int GD, int nseg, int seg, int slab_offset_floats, float scale_x, float scale_y,
    GuideNN gn = GuideNN{nullptr, nullptr, nullptr, 0}, UpAdd up = UpAdd{nullptr, 0, 0, 0.f, 0.f}) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
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

  // Issue this thread's streaming loads before the staging pass so their HBM
  // latency overlaps the (L2-resident) grid reads and the barrier.
  // NT_LOADS (benchmark variant 8): nontemporal, LANE-CONTIGUOUS loads (lane l takes float4 number
  // l + 64k of the wave's input run) + an LDS transpose.  Nontemporal loads lower the no-compute
  // floor of this geometry from 41.2 to 39.0 us (36.2 us stand-alone), but only lane-contiguous:
  // an nt line is not kept for the next instruction, so the per-pixel pattern (three loads that
  // each touch every line of the run) re-fetches and loses 5 us.  The full kernel does not
  // cash it in (44.5 vs 43.6 us): it is bound by bytes in flight -- 10 workgroups per CU cover
  // latency + compute -- not by the memory system's peak.  DESIGN.md section 4.
  const int lane = threadIdx.x & 63;
  const int wave_x0 = xs + kPxPerThread * (int)(threadIdx.x & ~63u);
  const int wave_px = min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0;  // <= 0 for an idle wave
  float4* slab = reinterpret_cast<float4*>(colY + slab_offset_floats) +
                 (threadIdx.x >> 6) * (64 * (CIN > COUT ? CIN : COUT));
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 iv[(CIN * kPxPerThread) / 4];
  if constexpr (LDS_STORES && NT_LOADS) {
    if constexpr (!GUIDE_NN) {
      if (active) g4 = load_stream4(guide + p);
    }
    const float* ibase = input + ((size_t)row * W + wave_x0) * CIN;
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const int e = lane + 64 * k;
      if (e < wave_px * CIN / 4) iv[k] = load_stream4(ibase + 4 * e);
    }
  } else if (active) {
    if constexpr (!GUIDE_NN) g4 = *reinterpret_cast<const float4*>(guide + p);
    const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
    for (int q = 0; q < (CIN * kPxPerThread) / 4; ++q) iv[q] = ip[q];
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);

  if constexpr (LDS_STORES && NT_LOADS) {  // transpose the input run: lane-contiguous -> this lane's 4 pixels
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const int e = lane + 64 * k;
      if (e < wave_px * CIN / 4) slab[e] = iv[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
#pragma unroll
      for (int q = 0; q < CIN; ++q) iv[q] = slab[lane * CIN + q];
    }
    __builtin_amdgcn_wave_barrier();  // the slab is reused for the output below
  }

  float gs[4] = {g4.x, g4.y, g4.z, g4.w};
  const float xf0 = (float)x + 0.5f;  // (x + k) + 0.5f == xf0 + k exactly (x < 2^23)
  const float* inf = reinterpret_cast<const float*>(iv);
  float4 ov[(COUT * kPxPerThread) / 4];
  float* of = reinterpret_cast<float*>(ov);
  if (active) {
    if constexpr (GUIDE_NN) {
      guide_nn_quad<CIN>(gn, inf, gs);
      if (gn.guide_out) *reinterpret_cast<float4*>(gn.guide_out + p) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      float in[CIN], o[COUT];
#pragma unroll
      for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
      slice_apply_pixel<CIN, COUT, OFFSET>(r, xf0 + (float)k, gs[k], in, o);
#pragma unroll
      for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
    }
    if constexpr (UPADD) {
      // row terms are workgroup-uniform; the 2 x 2 x COUT gathers hit the (small, cache-resident)
      // coarse level
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
  }
  if constexpr (!LDS_STORES) {
    if (!active) return;
    float4* op = reinterpret_cast<float4*>(out + p * COUT);
#pragma unroll
    for (int q = 0; q < (COUT * kPxPerThread) / 4; ++q) op[q] = ov[q];
    return;
  }
  // Store phase.  A lane's 4 pixels are 4*COUT contiguous floats, i.e. per-lane 16-B
  // stores at a 16*COUT-byte stride -- measured 6.5 us slower per 4K frame than
  // lane-contiguous stores (the LOADS do not care).  So each wave transposes its tile
  // through a private LDS slab: ds_write_b128 at the per-pixel stride (conflict-free:
  // 12-dword stride over 8-lane groups), then lane l reads float4 number l + 64k and
  // stores it -- every global_store_dwordx4 covers one dense 1 KiB run.
  if (active) {
#pragma unroll
    for (int q = 0; q < COUT; ++q) slab[lane * COUT + q] = ov[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // nontemporal buffer stores on a descriptor over exactly this wave's run (rows_common.hip.h)
  const int nvalid = wave_px * COUT / 4;  // float4s
  const __amdgpu_buffer_rsrc_t orsrc =
      make_rsrc_uniform(out + ((size_t)row * W + wave_x0) * COUT, nvalid > 0 ? (unsigned)nvalid * 16u : 0u);
#pragma unroll
  for (int k = 0; k < COUT; ++k) buf_store16<kAuxStream>(slab[lane + 64 * k], orsrc, (unsigned)(lane + 64 * k) * 16u);
}

// ---- scalar variant: any W / alignment; thread t takes pixels xs + t + k*blockDim ------
template <int CIN, int COUT, bool OFFSET>
__global__ __launch_bounds__(256) void apply_fwd_rows_scalar(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, float* __restrict__ out, int H, int W, int GH, int GW,
    int GD, int nseg, int seg, float scale_x, float scale_y) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  extern __shared__ __attribute__((aligned(16))) float colY[];
  const int bid = blockIdx.x;
  const int segi = bid % nseg;
  const int row = bid / nseg;
  const int y = row % H;
  const int b = row / H;
  const int xs = segi * seg;
  const int xe = min(xs + seg, W);
  const float* grid_b = grid + (size_t)b * GH * GW * GD * C;
  const size_t prow = (size_t)row * W;

  float gs[kPxPerThread];
  float in[kPxPerThread][CIN];
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const int x = xs + threadIdx.x + k * blockDim.x;
    if (x < xe) {
      gs[k] = guide[prow + x];
#pragma unroll
      for (int j = 0; j < CIN; ++j) in[k][j] = input[(prow + x) * CIN + j];
    }
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);

#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const int x = xs + threadIdx.x + k * blockDim.x;
    if (x < xe) {
      float o[COUT];
      slice_apply_pixel<CIN, COUT, OFFSET>(r, (float)x + 0.5f, gs[k], in[k], o);
#pragma unroll
      for (int i = 0; i < COUT; ++i) out[(prow + x) * COUT + i] = o[i];
    }
  }
}

struct LaunchGeom {
  Plan pl;
  int slab_off;
  size_t lds;
  long long nblocks;
};

template <int C, int COUT>
LaunchGeom geom_for(const ApplyArgs& a) {
  const int slab_ch = a.Cin > COUT ? a.Cin : COUT;  // the slab transposes the input run, then the output
  LaunchGeom g;
  const bool aligned = (((uintptr_t)a.guide | (uintptr_t)a.input | (uintptr_t)a.out |
                         (uintptr_t)a.grid) & 15u) == 0;
  g.pl = make_row_plan(a.W, a.GW, aligned);
  // dynamic LDS: [colY image][one 64 x 4*COUT-float output slab per wave (vec4 kernel)]
  g.slab_off = round_up(g.pl.max_cols * a.GD * C, 4);
  g.lds = ((size_t)g.slab_off + (size_t)(g.pl.threads / 64) * 64 * kPxPerThread * slab_ch) * sizeof(float);
  g.nblocks = (long long)a.B * a.H * g.pl.nseg;
  return g;
}

template <int CIN, int COUT, bool OFFSET>
hipError_t launch_t(const ApplyArgs& a, hipStream_t s, const char** name) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const LaunchGeom g = geom_for<C, COUT>(a);
  const float sx = (float)a.GW / a.W, sy = (float)a.GH / a.H;
  if (g.pl.vec4) {
    apply_fwd_rows_vec4<CIN, COUT, OFFSET><<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
        a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, g.slab_off,
        sx, sy);
    *name = "apply_fwd_rows/vec4";
  } else {
    apply_fwd_rows_scalar<CIN, COUT, OFFSET><<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
        a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, sx, sy);
    *name = "apply_fwd_rows/scalar";
  }
  return hipGetLastError();
}

constexpr size_t kMaxLdsBytes = 64 * 1024;  // keep >= 2 workgroups per CU

template <int CIN, int COUT, bool OFFSET>
hipError_t launch_nnguide_t(const ApplyArgs& a, const GuideNN& gn, hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const LaunchGeom g = geom_for<C, COUT>(a);
  apply_fwd_rows_vec4<CIN, COUT, OFFSET, true><<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
      a.grid, nullptr, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, g.slab_off,
      (float)a.GW / a.W, (float)a.GH / a.H, gn);
  return hipGetLastError();
}

template <bool GUIDE_NN>
hipError_t launch_upadd_t(const ApplyArgs& a, const GuideNN& gn, const float* coarse, int Hc, int Wc,
                          hipStream_t s) {
  const LaunchGeom g = geom_for<12, 3>(a);
  const UpAdd up{coarse, Hc, Wc, resize_scale(Hc, a.H), resize_scale(Wc, a.W)};
  apply_fwd_rows_vec4<3, 3, true, GUIDE_NN, true, true>
      <<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
          a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, g.slab_off,
          (float)a.GW / a.W, (float)a.GH / a.H, gn, up);
  return hipGetLastError();
}

}  // namespace

// Fused point-wise-NN guide + slice-apply.  Same shape support as the vec4 row kernel.
bool apply_fwd_nnguide_supported(const ApplyArgs& a, const float* guide_out) {
  if (!((a.Cin == 3 && a.Cout == 3) || (a.Cin == 1 && a.Cout == 1))) return false;
  ApplyArgs t = a;
  t.guide = a.input;  // alignment check stand-in: no guide buffer is read
  if ((uintptr_t)guide_out & 15u) return false;
  if (!apply_fwd_rows_supported(t)) return false;
  return make_row_plan(t.W, t.GW, (((uintptr_t)t.input | (uintptr_t)t.out | (uintptr_t)t.grid) & 15u) == 0).vec4;
}

hipError_t launch_apply_fwd_nnguide(const ApplyArgs& a, const float* conv1, const float* conv2,
                                    int n_feats, float* guide_out, hipStream_t s,
                                    const char** name) {
  if (apply_fwd_seg_nnguide_supported(a, guide_out)) {
    const hipError_t e = launch_apply_fwd_seg_nnguide(a, conv1, conv2, n_feats, guide_out, s, name);
    if (e != hipErrorNotSupported) return e;  // (a network too wide for its LDS copy falls through)
  }
  const GuideNN gn{conv1, conv2, guide_out, n_feats, a.fast_sigmoid, a.guide_prescaled};
  ApplyArgs t = a;
  t.guide = a.input;
  *name = "apply_fwd_rows/vec4+nnguide";
  if (a.Cin == 3 && a.Cout == 3 && a.has_offset) return launch_nnguide_t<3, 3, true>(t, gn, s);
  if (a.Cin == 3 && a.Cout == 3 && !a.has_offset) return launch_nnguide_t<3, 3, false>(t, gn, s);
  if (a.Cin == 1 && a.Cout == 1 && a.has_offset) return launch_nnguide_t<1, 1, true>(t, gn, s);
  if (a.Cin == 1 && a.Cout == 1 && !a.has_offset) return launch_nnguide_t<1, 1, false>(t, gn, s);
  return hipErrorInvalidValue;
}

// Slice-apply (+ optional fused guide network) + bilinear up-add of the coarser pyramid level.
// One specialisation: Cin = Cout = 3 with offset (the reference's pyramid model), vec4 geometry.
bool apply_fwd_upadd_supported(const ApplyArgs& a, const float* coarse, bool guide_nn) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset) || ((uintptr_t)coarse & 3u)) return false;
  ApplyArgs t = a;
  if (guide_nn) t.guide = a.input;
  if (!apply_fwd_rows_supported(t)) return false;
  const bool aligned = (((uintptr_t)t.guide | (uintptr_t)t.input | (uintptr_t)t.out | (uintptr_t)t.grid) & 15u) == 0;
  return make_row_plan(t.W, t.GW, aligned).vec4;
}

hipError_t launch_apply_fwd_upadd(const ApplyArgs& a, const float* coarse, int Hc, int Wc,
                                  const float* conv1, const float* conv2, int n_feats,
                                  hipStream_t s, const char** name) {
  if (apply_fwd_seg_upadd_supported(a, coarse, conv1 != nullptr)) {
    const hipError_t e = launch_apply_fwd_seg_upadd(a, coarse, Hc, Wc, conv1, conv2, n_feats, s, name);
    if (e != hipErrorNotSupported) return e;
  }
  if (conv1) {
    const GuideNN gn{conv1, conv2, nullptr, n_feats, a.fast_sigmoid, a.guide_prescaled};
    ApplyArgs t = a;
    t.guide = a.input;  // alignment stand-in: no guide buffer is read
    *name = "apply_fwd_rows/vec4+nnguide+upadd";
    return launch_upadd_t<true>(t, gn, coarse, Hc, Wc, s);
  }
  *name = "apply_fwd_rows/vec4+upadd";
  return launch_upadd_t<false>(a, GuideNN{nullptr, nullptr, nullptr, 0}, coarse, Hc, Wc, s);
}

bool apply_fwd_rows_supported(const ApplyArgs& a) {
  const bool shape = apply_fast_shape(a.Cin, a.Cout, a.has_offset);
  if (!shape) return false;
  // stage_row reads the grid as float4 when C % 4 == 0.
  if ((a.Cout * a.Cj) % 4 == 0 && ((uintptr_t)a.grid & 15u)) return false;
  if ((long long)a.B * a.H * ((a.W + 511) / 512) > 0x7fffffffLL) return false;
  const Plan pl = make_row_plan(a.W, a.GW, true);
  const size_t lds = ((size_t)pl.max_cols * a.GD * a.Cout * a.Cj + 4 +
                      (size_t)(pl.threads / 64) * 64 * kPxPerThread * (a.Cin > a.Cout ? a.Cin : a.Cout)) *
                     sizeof(float);
  return lds <= kMaxLdsBytes;
}

hipError_t launch_apply_fwd_rows(const ApplyArgs& a, hipStream_t s, const char** name) {
#ifdef HDRNET_TOOLS_BUILD
  if (a.variant != 0 && a.variant != 19) {  // benchmark-only kernels (tools build), never selected by flags == 0
    const hipError_t e = launch_apply_fwd_variant(a, s, name);
    if (e != hipErrorNotSupported) return e;
  }
  const bool round1_kernel = a.variant == 19;  // A/B: the round-1 product kernel below
#else
  const bool round1_kernel = false;
#endif
  // The product kernel for vec4-able inputs lives in apply_fwd_seg.hip; what remains here is the
  // scalar kernel (any W / alignment) and the fused guide-network / pyramid instantiations.
  if (!round1_kernel && apply_fwd_seg_supported(a)) return launch_apply_fwd_seg(a, s, name);
#define HDRNET_CASE(CI, CO, OFF) \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) return launch_t<CI, CO, OFF>(a, s, name);
  HDRNET_APPLY_FAST_SHAPES(HDRNET_CASE)
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

#ifdef HDRNET_TOOLS_BUILD
// The direct-store / plain-load instantiations, for tools/ab_bench.py only (apply_fwd_variants.hip
// routes here): which = 0 per-lane stores, 1 = nontemporal lane-contiguous input loads.
hipError_t launch_apply_fwd_rows_direct_stores(const ApplyArgs& a, hipStream_t s, const char** name, int which) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return hipErrorInvalidValue;
  const LaunchGeom g = geom_for<12, 3>(a);
  if (!g.pl.vec4) return hipErrorInvalidValue;
  if (which == 0)
    apply_fwd_rows_vec4<3, 3, true, false, false><<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
        a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, g.slab_off,
        (float)a.GW / a.W, (float)a.GH / a.H);
  else
    apply_fwd_rows_vec4<3, 3, true, false, true, false, true><<<(unsigned)g.nblocks, g.pl.threads, g.lds, s>>>(
        a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, g.pl.nseg, g.pl.seg, g.slab_off,
        (float)a.GW / a.W, (float)a.GH / a.H);
  *name = which == 0 ? "apply_fwd_rows/vec4-direct-stores" : "apply_fwd_rows/vec4-nt-loads";
  return hipGetLastError();
}

#endif  // HDRNET_TOOLS_BUILD

}  // namespace hdrnet_amd
