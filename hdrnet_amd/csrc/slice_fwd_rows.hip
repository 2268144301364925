// BilateralSlice forward (un-fused: returns all C sliced coefficients per pixel) for gfx950.
//
// Reference semantics: hdrnet/ops/bilateral_slice.cc:25-70 (CUDA twin bilateral_slice.cu.cc:34-91:
// one thread per output channel, 8 scattered grid loads each).
//
// Traffic is 4 B in and 4*C B out per pixel (52 B/px at C = 12) -- a write stream -- so the
// kernel is built around the stores: same y-pre-lerped LDS image and row segments as the
// fused forward (rows_common.hip.h, apply_fwd_rows.hip), but a lane takes pixels
// x0 + lane + 64*k, so that for each k the wave's 64 pixels x C floats are one contiguous
// run of the output; the wave transposes them through a private LDS slab and writes them
// as lane-contiguous 16-B stores (each global_store_dwordx4 covers a dense 1 KiB).
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

template <int C>
__global__ __launch_bounds__(256) void slice_fwd_rows(
    const float* __restrict__ grid, const float* __restrict__ guide, float* __restrict__ out,
    int H, int W, int GH, int GW, int GD, int nseg, int seg, int slab_offset_floats,
    float scale_x, float scale_y) {
  // C % 4 == 0: a pixel's C floats are float4s.  C in {1, 2} (the reference's own micro-benchmark
  // slices 2 channels, hdrnet_ops_jax_tf2_test.py:56-65): scalar slab writes; the run of 64 pixels
  // is still whole float4s because 64 C % 4 == 0.
  static_assert(C % 4 == 0 || C == 1 || C == 2, "float4 slab rows");
  extern __shared__ __attribute__((aligned(16))) float colY[];
  const int bid = blockIdx.x;
  const int segi = bid % nseg;
  const int row = bid / nseg;  // = b * H + y
  const int y = row % H;
  const int b = row / H;
  const int xs = segi * seg;
  const int xe = min(xs + seg, W);
  const float* grid_b = grid + (size_t)b * GH * GW * GD * C;
  const size_t prow = (size_t)row * W;
  const int lane = threadIdx.x & 63;
  const int wave_x0 = xs + (int)(threadIdx.x & ~63u) * kPxPerThread;  // first pixel of this wave

  float gs[kPxPerThread];
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const int x = wave_x0 + lane + 64 * k;
    gs[k] = (x < xe) ? guide[prow + x] : 0.0f;
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);
  float4* slab = reinterpret_cast<float4*>(colY + slab_offset_floats) + (threadIdx.x >> 6) * (64 * C / 4);  // 64 C % 4 == 0

#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    const int xk0 = wave_x0 + 64 * k;  // wave-uniform
    if (xk0 >= xe) break;
    const int x = xk0 + lane;
    if (x < xe) {
      const SliceTerms t = slice_terms<C, false>(r, (float)x + 0.5f, gs[k]);
      CoefVec<C> coef;
      accum_vec<C, true>(coef, r.colY, t.a00, t.wx0 * t.wz0);
      accum_vec<C, false>(coef, r.colY, t.a01, t.wx0 * t.wz1);
      accum_vec<C, false>(coef, r.colY, t.a10, t.wx1 * t.wz0);
      accum_vec<C, false>(coef, r.colY, t.a11, t.wx1 * t.wz1);
      if constexpr (C % 4 == 0) {
#pragma unroll
        for (int q = 0; q < C / 4; ++q)
          slab[lane * (C / 4) + q] = make_float4(coef.v[2 * q][0], coef.v[2 * q][1], coef.v[2 * q + 1][0],
                                                 coef.v[2 * q + 1][1]);
      } else {
        float* sf = reinterpret_cast<float*>(slab);
#pragma unroll
        for (int c = 0; c < C; ++c) sf[lane * C + c] = coef.get(c);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // nontemporal buffer stores on a descriptor over exactly this run (rows_common.hip.h)
    // (the descriptor ends at the run's last float: the bounds check is per dword, so a final
    //  partial float4 -- possible for C < 4 -- is written up to the run's end and no further)
    const unsigned run_bytes = (unsigned)(min(xe, xk0 + 64) - xk0) * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc_uniform(out + (prow + xk0) * C, run_bytes);
    constexpr int NQ = (64 * C / 4 + 63) / 64;  // store instructions per run: C / 4, or 1 for C < 4
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (C % 4 == 0 || lane + 64 * q < 64 * C / 4)
        buf_store16<kAuxStream>(slab[lane + 64 * q], orsrc, (unsigned)(lane + 64 * q) * 16u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

constexpr size_t kMaxLdsBytes = 64 * 1024;

bool plan_for(const SliceArgs& a, Plan* pl, size_t* lds, int* slab_off) {
  const bool aligned = (((uintptr_t)a.grid | (uintptr_t)a.out) & 15u) == 0;
  if (!aligned) return false;
  *pl = make_row_plan(a.W, a.GW, true);
  pl->vec4 = true;  // guide is read per pixel; only grid / out need 16-B alignment
  if ((long long)a.B * a.H * pl->nseg > 0x7fffffffLL) return false;
  *slab_off = round_up(pl->max_cols * a.GD * a.C, 4);
  *lds = ((size_t)*slab_off + (size_t)(pl->threads / 64) * 64 * a.C) * sizeof(float);
  return *lds <= kMaxLdsBytes;
}

template <int C>
hipError_t launch_c(const SliceArgs& a, const Plan& pl, size_t lds, int slab_off, hipStream_t s) {
  const long long nblocks = (long long)a.B * a.H * pl.nseg;
  slice_fwd_rows<C><<<(unsigned)nblocks, pl.threads, lds, s>>>(
      a.grid, a.guide, a.out, a.H, a.W, a.GH, a.GW, a.GD, pl.nseg, pl.seg, slab_off,
      (float)a.GW / a.W, (float)a.GH / a.H);
  return hipGetLastError();
}

}  // namespace

bool slice_fwd_rows_supported(const SliceArgs& a) {
  if (!(a.C == 1 || a.C == 2 || a.C == 4 || a.C == 8 || a.C == 12 || a.C == 16)) return false;
  Plan pl;
  size_t lds;
  int so;
  return plan_for(a, &pl, &lds, &so);
}

hipError_t launch_slice_fwd_rows(const SliceArgs& a, hipStream_t s, const char** name) {
  Plan pl;
  size_t lds;
  int so;
  if (!plan_for(a, &pl, &lds, &so)) return hipErrorInvalidValue;
  *name = "slice_fwd_rows";
  switch (a.C) {
    case 1: return launch_c<1>(a, pl, lds, so, s);
    case 2: return launch_c<2>(a, pl, lds, so, s);
    case 4: return launch_c<4>(a, pl, lds, so, s);
    case 8: return launch_c<8>(a, pl, lds, so, s);
    case 12: return launch_c<12>(a, pl, lds, so, s);
    case 16: return launch_c<16>(a, pl, lds, so, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace hdrnet_amd
