// Grid VJP (dgrid) of BilateralSliceApply / BilateralSlice for gfx950.
//
// Reference semantics: BilateralSliceApplyGridGrad, hdrnet/ops/bilateral_slice_apply.cc:84-138,
// and BilateralSliceGridGrad, hdrnet/ops/bilateral_slice.cc:72-118 (CPU code; the CUDA twin
// bilateral_slice_apply.cu.cc:128-206 mis-decodes the channel index).  The
// reference gathers: one thread per grid element loops over its +-1-cell pixel window with
// mirror boundaries -- 24 576 threads x 32 400 pixels at 1080p.  The mirror-gather is the
// exact transpose of the forward's clamp-to-edge scatter, and that is how it is computed here:
//
//   dgrid[gy, gx, gz, c] = sum over pixels  wy(gy; y) * wx(gx; x) * wz'(gz; guide) * V[pixel, c]
//   V[pixel, (i, j)] = dout_i * (j < Cin ? in_j : 1)          (slice: V[pixel, c] = dout_c)
//   wz' = the smoothed tent, forced to 1 in the outermost half cells (:121-125)
//
// For a run of pixels of one image row that share gx0 (the lower x corner) this is a dense
// contraction over the pixels:  D[k, c] += sum_px A[k, px] * V[px, c]  with 16 rows
// k = (xcorner, gz) and A = wx * wz'.  That is exactly one v_mfma_f32_16x16x4_f32 per 4
// pixels (f32 in, f32 accumulate, bit-equal to an fmaf chain), so the contraction runs on
// the matrix pipe beside the VALU that builds A.  It is the one place in this library where
// MFMA is the right tool: a real reduction over K = pixels, not a reshaped gather.
//
// Stage 1 (this file, grid_grad_stage1): a workgroup owns RG consecutive rows x one row
//   segment; each wave walks whole rows in chunks of <= 64 pixels cut at gx0 changes,
//   stages the chunk's V rows and guide values in LDS, issues the next chunk's loads, runs
//   the 16 MFMA steps, and at every gx0 change folds the 16x16 accumulator -- scaled by the
//   row's two y weights -- into its private LDS accumulator (3 grid rows x the segment's
//   columns).  The four wave accumulators are summed in fixed order and written as one
//   partial tile.  No atomics anywhere: the result is deterministic.
// Stage 2 (grid_grad_stage2): one thread per dgrid element adds, in fixed order, the few
//   partial tiles whose window covers it.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using rows::max_cols_for;
using rows::round_up;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;
constexpr int kVStride = 20;  // floats per V row in LDS (16 + pad: conflict-free b128 writes)

struct GGParams {
  const float* guide;
  const float* input;  // null for slice
  const float* dout;
  float* partial;  // [B][nyg][nseg][3][ncol][GD][C]
  int H, W, GH, GW, GD;
  int rg, nyg, nseg, seg, ncol;
  float scale_x, scale_y;  // GW / W, GH / H  (forward's expressions)
};

__device__ __forceinline__ int gx0_of(int x, float scale_x) {
  return floor_to_int(__fmul_rn(x + 0.5f, scale_x) - 0.5f);
}

// CIN/COUT/OFFSET as in the forward; APPLY = false: V = dout (C = COUT channels).
template <int CIN, int COUT, bool OFFSET, bool APPLY>
__global__ __launch_bounds__(kWaves * 64) void grid_grad_stage1(GGParams p) {
  constexpr int CJ = APPLY ? CIN + (OFFSET ? 1 : 0) : 1;
  constexpr int C = COUT * CJ;
  static_assert(C <= 16, "one 16-column MFMA tile");
  constexpr int CIN_Q = (APPLY && CIN > 0) ? CIN : 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int acc_floats = 3 * p.ncol * p.GD * C;
  float* acc = lds + wave * acc_floats;                                // [3][ncol][GD][C]
  float* vt = lds + kWaves * acc_floats + wave * (64 * kVStride + 64);  // V rows, then gzf[64]
  float* gz = vt + 64 * kVStride;

  const int segi = blockIdx.x % p.nseg;
  const int yg = (blockIdx.x / p.nseg) % p.nyg;
  const int b = blockIdx.x / (p.nseg * p.nyg);
  const int xs = segi * p.seg, xe = min(xs + p.seg, p.W);
  const int y_first = yg * p.rg, y_end = min(y_first + p.rg, p.H);
  const int gxlo = clamp_index(gx0_of(xs, p.scale_x), 0, p.GW - 1);
  const int gy_base =
      clamp_index(floor_to_int(__fmul_rn(y_first + 0.5f, p.scale_y) - 0.5f), 0, p.GH - 1);

  for (int e = lane; e < acc_floats; e += 64) acc[e] = 0.0f;

  // Per-lane roles in the MFMA (v_mfma_f32_16x16x4_f32):
  //   A[k = lane & 15][kk = lane >> 4], B[kk = lane >> 4][c = lane & 15],
  //   D[k = 4 * (lane >> 4) + r][c = lane & 15] in register r.
  const int ak = lane & 15, sub = lane >> 4;
  const int a_xcol = ak >> 3, a_z = ak & 7;
  const bool a_zvalid = a_z < p.GD;
  const float a_zc = a_z + 0.5f;
  const bool a_lo = a_z == 0, a_hi = a_z == p.GD - 1;
  const float gd_f = (float)p.GD;
  const int bc = lane & 15;  // V column read by this lane
  const int d_xcol = sub >> 1;
  const int d_z0 = (sub & 1) * 4;

  for (int y = y_first + wave; y < y_end; y += kWaves) {
    // y terms of this row (bilateral_slice_apply.cc:42,47,55-56; weights un-clamped).
    const float gyf = __fmul_rn(y + 0.5f, p.scale_y);
    const int gy0 = floor_to_int(gyf - 0.5f);
    const float wy0 = tent_weight(gy0 + 0.5f, gyf);
    const float wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
    const int rel0 = clamp_index(gy0, 0, p.GH - 1) - gy_base;
    const int rel1 = clamp_index(gy0 + 1, 0, p.GH - 1) - gy_base;
    const size_t prow = ((size_t)b * p.H + y) * p.W;

    int pos = xs;
    // first chunk's loads
    float g_cur = 0.f, in_cur[CIN_Q], d_cur[COUT];
    {
      const int x = pos + lane;
      if (x < xe) {
        g_cur = p.guide[prow + x];
        if constexpr (APPLY && CIN > 0) {
#pragma unroll
          for (int j = 0; j < CIN; ++j) in_cur[j] = p.input[(prow + x) * CIN + j];
        }
#pragma unroll
        for (int i = 0; i < COUT; ++i) d_cur[i] = p.dout[(prow + x) * COUT + i];
      }
    }
    f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
    int g_run = gx0_of(pos, p.scale_x);

    while (pos < xe) {
      // chunk = leading lanes that are inside the segment and share gx0 with lane 0
      const int x = pos + lane;
      const bool same = (x < xe) && (gx0_of(x, p.scale_x) == g_run);
      const unsigned long long m = __ballot(same);
      const int len = (~m == 0ull) ? 64 : (int)__builtin_ctzll(~m);

      // stage V rows and gzf of the chunk (lane = pixel)
      {
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0.0f;
        if (lane < len) {
          if constexpr (APPLY) {
#pragma unroll
            for (int i = 0; i < COUT; ++i) {
#pragma unroll
              for (int j = 0; j < CJ; ++j) v[i * CJ + j] = (j < CIN) ? d_cur[i] * in_cur[j < CIN ? j : 0] : d_cur[i];
            }
          } else {
#pragma unroll
            for (int c = 0; c < C; ++c) v[c] = d_cur[c];
          }
        }
        f32x4* vrow = reinterpret_cast<f32x4*>(vt + lane * kVStride);
#pragma unroll
        for (int q = 0; q < 4; ++q) vrow[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        gz[lane] = __fmul_rn(g_cur, gd_f);  // gzf = guide * GD  (:120)
      }
      // next chunk's loads go out before the MFMA loop
      const int npos = pos + len;
      float g_nxt = 0.f, in_nxt[CIN_Q], d_nxt[COUT];
      {
        const int xn = npos + lane;
        if (xn < xe) {
          g_nxt = p.guide[prow + xn];
          if constexpr (APPLY && CIN > 0) {
#pragma unroll
            for (int j = 0; j < CIN; ++j) in_nxt[j] = p.input[(prow + xn) * CIN + j];
          }
#pragma unroll
          for (int i = 0; i < COUT; ++i) d_nxt[i] = p.dout[(prow + xn) * COUT + i];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // x-corner selection of this lane's A row; columns that clamp onto each other are
      // folded into one row so that the fold below never adds two rows to one address.
      float sel0, sel1;
      if (g_run < 0) {  // corner 0 is column -1 -> column 0 == corner 1
        sel0 = a_xcol ? 1.0f : 0.0f;
        sel1 = sel0;
      } else if (g_run >= p.GW - 1) {  // corner 1 is column GW -> column GW-1 == corner 0
        sel0 = a_xcol ? 0.0f : 1.0f;
        sel1 = sel0;
      } else {
        sel0 = a_xcol ? 0.0f : 1.0f;
        sel1 = a_xcol ? 1.0f : 0.0f;
      }
      const float gc0 = g_run + 0.5f, gc1 = g_run + 1 + 0.5f;

      const int nsteps = (len + 3) >> 2;
      for (int t = 0; t < nsteps; ++t) {
        const int px = 4 * t + sub;
        const float gxf = __fmul_rn((float)(pos + px) + 0.5f, p.scale_x);
        const float wx = sel0 * tent_weight(gc0, gxf) + sel1 * tent_weight(gc1, gxf);
        const float gzf = gz[px];
        float wz = smoothed_tent_weight(a_zc, gzf);
        if ((a_lo && gzf < 0.5f) || (a_hi && gzf > gd_f - 0.5f)) wz = 1.0f;  // :121-125
        const float a = (a_zvalid && px < len) ? wx * wz : 0.0f;
        const float bv = vt[px * kVStride + bc];
        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, dacc, 0, 0, 0);
      }

      // advance; fold the accumulator when the run of equal gx0 ends (or the row does)
      pos = npos;
      g_cur = g_nxt;
#pragma unroll
      for (int j = 0; j < CIN_Q; ++j) in_cur[j] = in_nxt[j];
#pragma unroll
      for (int i = 0; i < COUT; ++i) d_cur[i] = d_nxt[i];
      const int g_next = (pos < xe) ? gx0_of(pos, p.scale_x) : g_run + 1000000;
      if (g_next != g_run) {
        const bool fold_lo = g_run < 0, fold_hi = g_run >= p.GW - 1;
        const bool row_live = fold_lo ? (d_xcol == 1) : (fold_hi ? (d_xcol == 0) : true);
        const int col = clamp_index(g_run + d_xcol, 0, p.GW - 1) - gxlo;
        if (row_live && bc < C && col >= 0 && col < p.ncol) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int z = d_z0 + r;
            if (z < p.GD) {
              const int o = ((col * p.GD) + z) * C + bc;
              const int plane = p.ncol * p.GD * C;
              acc[rel0 * plane + o] += wy0 * dacc[r];
              acc[rel1 * plane + o] += wy1 * dacc[r];
            }
          }
        }
        dacc = f32x4{0.f, 0.f, 0.f, 0.f};
        g_run = g_next;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

  // fixed-order sum of the wave accumulators -> partial tile
  __syncthreads();
  float* dst = p.partial + (size_t)blockIdx.x * acc_floats;
  for (int e = threadIdx.x; e < acc_floats; e += kWaves * 64) {
    float s = lds[e];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += lds[w * acc_floats + e];
    dst[e] = s;
  }
}

// One thread per dgrid element; adds the partial tiles that cover it, in (yg, seg) order.
__global__ __launch_bounds__(256) void grid_grad_stage2(const float* __restrict__ partial,
                                                        float* __restrict__ dgrid, long long nelem,
                                                        int H, int W, int GH, int GW, int GD, int C,
                                                        int rg, int nyg, int nseg, int seg, int ncol,
                                                        float scale_x, float scale_y) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nelem) return;
  const int c = (int)(e % C);
  const int z = (int)((e / C) % GD);
  const int gx = (int)((e / ((long long)C * GD)) % GW);
  const int gy = (int)((e / ((long long)C * GD * GW)) % GH);
  const long long b = e / ((long long)C * GD * GW * GH);
  const int plane = ncol * GD * C;
  float s = 0.0f;
  // Conservative window of row groups / segments that can touch (gy, gx); the exact
  // membership test is the `rel` / `col` check below.
  const int yg_lo = max(0, (int)floorf((gy - 2.0f) / scale_y) / rg - 1);
  const int yg_hi = min(nyg, (int)ceilf((gy + 2.5f) / scale_y) / rg + 2);
  const int sg_lo = max(0, (int)floorf((gx - 2.0f) / scale_x) / seg - 1);
  const int sg_hi = min(nseg, (int)ceilf((gx + 2.5f) / scale_x) / seg + 2);
  for (int yg = max(yg_lo, 0); yg < yg_hi; ++yg) {
    const int gy_base =
        clamp_index(floor_to_int(__fmul_rn(yg * rg + 0.5f, scale_y) - 0.5f), 0, GH - 1);
    const int rel = gy - gy_base;
    if (rel < 0 || rel > 2) continue;
    for (int sg = sg_lo; sg < sg_hi; ++sg) {
      const int gxlo = clamp_index(gx0_of(sg * seg, scale_x), 0, GW - 1);
      const int col = gx - gxlo;
      if (col < 0 || col >= ncol) continue;
      const size_t tile = ((size_t)b * nyg + yg) * nseg + sg;
      s += partial[tile * (3 * (size_t)plane) + (size_t)rel * plane + ((size_t)col * GD + z) * C + c];
    }
  }
  dgrid[e] = s;
}

struct GGPlan {
  int rg, nyg, nseg, seg, ncol;
  size_t acc_floats, lds_bytes, ws_bytes;
};

bool gg_plan(int B, int H, int W, int GH, int GW, int GD, int C, GGPlan* pl) {
  if (GD > 8 || C > 16 || C < 1) return false;
  // Row segments of <= 512 pixels and <= ~3 grid cells, balanced over the row: at most 6
  // grid columns per workgroup keeps the four private wave accumulators small (LDS, not
  // registers, decides the occupancy of this kernel).
  long long seg_max = 3LL * W / GW;
  if (seg_max > 512) seg_max = 512;
  if (seg_max < 1) seg_max = 1;
  pl->nseg = (int)((W + seg_max - 1) / seg_max);
  pl->seg = (W + pl->nseg - 1) / pl->nseg;
  pl->ncol = max_cols_for(pl->seg, GW, W);
  // rows of a group may span at most 3 (clamped) grid rows: rg <= cell height
  int rg = H / GH;
  if (rg > 8) rg = 8;
  if (rg < 1) rg = 1;
  pl->rg = rg;
  pl->nyg = (H + rg - 1) / rg;
  pl->acc_floats = (size_t)3 * pl->ncol * GD * C;
  pl->lds_bytes = (kWaves * pl->acc_floats + (size_t)kWaves * (64 * kVStride + 64)) * sizeof(float);
  if (pl->lds_bytes > 64 * 1024) return false;
  const long long nblocks = (long long)B * pl->nyg * pl->nseg;
  if (nblocks > 0x7fffffffLL) return false;
  pl->ws_bytes = (size_t)nblocks * pl->acc_floats * sizeof(float);
  return true;
}

template <int CIN, int COUT, bool OFFSET, bool APPLY>
hipError_t gg_launch(const float* guide, const float* input, const float* dout, float* dgrid, int B,
                     int H, int W, int GH, int GW, int GD, void* ws, const GGPlan& pl, hipStream_t s) {
  constexpr int C = APPLY ? COUT * (CIN + (OFFSET ? 1 : 0)) : COUT;
  GGParams p{guide, input, dout, static_cast<float*>(ws), H, W, GH, GW, GD,
             pl.rg, pl.nyg, pl.nseg, pl.seg, pl.ncol, (float)GW / W, (float)GH / H};
  const long long nblocks = (long long)B * pl.nyg * pl.nseg;
  grid_grad_stage1<CIN, COUT, OFFSET, APPLY><<<(unsigned)nblocks, kWaves * 64, pl.lds_bytes, s>>>(p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long long nelem = (long long)B * GH * GW * GD * C;
  grid_grad_stage2<<<(unsigned)((nelem + 255) / 256), 256, 0, s>>>(
      static_cast<const float*>(ws), dgrid, nelem, H, W, GH, GW, GD, C, pl.rg, pl.nyg, pl.nseg,
      pl.seg, pl.ncol, (float)GW / W, (float)GH / H);
  return hipGetLastError();
}

bool apply_shape_ok(int Cin, int Cout, bool off) {
  return (Cin == 3 && Cout == 3) || (Cin == 3 && Cout == 4 && off) || (Cin == 1 && Cout == 1) ||
         (Cin == 1 && Cout == 3 && off) || (Cin == 4 && Cout == 4 && !off);
}

bool slice_c_ok(int C) { return C == 1 || C == 2 || C == 4 || C == 8 || C == 12 || C == 16; }

}  // namespace

size_t apply_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                                      bool has_offset) {
  GGPlan pl;
  if (!apply_shape_ok(Cin, Cout, has_offset)) return 0;
  if (!gg_plan(B, H, W, GH, GW, GD, Cout * (Cin + (has_offset ? 1 : 0)), &pl)) return 0;
  return pl.ws_bytes;
}

bool apply_grid_grad_mfma_supported(const ApplyGradArgs& a) {
  GGPlan pl;
  return apply_shape_ok(a.Cin, a.Cout, a.has_offset) &&
         gg_plan(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.Cout * a.Cj, &pl) && a.workspace != nullptr &&
         a.workspace_bytes >= pl.ws_bytes;
}

hipError_t launch_apply_grid_grad_mfma(const ApplyGradArgs& a, hipStream_t s, const char** name) {
  GGPlan pl;
  if (!gg_plan(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.Cout * a.Cj, &pl)) return hipErrorInvalidValue;
  *name = "grid_grad_mfma";
#define HDRNET_CASE(CI, CO, OFF)                                                                  \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF)                                         \
  return gg_launch<CI, CO, OFF, true>(a.guide, a.input, a.dout, a.dgrid, a.B, a.H, a.W, a.GH, a.GW, \
                                      a.GD, a.workspace, pl, s)
  HDRNET_CASE(3, 3, true);
  HDRNET_CASE(3, 3, false);
  HDRNET_CASE(3, 4, true);
  HDRNET_CASE(1, 1, true);
  HDRNET_CASE(1, 1, false);
  HDRNET_CASE(1, 3, true);
  HDRNET_CASE(4, 4, false);
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

size_t slice_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int C) {
  GGPlan pl;
  if (!slice_c_ok(C) || !gg_plan(B, H, W, GH, GW, GD, C, &pl)) return 0;
  return pl.ws_bytes;
}

bool slice_grid_grad_mfma_supported(const SliceGradArgs& a) {
  GGPlan pl;
  return slice_c_ok(a.C) && gg_plan(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.C, &pl) &&
         a.workspace != nullptr && a.workspace_bytes >= pl.ws_bytes;
}

hipError_t launch_slice_grid_grad_mfma(const SliceGradArgs& a, hipStream_t s, const char** name) {
  GGPlan pl;
  if (!gg_plan(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.C, &pl)) return hipErrorInvalidValue;
  *name = "grid_grad_mfma";
#define HDRNET_CASE(CC)                                                                            \
  if (a.C == CC)                                                                                   \
  return gg_launch<0, CC, false, false>(a.guide, nullptr, a.dout, a.dgrid, a.B, a.H, a.W, a.GH, a.GW, \
                                        a.GD, a.workspace, pl, s)
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
