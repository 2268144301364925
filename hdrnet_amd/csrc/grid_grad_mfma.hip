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
// For the pixels of one image row that share gx0 (the lower x corner) this is a dense
// contraction over the pixels:  D[k, c] += sum_px A[k, px] * V[px, c]  with 16 rows
// k = (xcorner, gz) and A = wx * wz'.  That is exactly one v_mfma_f32_16x16x4_f32 per 4
// pixels (f32 in, f32 accumulate, bit-equal to an fmaf chain), so the contraction runs on
// the matrix pipe.  It is the one place in this library where MFMA is the right tool: a real
// reduction over K = pixels, not a reshaped gather.
//
// Stage 1 (grid_grad_stage1): ONE WAVE owns one x-interval (all pixels with gx0 == g,
//   g = -1 .. GW-1) of RG consecutive rows.  Per row it loads the interval's pixels
//   (guide, input, dout: up to 4 chunks of 64 in flight), and per chunk each lane builds
//   ITS pixel's two operand rows -- V (dout x [in; 1]) and A (16 weights, the two x corners
//   folded into one row where they clamp onto the same column) -- writes them to a private
//   LDS slab and the wave reads them back transposed into 16 MFMAs.  The row's 16x16 result
//   is scaled by its two y weights into three REGISTER tiles (the <= 3 grid rows the group
//   touches); after the last row the tiles go to the workspace.  No LDS accumulator, no
//   atomics, no barrier: waves are independent and the result is deterministic.
// Stage 2 (grid_grad_stage2): one thread per dgrid element adds, in fixed order, the
//   partial tiles of the row groups and the two intervals that cover it.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;      // waves per workgroup; they share ONE task and split its rows
// Operand rows in LDS: 16 floats per pixel, the row's four float4 slots XOR-swizzled by
// (pixel >> 1) & 3.  Writes (ds_write_b128, 8-lane groups, one row per lane) then cover 8
// distinct 4-bank groups; the transposed ds_read_b32 (32-lane halves = pixels 4t, 4t+1, 16
// columns each) cover all 32 banks once.  (A 20-float padded row measured 33 % conflict cycles.)
constexpr int kVStride = 16;
constexpr int kBatch = 4;      // chunks of 64 pixels loaded ahead per row
constexpr int kTileFloats = 3 * 16 * 16;  // partial tile: [rel 3][k 16][c 16]

struct GGParams {
  const float* guide;
  const float* input;  // null for slice
  const float* dout;
  float* partial;  // [B][nyg][GW + 1][3][16][16]
  int H, W, GH, GW, GD;
  int rg, nyg;
  long long ntasks;
  float scale_x, scale_y;  // GW / W, GH / H  (forward's expressions)
};

// LDS traffic of ONE wave needs no fence: the LDS executes a wave's instructions in order, so
// a ds_read issued after a ds_write of the same wave sees all 64 lanes' data.  (A
// `fence(release, "wavefront")` costs an s_waitcnt vmcnt(0), i.e. would drain the prefetch.)
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ int gx0_of(int x, float scale_x) {
  return floor_to_int(mul_rn(x + 0.5f, scale_x) - 0.5f);
}

// Smallest x in [0, W] with gx0_of(x) >= g (gx0_of is non-decreasing in x).
__device__ __forceinline__ int interval_start(int g, int W, float scale_x) {
  if (g <= -1) return 0;
  int x = (int)ceilf((g + 0.5f) / scale_x - 0.5f);
  x = min(max(x, 0), W);
  while (x > 0 && gx0_of(x - 1, scale_x) >= g) --x;
  while (x < W && gx0_of(x, scale_x) < g) ++x;
  return x;
}

__device__ __forceinline__ int gy_base_of(int y_first, float scale_y, int GH) {
  return clamp_index(floor_to_int(mul_rn(y_first + 0.5f, scale_y) - 0.5f), 0, GH - 1);
}

// CIN/COUT/OFFSET as in the forward; APPLY = false: V = dout (C = COUT channels).
template <int CIN, int COUT, bool OFFSET, bool APPLY>
__global__ __launch_bounds__(kWaves * 64) void grid_grad_stage1(GGParams p) {
  constexpr int CJ = APPLY ? CIN + (OFFSET ? 1 : 0) : 1;
  constexpr int C = COUT * CJ;
  static_assert(C <= 16, "one 16-column MFMA tile");
  constexpr int CIN_Q = (APPLY && CIN > 0) ? CIN : 1;
  __shared__ __attribute__((aligned(16))) float lds[kWaves * 2 * 64 * kVStride];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float* vt = lds + wave * (2 * 64 * kVStride);  // V rows [64][20]
  float* at = vt + 64 * kVStride;                // A rows [64][20]

  const long long task = blockIdx.x;  // one (image, row group, x-interval) per workgroup
  {  // V rows: zero once (the channel pad of each row is never written again)
    f32x4* vz = reinterpret_cast<f32x4*>(vt);
#pragma unroll
    for (int q = 0; q < 4; ++q) vz[lane * 4 + q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nint = p.GW + 1;
  const int g = (int)(task % nint) - 1;  // gx0 of this wave's pixels
  const int yg = (int)((task / nint) % p.nyg);
  const long long b = task / ((long long)nint * p.nyg);
  const int x_lo = interval_start(g, p.W, p.scale_x);
  const int x_hi = interval_start(g + 1, p.W, p.scale_x);
  const int y_first = yg * p.rg, y_end = min(y_first + p.rg, p.H);
  const int gy_base = gy_base_of(y_first, p.scale_y, p.GH);
  const float gd_f = (float)p.GD;
  const bool fold_lo = g < 0, fold_hi = g >= p.GW - 1;
  const float gc0 = g + 0.5f, gc1 = g + 1 + 0.5f;

  // MFMA lane roles (v_mfma_f32_16x16x4_f32): A[k = lane & 15][kk = lane >> 4],
  // B[kk = lane >> 4][c = lane & 15], D[k = 4 * (lane >> 4) + r][c = lane & 15] in register r.
  const int sub = lane >> 4, bc = lane & 15;
  // element (row r = 4t + sub, column bc) lives at r*16 + 4*((bc >> 2) ^ ((r >> 1) & 3)) + (bc & 3)
  auto rd_ofs = [&](int t) { return (4 * t + sub) * kVStride + 4 * ((bc >> 2) ^ ((2 * t + (sub >> 1)) & 3)) + (bc & 3); };

  f32x4 acc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Pixel batches of this wave, flattened over (row, batch-in-row): rows y_first + wave,
  // + kWaves, ...; per row ceil(interval / 256) batches of <= 4 chunks of 64 pixels.  The loads
  // of batch t+1 are issued before batch t is processed (two register sets), so a row's HBM
  // latency hides behind the previous batch's ~6k cycles of staging + MFMA.
  struct Batch {
    float g[kBatch], in[kBatch][CIN_Q], d[kBatch][COUT];
  };
  const int span = x_hi - x_lo;
  const int nbr = (span + 64 * kBatch - 1) / (64 * kBatch);  // batches per row
  const int nrows = (y_end - (y_first + wave) + kWaves - 1) / kWaves;
  const int nbt = (span > 0 && nrows > 0) ? nrows * nbr : 0;
  auto load_batch = [&](int t, Batch& bt) {
    const int r = t / nbr, bi = t - r * nbr;
    const int y = y_first + wave + r * kWaves;
    const size_t prow = ((size_t)b * p.H + y) * p.W;
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      // unconditional (clamped) loads: no exec-masked branch around VMEM keeps the compiler's
      // vmcnt counts exact; pixels past the interval are zero-weighted below.
      const size_t px = prow + min(xb + 64 * cb + lane, x_hi - 1);
      bt.g[cb] = p.guide[px];
      if constexpr (APPLY && CIN > 0) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) bt.in[cb][j] = p.input[px * CIN + j];
      }
#pragma unroll
      for (int i = 0; i < COUT; ++i) bt.d[cb][i] = p.dout[px * COUT + i];
    }
  };

  Batch cur, nxt;
  if (nbt > 0) load_batch(0, cur);
  f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacc2 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < nbt; ++t) {
    if (t + 1 < nbt) load_batch(t + 1, nxt);
    const int r = t / nbr, bi = t - r * nbr;
    const int y = y_first + wave + r * kWaves;
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      const int x0 = xb + 64 * cb;
      if (x0 < x_hi) {  // wave-uniform
        const int len = min(64, x_hi - x0);
        const int x = x0 + lane;
        // Stage both MFMA operands, lane = pixel, branch-free (idle lanes carry zero weights
        // and zero V).  Columns that clamp onto each other (g = -1: corner 0 -> column 0 ==
        // corner 1; g = GW-1: corner 1 -> column GW-1 == corner 0) are folded into ONE A row,
        // so stage 2 never sees a column twice.
        const float live = (lane < len) ? 1.0f : 0.0f;
        const float gxf = mul_rn((float)x + 0.5f, p.scale_x);
        const float wxa = tent_weight(gc0, gxf) * live;
        const float wxb = tent_weight(gc1, gxf) * live;
        const float w0 = fold_lo ? 0.0f : (fold_hi ? wxa + wxb : wxa);
        const float w1 = fold_lo ? wxa + wxb : (fold_hi ? 0.0f : wxb);
        // z: only the two corners around gzf carry weight (:121); the outermost half cells are
        // forced to 1 (:122-125).  Two v_sqrt_f32 per pixel (1 ulp; argument >= 1e-8, no
        // denormals; a weight moves by <= 6e-8, far below the summation noise of a 30 000-term
        // reduction), then one select chain per gz row.
        const float gzf = mul_rn(cur.g[cb], gd_f);  // gzf = guide * GD  (:120)
        const float fz = floorf(gzf - 0.5f);
        const float dza = (fz + 0.5f) - gzf, dzb = (fz + 1.5f) - gzf;
        const float wa = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dza, dza, kSmoothEps)), 0.0f);
        const float wb = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dzb, dzb, kSmoothEps)), 0.0f);
        const int za = (int)__builtin_amdgcn_fmed3f(fz, -2.0f, gd_f + 1.0f), zb = za + 1;
        const bool lo = gzf < 0.5f, hi = gzf > gd_f - 0.5f;
        f32x4 wzv[2];
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          float wz = (z == za) ? wa : ((z == zb) ? wb : 0.0f);
          if (z == 0) wz = lo ? 1.0f : wz;
          wz = (z == p.GD - 1 && hi) ? 1.0f : wz;
          wz = (z < p.GD) ? wz : 0.0f;
          wzv[z >> 2][z & 3] = wz;
        }
        float vflat[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) vflat[c] = 0.0f;
        if constexpr (APPLY) {
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
            const float di = cur.d[cb][i] * live;
#pragma unroll
            for (int j = 0; j < CJ; ++j) vflat[i * CJ + j] = (j < CIN) ? di * cur.in[cb][j < CIN ? j : 0] : di;
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) vflat[c] = cur.d[cb][c] * live;
        }
        f32x4* vrow = reinterpret_cast<f32x4*>(vt + lane * kVStride);
        f32x4* ar = reinterpret_cast<f32x4*>(at + lane * kVStride);
        const int wsw = (lane >> 1) & 3;
        // V: only the float4 slots that hold channels are written; the rest of the row was
        // zeroed once at kernel start and is never touched again.
#pragma unroll
        for (int q = 0; q < (C + 3) / 4; ++q)
          vrow[q ^ wsw] = f32x4{vflat[4 * q], vflat[4 * q + 1], vflat[4 * q + 2], vflat[4 * q + 3]};
        ar[0 ^ wsw] = w0 * wzv[0];
        ar[1 ^ wsw] = w0 * wzv[1];
        ar[2 ^ wsw] = w1 * wzv[0];
        ar[3 ^ wsw] = w1 * wzv[1];
        wave_lds_order();
        // D[k, c] += sum_px A[k, px] * V[px, c]; lane l reads A[px = 4t + (l >> 4)][l & 15]
        // and V likewise.  Two accumulators break the dependent-issue chain.
        if (len == 64) {  // full chunk: straight-line, all 32 LDS reads can run ahead
          float av[16], bv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            av[u] = at[rd_ofs(u)];
            bv[u] = vt[rd_ofs(u)];
          }
#pragma unroll
          for (int u = 0; u < 16; u += 2) {
            dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], dacc, 0, 0, 0);
            dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[u + 1], dacc2, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int tg = 0; tg < 4; ++tg) {
            if (16 * tg < len) {  // wave-uniform
              float av[4], bv[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                av[u] = at[rd_ofs(4 * tg + u)];
                bv[u] = vt[rd_ofs(4 * tg + u)];
              }
              dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], dacc, 0, 0, 0);
              dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], dacc2, 0, 0, 0);
              dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], dacc, 0, 0, 0);
              dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], dacc2, 0, 0, 0);
            }
          }
        }
        wave_lds_order();
      }
    }
    if (bi == nbr - 1) {
      // last batch of the row: fold the row's 16x16 result, scaled by its two y weights
      // (bilateral_slice_apply.cc:42,47,55-56; weights un-clamped, indices clamped), into the
      // register tiles of the (<= 3) grid rows the group touches.
      const float gyf = mul_rn(y + 0.5f, p.scale_y);
      const int gy0 = floor_to_int(gyf - 0.5f);
      const float wy0 = tent_weight(gy0 + 0.5f, gyf);
      const float wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
      const int rel0 = clamp_index(gy0, 0, p.GH - 1) - gy_base;
      const int rel1 = clamp_index(gy0 + 1, 0, p.GH - 1) - gy_base;
      dacc += dacc2;
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const float sr = (rel0 == rr ? wy0 : 0.0f) + (rel1 == rr ? wy1 : 0.0f);
        acc[rr] += sr * dacc;
      }
      dacc = f32x4{0.f, 0.f, 0.f, 0.f};
      dacc2 = dacc;
    }
    cur = nxt;
  }
  // Sum the four waves' register tiles in fixed order (wave 0 + 1 + 2 + 3) through LDS -- the
  // operand slabs are free now -- and write one partial tile per workgroup.
  __syncthreads();
  float* red = lds + wave * kTileFloats;  // kTileFloats = 768 <= 2 * 64 * kVStride
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red[(r * 16 + 4 * sub + q) * 16 + bc] = acc[r][q];
  }
  __syncthreads();
  float* dst = p.partial + (size_t)task * kTileFloats;
  for (int e = threadIdx.x; e < kTileFloats; e += kWaves * 64) {
    float sum = lds[e];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += lds[w * kTileFloats + e];
    dst[e] = sum;
  }
}

// Stage 2.  One 256-thread workgroup per grid cell (b, gy, gx, gz): lane (part, c) adds the
// partial tiles of row groups yg = yg_lo + part, +16, ... for channel c -- for each group the
// interval g = gx (its x-corner-0 row) and the interval g = gx - 1 (its x-corner-1 row) --
// then the 16 partial sums per channel are added in fixed order.  Reads are 64-B runs (the 16
// channels of one tile row); the result is deterministic.
__global__ __launch_bounds__(256) void grid_grad_stage2(const float* __restrict__ partial,
                                                        float* __restrict__ dgrid, int GH, int GW,
                                                        int GD, int C, int rg, int nyg,
                                                        float scale_y) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, part = threadIdx.x >> 4;
  const long long cell = blockIdx.x;  // ((b * GH + gy) * GW + gx) * GD + z
  const int z = (int)(cell % GD);
  const int gx = (int)((cell / GD) % GW);
  const int gy = (int)((cell / ((long long)GD * GW)) % GH);
  const long long b = cell / ((long long)GD * GW * GH);
  const int nint = GW + 1;
  // Conservative window of row groups that can touch gy; exact membership is `rel`.
  const int yg_lo = max(0, (int)floorf((gy - 2.0f) / scale_y) / rg - 1);
  const int yg_hi = min(nyg, (int)ceilf((gy + 2.5f) / scale_y) / rg + 2);
  float s = 0.0f;
  for (int yg = yg_lo + part; yg < yg_hi; yg += 16) {
    const int rel = gy - gy_base_of(yg * rg, scale_y, GH);
    if (rel < 0 || rel > 2) continue;
    const size_t t0 = ((size_t)b * nyg + yg) * nint;
    s += partial[(t0 + gx + 1) * kTileFloats + (rel * 16 + z) * 16 + c];
    s += partial[(t0 + gx) * kTileFloats + (rel * 16 + 8 + z) * 16 + c];
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0 && c < C) {
    float t = red[0][c];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][c];
    dgrid[cell * C + c] = t;
  }
}

struct GGPlan {
  int rg, nyg;
  long long ntasks;
  size_t ws_bytes;
};

bool gg_plan(int B, int H, int W, int GH, int GW, int GD, int C, GGPlan* pl) {
  if (GD > 8 || C > 16 || C < 1) return false;
  // rows of a group may span at most 3 (clamped) grid rows: rg <= cell height
  int rg = H / GH;
  if (rg > 8) rg = 8;
  if (rg < 1) rg = 1;
  pl->rg = rg;
  pl->nyg = (H + rg - 1) / rg;
  pl->ntasks = (long long)B * pl->nyg * (GW + 1);
  if (pl->ntasks > 0x7fffffffLL || (long long)B * GH * GW * GD > 0x7fffffffLL) return false;
  pl->ws_bytes = (size_t)pl->ntasks * kTileFloats * sizeof(float);
  return true;
}

template <int CIN, int COUT, bool OFFSET, bool APPLY>
hipError_t gg_launch(const float* guide, const float* input, const float* dout, float* dgrid, int B,
                     int H, int W, int GH, int GW, int GD, void* ws, const GGPlan& pl, hipStream_t s) {
  constexpr int C = APPLY ? COUT * (CIN + (OFFSET ? 1 : 0)) : COUT;
  GGParams p{guide, input, dout, static_cast<float*>(ws), H, W, GH, GW, GD,
             pl.rg, pl.nyg, pl.ntasks, (float)GW / W, (float)GH / H};
  const long long nblocks = pl.ntasks;
  grid_grad_stage1<CIN, COUT, OFFSET, APPLY><<<(unsigned)nblocks, kWaves * 64, 0, s>>>(p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long long ncell = (long long)B * GH * GW * GD;
  grid_grad_stage2<<<(unsigned)ncell, 256, 0, s>>>(static_cast<const float*>(ws), dgrid, GH, GW, GD,
                                                   C, pl.rg, pl.nyg, (float)GH / H);
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
