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
// pixels (f32 in, f32 accumulate, bit-equal to an fmaf chain).  A real reduction over
// K = pixels, not a reshaped gather -- the one place in this library where MFMA fits.  What it
// buys is the data movement (the z scatter becomes dense rows; no atomics, deterministic), NOT
// arithmetic rate: on gfx950 the f32-input MFMA runs at the VALU's own 32 FMA / cycle / SIMD and
// does not overlap with VALU work of any wave on that SIMD (tools/debug/ubench/
// mfma_valu_overlap.hip: MFMA-only 483 us + VALU-only 469 us -> 919 us interleaved), and the
// tile is 19 % dense (4 live weights x 12 channels of 16 x 16).  Stage 1 is therefore bound by
// FP32 issue: 16 MFMA (512 cycles) + ~115 VALU per 64-pixel chunk.
//
// Stage 1 (grid_grad_stage1): one workgroup of 4 waves owns one x-interval (all pixels with
//   gx0 == g, g = -1 .. GW-1) of RG consecutive rows; its waves take alternate rows.  Per row a
//   wave loads the interval's pixels (guide, input, dout: buffer loads, 2 chunks of 64 ahead),
//   and per chunk each lane stages ITS pixel's operands -- V (dout x [in; 1]) and A (the two x
//   corners folded into one row where they clamp onto the same column) -- in a private LDS
//   slab; the wave reads them back as MFMA operands.  The row's 16x16 result is scaled by its
//   two y weights into three REGISTER tiles (the <= 3 grid rows the group touches); after the
//   last row the four waves' tiles are added in fixed order and go to the workspace.  No LDS
//   accumulator, no atomics: the result is deterministic.
// Stage 2 (grid_grad_stage2): one workgroup per grid cell adds, in fixed order, the partial
//   tiles of the row groups and the two intervals that cover it.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;      // waves per workgroup; they share ONE task and split its rows
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

// ---- stage 1 ---------------------------------------------------------------------------------
// The operand slabs are stored k-major / channel-major ([16][kTStride] floats per wave):
//   * the A operand is written as a SCATTER of its <= 4 live entries (2 x corners x 2 z corners)
//     into a zeroed slab and re-zeroed after the MFMAs, instead of building 16 dense floats per
//     pixel with a select chain (the z tent has two live taps; the forced-1 half cells one);
//   * lane (sub, bc) of MFMA u takes pixel 16 * sub + u (any bijection of pixels onto (u, kk)
//     computes the same sum), so its 16 operand values are CONTIGUOUS: 4 ds_read_b128 per
//     operand instead of 16 ds_read_b32, bank-conflict-free with the 68-float row stride;
//   * x weights depend on (chunk, lane) only, not on the row: computed once per wave.
// CIN/COUT/OFFSET as in the forward; APPLY = false: V = dout (C = COUT channels).
// Wave-uniform row base + 32-bit per-lane byte offset: buffer loads keep the addressing on the
// scalar unit (global_load with 64-bit per-lane addresses cost ~2 extra VALU per load here) and
// fetch a pixel's channels in one instruction (dword alignment is all a buffer load needs).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}

template <int N>
__device__ __forceinline__ void buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float* dst) {
  if constexpr (N >= 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
    dst[2] = __uint_as_float(v.z); dst[3] = __uint_as_float(v.w);
    if constexpr (N > 4) buf_load<N - 4>(rs, byte_off + 16, dst + 4);
  } else if constexpr (N == 3) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, byte_off, 0, 0);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y); dst[2] = __uint_as_float(v.z);
  } else if constexpr (N == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
  } else if constexpr (N == 1) {
    dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
  }
}

constexpr int kTStride = 68;  // floats per operand row: 64 pixels + 4 (16-B aligned, 4-bank skew)

template <int CIN, int COUT, bool OFFSET, bool APPLY>
__global__ __launch_bounds__(kWaves * 64) void grid_grad_stage1(GGParams p) {
  constexpr int CJ = APPLY ? CIN + (OFFSET ? 1 : 0) : 1;
  constexpr int C = COUT * CJ;
  static_assert(C <= 16, "one 16-column MFMA tile");
  constexpr int CIN_Q = (APPLY && CIN > 0) ? CIN : 1;
  constexpr int kBatch = 2;  // chunks of 64 pixels loaded ahead (4: 132 VGPRs, 3 waves / SIMD, 6 % slower)
  constexpr int kSlab = (16 + C) * kTStride;  // floats per wave: A^T [16][68] then V^T [C][68]
  static_assert(kSlab >= kTileFloats, "the final reduction reuses the slabs");
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kSlab];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float* at = lds + wave * kSlab;    // A^T[k][px]
  float* vt = at + 16 * kTStride;    // V^T[c][px]
  {  // zero the A slab once; afterwards every chunk restores the entries it wrote
    f32x4* az = reinterpret_cast<f32x4*>(at);
#pragma unroll
    for (int q = 0; q < (16 * kTStride / 4 + 63) / 64; ++q)
      if (lane + 64 * q < 16 * kTStride / 4) az[lane + 64 * q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long long task = blockIdx.x;  // one (image, row group, x-interval) per workgroup
  const int nint = p.GW + 1;
  const int g = (int)(task % nint) - 1;  // gx0 of this workgroup's pixels
  const int yg = (int)((task / nint) % p.nyg);
  const long long b = task / ((long long)nint * p.nyg);
  const int x_lo = interval_start(g, p.W, p.scale_x);
  const int x_hi = interval_start(g + 1, p.W, p.scale_x);
  const int y_first = yg * p.rg, y_end = min(y_first + p.rg, p.H);
  const int gy_base = gy_base_of(y_first, p.scale_y, p.GH);
  const float gd_f = (float)p.GD;
  const bool fold_lo = g < 0, fold_hi = g >= p.GW - 1;
  const float gc0 = g + 0.5f, gc1 = g + 1 + 0.5f;

  // Two x weights of pixel x; columns that clamp onto each other (g = -1: corner 0 -> column 0
  // == corner 1; g = GW-1: corner 1 -> column GW-1 == corner 0) are folded into ONE A row, so
  // stage 2 never sees a column twice.  Pixels past the interval get zero weights.
  auto x_weights = [&](int x, float& w0, float& w1) {
    const float live = (x < x_hi) ? 1.0f : 0.0f;
    const float gxf = mul_rn((float)x + 0.5f, p.scale_x);
    const float wxa = tent_weight(gc0, gxf) * live;
    const float wxb = tent_weight(gc1, gxf) * live;
    w0 = fold_lo ? 0.0f : (fold_hi ? wxa + wxb : wxa);
    w1 = fold_lo ? wxa + wxb : (fold_hi ? 0.0f : wxb);
  };
  const int span = x_hi - x_lo;
  const int nbr = (span + 64 * kBatch - 1) / (64 * kBatch);  // batches per row
  float w0c[kBatch], w1c[kBatch];  // the common case nbr == 1: one batch per row, same x every row
#pragma unroll
  for (int cb = 0; cb < kBatch; ++cb) x_weights(x_lo + 64 * cb + lane, w0c[cb], w1c[cb]);

  // MFMA lane roles (v_mfma_f32_16x16x4_f32): A[k = lane & 15][kk = lane >> 4],
  // B[kk = lane >> 4][c = lane & 15], D[k = 4 * (lane >> 4) + r][c = lane & 15] in register r.
  const int sub = lane >> 4, bc = lane & 15;
  const f32x4* a_rd = reinterpret_cast<const f32x4*>(at + bc * kTStride + 16 * sub);
  const f32x4* v_rd = reinterpret_cast<const f32x4*>(vt + min(bc, C - 1) * kTStride + 16 * sub);

  f32x4 acc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Batch {
    float g[kBatch], in[kBatch][CIN_Q], d[kBatch][COUT];
  };
  const int nrows = (y_end - (y_first + wave) + kWaves - 1) / kWaves;
  const int nbt = (span > 0 && nrows > 0) ? nrows * nbr : 0;
  auto load_batch = [&](int t, Batch& bt) {
    const int r = t / nbr, bi = t - r * nbr;
    const int y = y_first + wave + r * kWaves;
    const size_t prow = ((size_t)b * p.H + y) * p.W;  // wave-uniform
    const __amdgpu_buffer_rsrc_t grs = row_rsrc(p.guide + prow);
    const __amdgpu_buffer_rsrc_t irs = row_rsrc((APPLY && CIN > 0) ? p.input + prow * CIN : p.guide);
    const __amdgpu_buffer_rsrc_t drs = row_rsrc(p.dout + prow * COUT);
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      // unconditional (clamped) loads: no exec-masked branch around VMEM keeps the compiler's
      // vmcnt counts exact; pixels past the interval carry zero x weights.
      const unsigned px = (unsigned)min(xb + 64 * cb + lane, x_hi - 1);
      buf_load<1>(grs, px * 4u, &bt.g[cb]);
      if constexpr (APPLY && CIN > 0) buf_load<CIN>(irs, px * (4u * CIN), bt.in[cb]);
      buf_load<COUT>(drs, px * (4u * COUT), bt.d[cb]);
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
        float w0 = w0c[cb], w1 = w1c[cb];
        if (nbr > 1) x_weights(x0 + lane, w0, w1);  // wave-uniform; only intervals wider than 256 px
        // z: only the two corners around gzf carry weight (:121); the outermost half cells are
        // forced to 1 (:122-125).  Two v_sqrt_f32 per pixel (1 ulp; argument >= 1e-8, no
        // denormals; a weight moves by <= 6e-8, far below the summation noise of a 30 000-term
        // reduction).  P = (za, wa), Q = (za + 1, wb); Q is written first, so where it clamps onto
        // P's slot (za == 7) P's value wins; there wb == 0 anyway.
        const float gzf = mul_rn(cur.g[cb], gd_f);  // gzf = guide * GD  (:120)
        const float fz = floorf(gzf - 0.5f);
        const float dza = (fz + 0.5f) - gzf, dzb = (fz + 1.5f) - gzf;
        float wP = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dza, dza, kSmoothEps)), 0.0f);
        float wQ = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dzb, dzb, kSmoothEps)), 0.0f);
        const bool lo = gzf < 0.5f, hi = gzf > gd_f - 0.5f;
        int zP = min(max((int)__builtin_amdgcn_fmed3f(fz, -2.0f, 9.0f), 0), 7), zQ = min(zP + 1, 7);
        if (lo) { zP = 0; zQ = 1; }
        if (hi) { zP = p.GD - 1; zQ = (p.GD - 1) ^ 1; }
        if (lo || hi) { wP = 1.0f; wQ = 0.0f; }
        float* aP = at + zP * kTStride + lane;
        float* aQ = at + zQ * kTStride + lane;
        aQ[0] = w0 * wQ;
        aQ[8 * kTStride] = w1 * wQ;
        aP[0] = w0 * wP;
        aP[8 * kTStride] = w1 * wP;
        // V^T[c][px]: dout x [in; 1] (slice: dout)
        if constexpr (APPLY) {
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
              vt[(i * CJ + j) * kTStride + lane] = (j < CIN) ? cur.d[cb][i] * cur.in[cb][j < CIN ? j : 0] : cur.d[cb][i];
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) vt[c * kTStride + lane] = cur.d[cb][c];
        }
        wave_lds_order();
        // D[k, c] += sum_px A[k, px] * V[px, c]; two accumulators break the dependent-issue chain.
        f32x4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          av[q] = a_rd[q];
          bv[q] = v_rd[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][0], bv[q][0], dacc, 0, 0, 0);
          dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][1], bv[q][1], dacc2, 0, 0, 0);
          dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][2], bv[q][2], dacc, 0, 0, 0);
          dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][3], bv[q][3], dacc2, 0, 0, 0);
        }
        wave_lds_order();
        aQ[0] = 0.0f;
        aQ[8 * kTStride] = 0.0f;
        aP[0] = 0.0f;
        aP[8 * kTStride] = 0.0f;
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
  float* red = lds + wave * kSlab;
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
    for (int w = 1; w < kWaves; ++w) sum += lds[w * kSlab + e];
    dst[e] = sum;
  }
}

// ---- stage 1, sorted form ----------------------------------------------------------------------
// The dense 16 x 16 x 4 tile above spends 81 % of its MACs on zeros: of the 16 rows k = (x corner,
// gz) a pixel has 4 live ones (2 x corners x 2 z taps).  Here the MFMA is the 16-block
// v_mfma_f32_4x4x1_16B_f32: every block is ONE pixel's outer product
//     (4 weights: x corner x z tap)  (x)  (4 channels)
// accumulated into the block's own 4 x 4 registers -- every MAC is live.  A block's accumulator can
// only serve pixels of ONE z bin (the lower tap's plane zP), so per 64-pixel chunk:
//   1. bins: 8 ballots give the per-bin counts (SGPRs) and each lane's rank inside its bin;
//   2. block slots: an MFMA has 4 pixel slots (x 4 channel groups = 16 blocks); with two accumulator
//      sets there are 8 slots.  The nb non-empty bins get P = 8 / 4 / 2 / 1 slots each (nb <= 1, 2,
//      4, 8), pixel of rank r goes to part r & (P-1), round r >> log2 P -- a smooth guide (one or two
//      bins per chunk) is spread over all slots, a noisy one uses a slot per bin;
//   3. each lane writes its pixel's 4 weights + C channel values to its slot's record area in LDS
//      ([slot][component][round], so an MFMA lane reads 4 rounds of its operand as one ds_read_b128);
//      the weight rows are zero-filled first, which is all the padding there is;
//   4. max-over-slots rounds of MFMAs per set; 5. the two accumulators are flushed into the wave's
//      row tile [x corner][z plane 0..8][c] with ds_add_f32 (a wave's LDS atomics execute in program
//      and lane order: deterministic), plane 8 being the sink of the upper tap of plane 7.
// At the end of a row the row tile is folded, scaled by the two y weights, into the same register
// tiles as before; stage 2 is unchanged.  Per chunk: ~24 MFMA x 8 cycles + ~100 VALU instead of
// 16 MFMA x 32 cycles + ~115 VALU.
constexpr int kRounds = 16;                       // rounds per pass = record capacity of a slot
constexpr int kRowTileFloats = 2 * 9 * 16;        // [x corner][z plane 0..8][c 16]

template <int CIN, int COUT, bool OFFSET, bool APPLY>
__global__ __launch_bounds__(kWaves * 64) void grid_grad_stage1_sorted(GGParams p) {
  constexpr int CJ = APPLY ? CIN + (OFFSET ? 1 : 0) : 1;
  constexpr int C = COUT * CJ;
  static_assert(C <= 16, "16 channel columns");
  constexpr int CQ = (C + 3) / 4;                 // channel groups of 4 (MFMA block columns)
  constexpr int NCOMP = 4 + 4 * CQ;               // record components: 4 weights + channels
  constexpr int kSlotFloats = NCOMP * kRounds;    // [component][round]
  constexpr int kRecFloats = 8 * kSlotFloats;     // 8 slots
  constexpr int CIN_Q = (APPLY && CIN > 0) ? CIN : 1;
  constexpr int kBatch = 2;
  constexpr int kWaveFloats = kRecFloats + kRowTileFloats + 8;  // + slot -> bin table
  static_assert(kWaveFloats >= kTileFloats, "the final reduction reuses the wave's LDS");
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kWaveFloats];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float* rec = lds + wave * kWaveFloats;          // records
  float* rowt = rec + kRecFloats;                 // row tile
  int* slotbin = reinterpret_cast<int*>(rowt + kRowTileFloats);
  {  // zero everything once (stale record values must at least be finite)
    f32x4* z4 = reinterpret_cast<f32x4*>(rec);
    for (int e = lane; e < kWaveFloats / 4; e += 64) z4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long long task = blockIdx.x;
  const int nint = p.GW + 1;
  const int g = (int)(task % nint) - 1;
  const int yg = (int)((task / nint) % p.nyg);
  const long long b = task / ((long long)nint * p.nyg);
  const int x_lo = interval_start(g, p.W, p.scale_x);
  const int x_hi = interval_start(g + 1, p.W, p.scale_x);
  const int y_first = yg * p.rg, y_end = min(y_first + p.rg, p.H);
  const int gy_base = gy_base_of(y_first, p.scale_y, p.GH);
  const float gd_f = (float)p.GD;
  const bool fold_lo = g < 0, fold_hi = g >= p.GW - 1;
  const float gc0 = g + 0.5f, gc1 = g + 1 + 0.5f;
  auto x_weights = [&](int x, float& w0, float& w1) {
    const float live = (x < x_hi) ? 1.0f : 0.0f;
    const float gxf = mul_rn((float)x + 0.5f, p.scale_x);
    const float wxa = tent_weight(gc0, gxf) * live;
    const float wxb = tent_weight(gc1, gxf) * live;
    w0 = fold_lo ? 0.0f : (fold_hi ? wxa + wxb : wxa);
    w1 = fold_lo ? wxa + wxb : (fold_hi ? 0.0f : wxb);
  };
  const int span = x_hi - x_lo;
  const int nbr = (span + 64 * kBatch - 1) / (64 * kBatch);
  float w0c[kBatch], w1c[kBatch];
#pragma unroll
  for (int cb = 0; cb < kBatch; ++cb) x_weights(x_lo + 64 * cb + lane, w0c[cb], w1c[cb]);

  // MFMA lane roles (v_mfma_f32_4x4x1_16B_f32): block = lane >> 2 = 4 * (pixel slot r) + (channel
  // group q); A[i = lane & 3] = weight i of the slot's pixel, B[j = lane & 3] = channel 4 q + j;
  // D register v of lane (block, j) = row i = v, column j.
  const int mr = lane >> 4, mq = (lane >> 2) & 3, mi = lane & 3;
  const int mqc = min(mq, CQ - 1);  // surplus channel groups re-read the last one; never flushed
  // set s: slot 4 s + mr
  const f32x4* a_rd[2] = {reinterpret_cast<const f32x4*>(rec + (0 + mr) * kSlotFloats + mi * kRounds),
                          reinterpret_cast<const f32x4*>(rec + (4 + mr) * kSlotFloats + mi * kRounds)};
  const f32x4* b_rd[2] = {
      reinterpret_cast<const f32x4*>(rec + (0 + mr) * kSlotFloats + (4 + 4 * mqc + mi) * kRounds),
      reinterpret_cast<const f32x4*>(rec + (4 + mr) * kSlotFloats + (4 + 4 * mqc + mi) * kRounds)};
  const bool flusher = mq < CQ && (4 * mq + mi) < C;

  // final-tile lane roles (as the dense kernel: D[k = 4 * sub + r][c = bc])
  const int sub = lane >> 4, bc = lane & 15;
  f32x4 acc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Batch {
    float g[kBatch], in[kBatch][CIN_Q], d[kBatch][COUT];
  };
  const int nrows = (y_end - (y_first + wave) + kWaves - 1) / kWaves;
  const int nbt = (span > 0 && nrows > 0) ? nrows * nbr : 0;
  auto load_batch = [&](int t, Batch& bt) {
    const int r = t / nbr, bi = t - r * nbr;
    const int y = y_first + wave + r * kWaves;
    const size_t prow = ((size_t)b * p.H + y) * p.W;
    const __amdgpu_buffer_rsrc_t grs = row_rsrc(p.guide + prow);
    const __amdgpu_buffer_rsrc_t irs = row_rsrc((APPLY && CIN > 0) ? p.input + prow * CIN : p.guide);
    const __amdgpu_buffer_rsrc_t drs = row_rsrc(p.dout + prow * COUT);
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      const unsigned px = (unsigned)min(xb + 64 * cb + lane, x_hi - 1);
      buf_load<1>(grs, px * 4u, &bt.g[cb]);
      if constexpr (APPLY && CIN > 0) buf_load<CIN>(irs, px * (4u * CIN), bt.in[cb]);
      buf_load<COUT>(drs, px * (4u * COUT), bt.d[cb]);
    }
  };

  Batch cur, nxt;
  if (nbt > 0) load_batch(0, cur);
  for (int t = 0; t < nbt; ++t) {
    if (t + 1 < nbt) load_batch(t + 1, nxt);
    const int r = t / nbr, bi = t - r * nbr;
    const int y = y_first + wave + r * kWaves;
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      const int x0 = xb + 64 * cb;
      if (x0 < x_hi) {  // wave-uniform
        float w0 = w0c[cb], w1 = w1c[cb];
        if (nbr > 1) x_weights(x0 + lane, w0, w1);
        const bool live = x0 + lane < x_hi;
        // z taps (:120-125), as in the dense kernel: P = (zP, wP), Q = (zP + 1, wQ); the outermost
        // half cells are forced to (1, 0).
        const float gzf = mul_rn(cur.g[cb], gd_f);
        const float fz = floorf(gzf - 0.5f);
        const float dza = (fz + 0.5f) - gzf, dzb = (fz + 1.5f) - gzf;
        float wP = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dza, dza, kSmoothEps)), 0.0f);
        float wQ = std_max(1.0f - __builtin_amdgcn_sqrtf(fmaf(dzb, dzb, kSmoothEps)), 0.0f);
        const bool lo = gzf < 0.5f, hi = gzf > gd_f - 0.5f;
        int zP = min(max((int)__builtin_amdgcn_fmed3f(fz, -2.0f, 9.0f), 0), p.GD - 1);
        if (lo) zP = 0;
        if (hi) zP = p.GD - 1;
        if (lo || hi) { wP = 1.0f; wQ = 0.0f; }
        const int bin = live ? zP : 8;  // 8: no record

        // 1. per-bin counts (uniform) and this lane's rank inside its bin
        int cnt[8];
        int rank = 0;
        unsigned nm = 0;  // non-empty bins
#pragma unroll
        for (int z = 0; z < 8; ++z) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(bin == z);
          cnt[z] = __builtin_popcountll(m);
          const int rk = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          rank = (bin == z) ? rk : rank;
          nm |= (cnt[z] > 0 ? 1u : 0u) << z;
        }
        // 2. slots: nb non-empty bins x P parts
        const int nb = __builtin_popcount(nm);
        const int lgP = nb <= 1 ? 3 : (nb <= 2 ? 2 : (nb <= 4 ? 1 : 0));
        const int nzidx = __builtin_popcount(nm & ((1u << (bin & 7)) - 1u));
        const int slot = (nzidx << lgP) + (rank & ((1 << lgP) - 1));
        const int idx = rank >> lgP;
        int rounds0 = 0, rounds1 = 0;  // uniform: rounds needed by the slots of set 0 / set 1
        {
          int k = 0;
#pragma unroll
          for (int z = 0; z < 8; ++z) {
            const int len = (cnt[z] + (1 << lgP) - 1) >> lgP;
            const int s0 = k << lgP;  // first slot of this bin (if non-empty)
            if (cnt[z] > 0) {
              if (s0 < 4) rounds0 = max(rounds0, len);
              if (s0 + (1 << lgP) > 4) rounds1 = max(rounds1, len);
              ++k;
            }
          }
        }
        // record components: weights a[i = 2 * xcorner + tap], channels V[c]
        float comp[NCOMP];
        comp[0] = w0 * wP;
        comp[1] = w0 * wQ;
        comp[2] = w1 * wP;
        comp[3] = w1 * wQ;
#pragma unroll
        for (int c = 0; c < 4 * CQ; ++c) comp[4 + c] = 0.0f;
        if constexpr (APPLY) {
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
#pragma unroll
            for (int j = 0; j < CJ; ++j)
              comp[4 + i * CJ + j] = (j < CIN) ? cur.d[cb][i] * cur.in[cb][j < CIN ? j : 0] : cur.d[cb][i];
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) comp[4 + c] = cur.d[cb][c];
        }
        if (live) slotbin[slot] = zP;  // every pixel of a slot writes the same value

        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        const int maxr = max(rounds0, rounds1);
        for (int pass = 0; pass * kRounds < maxr; ++pass) {  // uniform; one pass unless a slot holds > 16 px
          // 3. zero the weight rows of all 8 slots, then scatter this pass's records
          {
            const int f = lane;  // 8 slots x 4 weight rows x 16 rounds = 128 float4
            f32x4* zr0 = reinterpret_cast<f32x4*>(rec + (f >> 4) * kSlotFloats + (f & 15) * 4);
            f32x4* zr1 = reinterpret_cast<f32x4*>(rec + ((f + 64) >> 4) * kSlotFloats + (f & 15) * 4);
            *zr0 = f32x4{0.f, 0.f, 0.f, 0.f};
            *zr1 = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if (live && (idx >> 4) == pass) {
            float* rp = rec + slot * kSlotFloats + (idx & (kRounds - 1));
#pragma unroll
            for (int c = 0; c < 4 + C; ++c) rp[c * kRounds] = comp[c];
          }
          wave_lds_order();
          // 4. rounds of MFMAs, 4 per ds_read_b128 pair; the two sets alternate
          const int r0n = min(max(rounds0 - pass * kRounds, 0), kRounds);
          const int r1n = min(max(rounds1 - pass * kRounds, 0), kRounds);
#pragma unroll
          for (int u4 = 0; u4 < kRounds / 4; ++u4) {
            if (4 * u4 < r0n) {  // uniform
              const f32x4 av = a_rd[0][u4], bv = b_rd[0][u4];
#pragma unroll
              for (int u = 0; u < 4; ++u) d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u], bv[u], d0, 0, 0, 0);
            }
            if (4 * u4 < r1n) {
              const f32x4 av = a_rd[1][u4], bv = b_rd[1][u4];
#pragma unroll
              for (int u = 0; u < 4; ++u) d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[u], bv[u], d1, 0, 0, 0);
            }
          }
          wave_lds_order();
        }
        // 5. flush: lane (slot mr, group mq, column mi) adds D[i][mi], i = (xcorner, tap), into
        //    row tile [xcorner][bin + tap][4 mq + mi]
        if (flusher) {
          float* t0 = rowt + slotbin[mr] * 16 + 4 * mq + mi;
          float* t1 = rowt + slotbin[4 + mr] * 16 + 4 * mq + mi;
          if (rounds0 > 0) {
            __hip_atomic_fetch_add(t0, d0[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t0 + 16, d0[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t0 + 9 * 16, d0[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t0 + 9 * 16 + 16, d0[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          if (rounds1 > 0) {
            __hip_atomic_fetch_add(t1, d1[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t1 + 16, d1[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t1 + 9 * 16, d1[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(t1 + 9 * 16 + 16, d1[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        }
        wave_lds_order();
      }
    }
    if (bi == nbr - 1) {
      // end of the row: fold the row tile, scaled by the row's two y weights, into the register
      // tiles of the (<= 3) grid rows the group touches, and clear it.
      const float gyf = mul_rn(y + 0.5f, p.scale_y);
      const int gy0 = floor_to_int(gyf - 0.5f);
      const float wy0 = tent_weight(gy0 + 0.5f, gyf);
      const float wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
      const int rel0 = clamp_index(gy0, 0, p.GH - 1) - gy_base;
      const int rel1 = clamp_index(gy0 + 1, 0, p.GH - 1) - gy_base;
      // lane (sub, bc): rows k = 4 sub + q -> x corner sub >> 1, plane 4 (sub & 1) + q
      float* tp = rowt + ((sub >> 1) * 9 + 4 * (sub & 1)) * 16 + bc;
      f32x4 dacc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dacc[q] = tp[q * 16];
        tp[q * 16] = 0.0f;
      }
      if (lane < 32) rowt[((lane >> 4) * 9 + 8) * 16 + (lane & 15)] = 0.0f;  // the sink plane
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const float sr = (rel0 == rr ? wy0 : 0.0f) + (rel1 == rr ? wy1 : 0.0f);
        acc[rr] += sr * dacc;
      }
      wave_lds_order();
    }
    cur = nxt;
  }
  __syncthreads();
  float* red = lds + wave * kWaveFloats;
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
    for (int w = 1; w < kWaves; ++w) sum += lds[w * kWaveFloats + e];
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
                     int H, int W, int GH, int GW, int GD, void* ws, const GGPlan& pl, hipStream_t s,
                     bool dense) {
  constexpr int C = APPLY ? COUT * (CIN + (OFFSET ? 1 : 0)) : COUT;
  GGParams p{guide, input, dout, static_cast<float*>(ws), H, W, GH, GW, GD,
             pl.rg, pl.nyg, pl.ntasks, (float)GW / W, (float)GH / H};
  const long long nblocks = pl.ntasks;
  if (dense)
    grid_grad_stage1<CIN, COUT, OFFSET, APPLY><<<(unsigned)nblocks, kWaves * 64, 0, s>>>(p);
  else
    grid_grad_stage1_sorted<CIN, COUT, OFFSET, APPLY><<<(unsigned)nblocks, kWaves * 64, 0, s>>>(p);
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
  const bool dense = a.variant == 1;
  *name = dense ? "grid_grad_mfma/dense" : "grid_grad_mfma";
#define HDRNET_CASE(CI, CO, OFF)                                                                  \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF)                                         \
  return gg_launch<CI, CO, OFF, true>(a.guide, a.input, a.dout, a.dgrid, a.B, a.H, a.W, a.GH, a.GW, \
                                      a.GD, a.workspace, pl, s, dense)
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
  const bool dense = a.variant == 1;
  *name = dense ? "grid_grad_mfma/dense" : "grid_grad_mfma";
#define HDRNET_CASE(CC)                                                                            \
  if (a.C == CC)                                                                                   \
  return gg_launch<0, CC, false, false>(a.guide, nullptr, a.dout, a.dgrid, a.B, a.H, a.W, a.GH, a.GW, \
                                        a.GD, a.workspace, pl, s, dense)
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
