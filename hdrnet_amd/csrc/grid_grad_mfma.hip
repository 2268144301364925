// Grid VJP (dgrid) of BilateralSliceApply / BilateralSlice for gfx950, with the per-pixel VJPs (dguide, dinput)
// optionally fused into the same pass.
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
// does not overlap with VALU work of any wave on that SIMD, and the tile is 19 % dense (4 live
// weights x 12 channels of 16 x 16): stage 1 is bound by FP32 issue.
// Grids of 9 .. 16 planes (luma_bins = 16) take two such tiles per task (template parameter NH).
//
// Stage 1 (grid_grad_stage1): one workgroup of 4 waves owns one x-interval (all pixels with
//   gx0 == g, g = -1 .. GW-1) of rg consecutive rows (rg fitted per launch to whole rounds of
//   resident workgroups, gg_plan); its waves take alternate rows.  Per row a wave loads the
//   interval's pixels (guide, input, dout: buffer loads, one batch of 2 chunks of 64 ahead), and
//   per chunk each lane stages ITS pixel's operands -- V (dout x [in; 1]) and A (x corner 0 in rows
//   0-7, corner 1 in rows 8-15) -- in a private LDS slab; the wave reads them back as MFMA
//   operands (conflict-free ds_read_b64 pattern).  The row's 16x16 result is scaled by its two y
//   weights into three REGISTER tiles (the <= 3 grid rows the group touches); after the last row
//   the four waves' tiles are added in fixed order and go to the workspace.  No LDS accumulator,
//   no atomics: the result is deterministic.  With dguide / dinput requested (WG / WI) the same
//   pass also evaluates the per-pixel VJPs from a per-wave coefficient image.
// Stage 2 (grid_grad_stage2): one workgroup per grid column (all planes) adds, in fixed order, the partial
//   tiles of the row groups and the two intervals that cover it (+ the clamp-to-edge halves of the
//   border intervals).
//
// What was tried and measured on the way here (the sorted 4x4x1-MFMA form, the bf16 split, launch shapes,
// folding stage 2 into stage 1, ...): docs/EXPERIMENTS.md section 4.2 and "Kernel-file lab notes".
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using rows::vjp_blend;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SPLIT operands (tools variant 2): an f32 value x travels as ONE dword {hi, lo} of two bf16.  Two values at once
// (round 5; round 2's form took 4 instructions per value): hi = the upper 16 bits (truncation, so the remainder x - hi
// is exact), lo = the remainder ROUNDED to bf16 (v_cvt_pk_bf16_f32, both values in one instruction): x = hi + lo to
// 2^-17 relative, unbiased (a truncated lo -- 3 instructions per value, another 2-5 % faster -- biases every product by
// ~-1e-5).  Per pair: 2 v_and, 1 v_pk_add (negated), 1 cvt, 2 v_perm.
typedef float f32x2_sp __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_sp __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_sp split_pack2(f32x2_sp x) {
  const unsigned u0 = __float_as_uint(x.x), u1 = __float_as_uint(x.y);
  const f32x2_sp hi = {__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)};
  const f32x2_sp lo = x - hi;
  const unsigned lp = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_sp));  // {bf16(lo0), bf16(lo1)}
  return f32x2_sp{__uint_as_float(__builtin_amdgcn_perm(lp, u0, 0x05040302u)),   // {u0[31:16], lp[15:0]}
                  __uint_as_float(__builtin_amdgcn_perm(lp, u1, 0x07060302u))};  // {u1[31:16], lp[31:16]}
}

__device__ __forceinline__ float split_pack(float x) { return split_pack2(f32x2_sp{x, x}).x; }

// SPLIT = 2 (tools variant 10, round 6): the f16 form with fp32-grade products.  x travels as one dword {hi, lo'} of two
// f16: hi = f16(x) (round to nearest: the remainder x - hi is exact in f32), lo' = f16((x - hi) * 2^11) -- the remainder
// re-scaled into f16's normal range, so hi + 2^-11 lo' carries ~22 bits.  [a_hi, 0] x [v_hi, v_lo'] accumulates hi hi into
// one tile, [a_hi, a_lo'] x [v_lo', v_hi] (the dword rotated) the two cross terms hi lo' + lo' hi into a second; the row
// fold combines them as acc0 + 2^-11 acc1.  The dropped lo' lo' term is 2^-22 relative.  Per pair: 2 cvt_pk, 2 cvt back,
// 1 v_pk_add, 1 v_pk_mul, 2 v_perm = 4 per value.  RANGE: f16 holds 6e-8 .. 65504 -- V = dout x [in; 1] of a real
// training step (dout ~ 1 / pixels ~ 1e-7) sits in f16's denormals, where hi keeps a few bits and only lo' is normal:
// ~12 bits in all (measured: profiles/r06/bwd_steps.md).  An experiment, not a product path.
typedef _Float16 f16x2_sp __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x2_sp split_pack2_f16(f32x2_sp x) {
  const f16x2_sp h = __builtin_convertvector(x, f16x2_sp);
  const f32x2_sp hf = __builtin_convertvector(h, f32x2_sp);
  const f32x2_sp r = (x - hf) * f32x2_sp{2048.0f, 2048.0f};
  const f16x2_sp l = __builtin_convertvector(r, f16x2_sp);
  const unsigned hp = __builtin_bit_cast(unsigned, h), lp = __builtin_bit_cast(unsigned, l);
  return f32x2_sp{__uint_as_float(__builtin_amdgcn_perm(lp, hp, 0x05040100u)),   // {lp[15:0], hp[15:0]}: hi in the low half
                  __uint_as_float(__builtin_amdgcn_perm(lp, hp, 0x07060302u))};  // {lp[31:16], hp[31:16]}
}

constexpr int kWaves = 4;      // waves per workgroup; they share ONE task and split its rows
constexpr int kTileFloats = 3 * 16 * 16;  // partial tile: [rel 3][k 16][c 16]
constexpr int kTraceChunks = 16;          // tools phase trace: chunks recorded per workgroup

struct GGParams {
  const float* guide;
  const float* input;  // null for slice
  const float* dout;
  const float* grid;   // fused backward only (WG || WI)
  float* dguide;       // fused backward: [B][H][W] or null
  float* dinput;       // fused backward: [B][H][W][CIN] or null
  float* partial;  // [B][nyg][GW + 1][3][16][16]
  int H, W, GH, GW, GD;
  int rg, nyg;
  long long ntasks;
  float scale_x, scale_y;  // GW / W, GH / H  (forward's expressions)
  long long* trace;        // tools variant 9 (ABL 6): per-chunk phase stamps of wave 0, [task][kTraceChunks][5]
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

// live = false: a descriptor of ZERO records -- a load through it touches no memory and returns 0, but it is
// still one VMEM instruction (see load_batch: the count of VMEM operations must not depend on the path).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const float* base, bool live = true) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, live ? 0x7fffffff : 0, 0x00020000);
}

// AUX: cache policy (rows_common.hip.h: kAuxNt for data that is read exactly once).
template <int N, int AUX = 0>
__device__ __forceinline__ void buf_load(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float* dst) {
  if constexpr (N >= 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, AUX);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
    dst[2] = __uint_as_float(v.z); dst[3] = __uint_as_float(v.w);
    if constexpr (N > 4) buf_load<N - 4, AUX>(rs, byte_off + 16, dst + 4);
  } else if constexpr (N == 3) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, byte_off, 0, AUX);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y); dst[2] = __uint_as_float(v.z);
  } else if constexpr (N == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, AUX);
    dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
  } else if constexpr (N == 1) {
    dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, AUX));
  }
}

constexpr int kTStride = 68;  // floats per operand row: 64 pixels + 4 (16-B aligned, 4-bank skew)

// SPLIT = true (tools variant 2): the contraction runs on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 16x the
// f32-input rate) with every f32 operand split into two bf16 terms (split_pack2 above).  A K-slot pair is one pixel's
// {hi, lo}: [a_hi, a_lo] x [v_hi, v_lo] gives hi hi + lo lo, the same A against the dword ROTATED by 16 bits gives the
// two cross terms -- all four products, f32 accumulate.  Per 64-pixel chunk 8 MFMAs of ~17 cycles instead of 16 of 32,
// for 3.5 VALU per staged value + one rotate per B dword (72 per chunk; round 2's form: 96, and no gain).  Measured in
// round 5 (profiles/r05/bwd_steps.md): dgrid -8 %, dgrid + dguide -4 %, all three -2 % at 4K -- and 6e-6 of dgrid's
// scale away from the exact-f32 contraction, as much as the summation-order noise the parity bar (1e-5 x scale) is set
// against.  Not the product for that reason; the product's contraction is bit-equal to an fmaf chain.
//
// WG / WI (FUSED BACKWARD): the same pass also produces the per-pixel VJPs -- dguide (WG) and dinput
// (WI), bilateral_slice_apply.cc:140-259 -- from the pixel data it has loaded anyway (guide, input,
// dout are read ONCE for all three gradients: 28 B/px in, 16 B/px out, instead of two kernels reading
// 28 B/px each).  In this geometry every pixel of the workgroup has gx0 == g, so the forward's
// y-pre-lerped coefficient image shrinks to the two grid columns (g, g + 1) (clamped), padded in z
// like apply_fwd_seg.hip: 2 x (GD + 2) x C floats per wave, re-blended by the wave at the start of
// each row from grid rows it prefetched with the row's first pixel batch.  The z tent and its
// derivative share one v_sqrt_f32 per tap with the dgrid weights.
//
// NH (plane halves): the 16 rows of the A tile are (x corner, plane 0 .. 7).  A grid of 9 .. 16 planes (luma_bins = 16,
// hdrnet/bin/train.py:235) is NH = 2 tiles per task, planes 0 .. 7 and 8 .. 15, each with its own accumulators; per
// 64-pixel chunk the wave ballots which halves hold a live tap and contracts only those (an image-like guide rarely
// straddles plane 7 | 8 inside 64 neighbouring pixels: then the chunk costs what it costs at GD <= 8).  Both halves go
// through the SAME 16-row A slab, scattered, contracted and re-zeroed once per live half.
//
// CW (column windows): the tile has 16 columns.  A shape of 17 .. 32 grid channels (4 -> 4 with offset: C = 20) is two
// launches of this kernel, window CW staging the channels 16 CW .. 16 CW + 15 as its V rows and writing its own partial
// tiles; stage 2 runs once per window.  Each launch re-reads the pixels (2 x 28 B/px): a shape outside the reference's
// configurations, kept off the ~100 x slower generic gather rather than tuned.  dgrid only (the fused VJPs read whole
// coefficient vectors).
template <int CIN, int COUT, bool OFFSET, bool APPLY, int SPLIT, bool WG = false, bool WI = false, int ABL = 0, int NH = 1, int CW = 0>
__global__ __launch_bounds__(kWaves * 64)
__attribute__((amdgpu_waves_per_eu(((WG || WI) && COUT * (APPLY ? CIN + (OFFSET ? 1 : 0) : 1) <= 12) ? ((NH == 1 && SPLIT == 0) ? 4 : 3) : 1))) void grid_grad_stage1(GGParams p) {
  constexpr int CJ = APPLY ? CIN + (OFFSET ? 1 : 0) : 1;
  constexpr int CFULL = COUT * CJ;
  constexpr int C0 = 16 * CW;                                  // first grid channel of this window
  constexpr int C = CFULL - C0 < 16 ? CFULL - C0 : 16;         // columns of this window's tile
  static_assert(C >= 1 && C <= 16 && CFULL <= 32, "one 16-column MFMA tile per window, two windows");
  static_assert(CFULL <= 16 || (!WG && !WI && SPLIT == 0 && ABL == 0 && APPLY), "channel windows: apply dgrid only");
  static_assert(NH == 1 || NH == 2, "8 or 16 planes");
  static_assert(NH == 1 || (!SPLIT && ABL == 0), "the tools variants exist for GD <= 8 only");
  constexpr bool FUSED = WG || WI;
  static_assert(!FUSED || C % 4 == 0, "fused backward: float4 coefficient vectors");
  static_assert(!WI || (APPLY && CIN > 0), "dinput needs an input");
  constexpr int CB = C * (int)sizeof(float);
  constexpr int CIN_Q = (APPLY && CIN > 0) ? CIN : 1;
  // chunks of 64 pixels loaded ahead (4: 132 VGPRs, 3 waves / SIMD, 6 % slower; 1 with dinput fused, which
  // then fits 4 waves / SIMD instead of 3: 113 -> 120 us, the shorter prefetch costs more than the wave
  // buys).  A wide BilateralSlice (dout = 48-64 B/px) is the exception: its batch is 26-34 registers, one
  // chunk ahead brings the pass from 3 to 4 waves / SIMD and 115 -> 110 us at 4K.
  constexpr int kBatch = (!APPLY && COUT >= 12) ? 1 : 2;
  constexpr int kLoadAux = FUSED ? rows::kAuxNt : 0;  // fused pass is an HBM stream: nontemporal pixel loads
  // [A, D = x difference, dzA = z difference of A, dzD][plane 0 .. GD + 1 (GD <= 8 NH)][c]
  constexpr int kImg = FUSED ? 4 * (8 * NH + 2) * C : 0;
  constexpr int kSlabOps = (16 + C) * kTStride + kImg;  // floats per wave: A^T [16][68], V^T [C][68], image
  constexpr int kSlab = kSlabOps >= NH * kTileFloats ? kSlabOps : NH * kTileFloats;  // the final reduction reuses the slabs
  __shared__ __attribute__((aligned(16))) float lds[kWaves * kSlab];
  if constexpr (ABL == 5) return;  // tools ablation: the launch alone
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float* at = lds + wave * kSlab;    // A^T[k][px]
  float* vt = at + 16 * kTStride;    // V^T[c][px]
  float* img = vt + C * kTStride;    // fused: coefficient image of the current row
  // one (image, row group, x-interval) per workgroup: grid = (GW + 1, nyg, B)
  const int nint = p.GW + 1;
  const int g = (int)blockIdx.x - 1;  // gx0 of this workgroup's pixels
  const int yg = blockIdx.y;
  const long long b = blockIdx.z;
  const long long task = ((long long)b * p.nyg + yg) * nint + blockIdx.x;
  const int x_lo = interval_start(g, p.W, p.scale_x);
  const int x_hi = interval_start(g + 1, p.W, p.scale_x);
  const int y_first = yg * p.rg, y_end = min(y_first + p.rg, p.H);
  const int gy_base = gy_base_of(y_first, p.scale_y, p.GH);
  const float gd_f = (float)p.GD;
  const float gc0 = g + 0.5f;

  // The forward's two x weights of pixel x (corner columns g, g + 1; un-clamped weights, :58-68) from ONE
  // cached float per (chunk-in-row, lane): dx = (g + 0.5) - gxf.  w0 = max(1 - |dx|, 0), and corner 1's
  // offset (g + 1.5) - gxf equals dx + 1 exactly for gxf >= 1 (both differences are exact) and to 3e-8
  // below.  Pixels past the interval carry dx = 2: both weights 0.  A rows 0-7 always carry corner 0 and
  // rows 8-15 corner 1; where a corner's column clamps onto the other's (g = -1, g = GW - 1) stage 2 adds
  // that tile half to the edge column.
  auto x_offset = [&](int x) {
    const float gxf = mul_rn((float)x + 0.5f, p.scale_x);
    return (x < x_hi) ? gc0 - gxf : 2.0f;
  };
  const int span = x_hi - x_lo;
  const int nbr = (span + 64 * kBatch - 1) / (64 * kBatch);  // batches per row
  // fused: this lane's element of the row's coefficient image (2 columns x GD planes x C / 4 float4)
  // (element e = lane + 64 s of slot s; NH = 2: up to 2 x 16 x 4 = 128 elements, two per lane)
  constexpr int C4 = C / 4 > 0 ? C / 4 : 1;
  constexpr int NSLOT = NH;
  const int nst = 2 * p.GD * C4;
  // (decoded for min(e, nst - 1): every lane issues the two staging loads -- a VMEM op under an exec
  //  mask makes the compiler's vmcnt bookkeeping fall back to vmcnt(0), which drains the prefetch)
  int st_col[NSLOT], st_src[NSLOT], st_z[NSLOT], st_dst[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) {
    const int st_e = min(lane + 64 * sl, nst - 1);
    st_col[sl] = st_e / (p.GD * C4);
    const int st_rem = st_e - st_col[sl] * (p.GD * C4);
    st_src[sl] = (min(max(g + st_col[sl], 0), p.GW - 1) * p.GD * C4 + st_rem);  // float4 index in a grid row
    st_z[sl] = st_rem / C4;
    st_dst[sl] = (st_col[sl] * (p.GD + 2) + st_z[sl] + 1) * C4 + (st_rem - st_z[sl] * C4);
  }
  const float* grid_b = FUSED ? p.grid + (size_t)b * p.GH * p.GW * p.GD * C : nullptr;
  const int colb = (p.GD + 2) * CB;
  const float zhi = (float)(p.GD - 1);

  // MFMA lane roles (v_mfma_f32_16x16x4_f32): A[k = lane & 15][kk = lane >> 4],
  // B[kk = lane >> 4][c = lane & 15], D[k = 4 * (lane >> 4) + r][c = lane & 15] in register r.
  // Operand reads are ds_read_b64: its lane groups are {0-31}, {32-63} (ds_read_b128's are 4 x 16 in an
  // interleaved lane order under which the 68-float row stride is 2-way conflicted, measured as 45 % of
  // all LDS cycles: profiles/r02/exp7).  Lane (sub, bc) reads the pixel pairs 32 (sub >> 1) + 2 (sub & 1)
  // + 4 e, e = 0 .. 7: bank pair = 2 bc + (sub & 1) + const within a group -- conflict-free -- and MFMA
  // u = 2 e + i contracts pixel P(sub, e) + i (any bijection of the 64 pixels onto (u, kk) is the same sum).
  const int sub = lane >> 4, bc = lane & 15;
  const int rd_off = 32 * (sub >> 1) + 2 * (sub & 1);
  // They are issued from inline asm: the compiler pairs neighbouring ds_read_b64 into ds_read2_b64 (half
  // the rate, ds_read_b128-like lane groups), and `volatile` reads serialise read -> wait -> MFMA.
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned a_addr = (unsigned)(uintptr_t)((lds_float*)at + bc * kTStride + rd_off);
  const unsigned v_addr = (unsigned)(uintptr_t)((lds_float*)vt + min(bc, C - 1) * kTStride + rd_off);

  f32x4 acc[NH][3];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[h][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  struct Batch {
    float g[kBatch], in[kBatch][CIN_Q], d[kBatch][COUT];
  };
  const int nrows = (y_end - (y_first + wave) + kWaves - 1) / kWaves;
  const int nbt = (ABL != 3 && span > 0 && nrows > 0) ? nrows * nbr : 0;  // ABL 3: prologue + epilogue only
  // (row, batch in row) of the next batch to load / to contract: advanced incrementally -- `t / nbr` is an
  // integer division by a run-time value, ~20 instructions each on this target
  int ld_y = y_first + wave, ld_bi = 0;
  int pr_y = y_first + wave, pr_bi = 0;
  // `live` = false past the wave's last batch: the loads are still ISSUED (through zero-record descriptors).
  // s_waitcnt vmcnt(N) takes a compile-time N = the fewest VMEM operations any path can have issued after the
  // one being waited for; with the refill under `if (t + 1 < nbt)` that minimum was "none", so every chunk
  // waited for the batch just requested instead of the one requested a batch earlier -- the prefetch never
  // overlapped anything (profiles/r03/bwd_pmc_before.txt: 42 % of all wave cycles parked at s_waitcnt).
  auto load_batch = [&](Batch& bt, bool live = true) {
    const int y = ld_y, bi = ld_bi;
    if (++ld_bi == nbr) {
      ld_bi = 0;
      ld_y += kWaves;
    }
    const size_t prow = ((size_t)b * p.H + y) * p.W;  // wave-uniform
    const __amdgpu_buffer_rsrc_t grs = row_rsrc(p.guide + prow, live);
    const __amdgpu_buffer_rsrc_t irs = row_rsrc((APPLY && CIN > 0) ? p.input + prow * CIN : p.guide, live);
    const __amdgpu_buffer_rsrc_t drs = row_rsrc(p.dout + prow * COUT, live);
    const int xb = x_lo + bi * 64 * kBatch;
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      // unconditional (clamped) loads: no exec-masked branch around VMEM keeps the compiler's
      // vmcnt counts exact; pixels past the interval carry zero x weights.
      const unsigned px = (unsigned)min(xb + 64 * cb + lane, x_hi - 1);
      // nontemporal only where ONE instruction covers the wave's whole run of a tensor (<= 16 B per
      // pixel): an nt line is not kept for a second instruction touching it (profiles/r01, r02/exp6)
      buf_load<1, kLoadAux>(grs, px * 4u, &bt.g[cb]);
      // (24-bit multiplies -- pixel and plane indices are far below 2^24: v_mul_u32_u24 / v_mad_u32_u24 are
      //  full-rate, the 32-bit v_mul_lo_u32 / v_mad_u64_u32 the compiler picks otherwise quarter-rate)
      if constexpr (APPLY && CIN > 0) buf_load<CIN, (CIN <= 4 ? kLoadAux : 0)>(irs, __umul24(px, 4u * CIN), bt.in[cb]);
      buf_load<COUT, (COUT <= 4 ? kLoadAux : 0)>(drs, __umul24(px, 4u * COUT), bt.d[cb]);
    }
  };

  // kAhead batches are in flight beyond the one being contracted (a ring of kAhead + 1 register sets, the
  // loop unrolled over it).  Two ahead in the fused pass measured the same as one (118.5 vs 120 us all
  // three, 101 vs 101 us dgrid + dguide at 4K; profiles/r02/exp21): what tools variant 4 removes is the
  // exposed first load of every wave, not a too-short steady-state distance.
  constexpr int kAhead = 1;
  Batch ring[kAhead + 1];
  load_batch(ring[0], nbt > 0);  // issued first: the rest of the prologue runs under its latency
  if constexpr (kAhead > 1 && ABL != 1 && ABL != 4) load_batch(ring[1], nbt > 1);
  // fused: this lane's element of the two grid rows the coefficient image blends.  They change only when
  // gy0 does (once per cell height), so they stay in registers across the wave's rows.
  f32x4 sa[NSLOT], sb[NSLOT];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) sa[sl] = sb[sl] = f32x4{0.f, 0.f, 0.f, 0.f};
  int gy0_held = -0x7fffffff;
  auto load_grid_rows = [&](int gy0) {
    gy0_held = gy0;
    const int gy0c = clamp_index(gy0, 0, p.GH - 1), gy1c = clamp_index(gy0 + 1, 0, p.GH - 1);
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      sa[sl] = reinterpret_cast<const f32x4*>(grid_b + (size_t)gy0c * p.GW * p.GD * C)[st_src[sl]];
      sb[sl] = reinterpret_cast<const f32x4*>(grid_b + (size_t)gy1c * p.GW * p.GD * C)[st_src[sl]];
    }
  };
  if constexpr (FUSED) {
    if (nbt > 0) load_grid_rows(floor_to_int(mul_rn(y_first + wave + 0.5f, p.scale_y) - 0.5f));
  }
  {  // zero the A slab once; afterwards every chunk restores the entries it wrote
    f32x4* az = reinterpret_cast<f32x4*>(at);
#pragma unroll
    for (int q = 0; q < (16 * kTStride / 4 + 63) / 64; ++q)
      if (lane + 64 * q < 16 * kTStride / 4) az[lane + 64 * q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // x offsets depend on (chunk-in-row, lane) only: cached for the first kXW chunks of a row (every
  // interval up to 256 px: 4K's 240, 1080p's 120), recomputed per chunk only beyond that
  constexpr int kXW = (WG && WI) ? 1 : 4;  // (all three gradients: registers; recomputed beyond)
  float dxc[kXW];
#pragma unroll
  for (int cb = 0; cb < kXW; ++cb) dxc[cb] = x_offset(x_lo + 64 * cb + lane);
  f32x4 dacc[NH], dacc2[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) dacc[h] = dacc2[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  // y terms of the row being contracted (bilateral_slice_apply.cc:42,47,55-56), formed once per row
  int row_gy0 = 0;
  float row_wy0 = 0.0f, row_wy1 = 0.0f;
  auto process = [&](int t, const Batch& cur, Batch& refill) {
    if constexpr (ABL != 1 && ABL != 4) {  // (tools ablation 1 / 4: the first batch is all a wave ever loads)
      load_batch(refill, t + kAhead < nbt);  // always issued (see load_batch)
    }
    if (t >= nbt) return;  // a padding step of the unrolled ring: it only keeps the VMEM count uniform
    const int y = pr_y, bi = pr_bi;
    if (++pr_bi == nbr) {
      pr_bi = 0;
      pr_y += kWaves;
    }
    const int xb = x_lo + bi * 64 * kBatch;
    if (bi == 0) {
      const float gyf = mul_rn(y + 0.5f, p.scale_y);
      row_gy0 = floor_to_int(gyf - 0.5f);
      row_wy0 = tent_weight(row_gy0 + 0.5f, gyf);
      row_wy1 = tent_weight(row_gy0 + 1 + 0.5f, gyf);
    }
    if constexpr (FUSED) {
      if (bi == 0) {  // new row: blend its coefficient image (wy folded in, planes padded in z)
        const int gy0 = row_gy0;
        const float wy0 = row_wy0, wy1 = row_wy1;
        if (gy0 != gy0_held) load_grid_rows(gy0);  // wave-uniform; at most once more per wave (rg <= cell height)
        // column g + 1 is stored as its difference to column g (element e - GD * C4 holds that element: always one of
        // slot 0, GD * C4 <= 64)
        f32x4 v[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) v[sl] = wy0 * sa[sl] + wy1 * sb[sl];
        {
          f32x4 o[NSLOT];
#pragma unroll
          for (int sl = 0; sl < NSLOT; ++sl) {
            const int partner = max(lane + 64 * sl - p.GD * C4, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[sl][e] = __shfl(v[0][e], partner);
          }
#pragma unroll
          for (int sl = 0; sl < NSLOT; ++sl)
            if (st_col[sl] == 1) v[sl] = v[sl] - o[sl];
        }
        // (round 5) ... and beside every plane its z DIFFERENCE to the plane above (element e + C4 holds plane z + 1 of
        // the same column; the topmost plane's neighbour is its own clamped copy: 0): dguide contracts this difference
        // directly instead of subtracting two contracted taps (below).  Formed from the blended values: its rounding,
        // ~1 ulp of a coefficient, is far below what the contraction of two full taps carried.
        f32x4 dv[NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
          const int ue = lane + 64 * sl + C4;  // the element above; slot ue / 64
          f32x4 upv;
          if constexpr (NSLOT == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) upv[e] = __shfl(v[0][e], min(ue, 63));
          } else if (sl == NSLOT - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) upv[e] = __shfl(v[NSLOT - 1][e], min(ue - 64 * sl, 63));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float same = __shfl(v[sl][e], min(ue - 64 * sl, 63));
              const float next = __shfl(v[NSLOT - 1][e], max(ue - 64 * (sl + 1), 0));
              upv[e] = (ue - 64 * sl < 64) ? same : next;
            }
          }
          dv[sl] = upv - v[sl];
          if (st_z[sl] == p.GD - 1) dv[sl] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
          if (lane + 64 * sl < nst) {
            f32x4* d4 = reinterpret_cast<f32x4*>(img);
            const int dzo = 2 * (p.GD + 2) * C4;  // the difference planes follow the two columns
            d4[st_dst[sl]] = v[sl];
            d4[st_dst[sl] + dzo] = dv[sl];
            if (st_z[sl] == 0) {
              d4[st_dst[sl] - C4] = v[sl];
              d4[st_dst[sl] - C4 + dzo] = f32x4{0.f, 0.f, 0.f, 0.f};  // plane -1 is the clamped copy of plane 0
            }
            if (st_z[sl] == p.GD - 1) d4[st_dst[sl] + C4] = v[sl];
          }
        }
        wave_lds_order();
      }
    }
    const size_t prow_out = ((size_t)b * p.H + y) * p.W;  // wave-uniform
#pragma unroll
    for (int cb = 0; cb < kBatch; ++cb) {
      const int x0 = xb + 64 * cb;
      if (x0 < x_hi) {  // wave-uniform
        [[maybe_unused]] long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
        if constexpr (ABL == 6) ts0 = clock64();
        const int ci = bi * kBatch + cb;  // chunk in row, wave-uniform
        float dx;
        if (ci == 0) dx = dxc[0];
        else if (kXW > 1 && ci == 1) dx = dxc[kXW > 1 ? 1 : 0];
        else if (kXW > 2 && ci == 2) dx = dxc[kXW > 2 ? 2 : 0];
        else if (kXW > 3 && ci == 3) dx = dxc[kXW > 3 ? 3 : 0];
        else dx = x_offset(x0 + lane);  // only intervals wider than 64 kXW px
        // max(1 - |dx|, 0) == clamp(1 - |dx|) to [0, 1] (the difference never exceeds 1): one instruction
        const float w0 = __builtin_amdgcn_fmed3f(1.0f - fabsf(dx), 0.0f, 1.0f);
        const float w1 = __builtin_amdgcn_fmed3f(1.0f - fabsf(dx + 1.0f), 0.0f, 1.0f);
        [[maybe_unused]] const float wb = w1;
        // z: only the two corners around gzf carry weight (:121); the outermost half cells are
        // forced to 1 (:122-125).  Two v_sqrt_f32 per pixel (1 ulp; argument >= 1e-8, no
        // denormals; a weight moves by <= 6e-8, far below the summation noise of a 30 000-term
        // reduction).  P = (za, wa), Q = (za + 1, wb); Q is written first, so where it clamps onto
        // P's slot (za == 7) P's value wins; there wb == 0 anyway.
        const float gzf = mul_rn(cur.g[cb], gd_f);  // gzf = guide * GD  (:120)
        const float fz = floorf(gzf - 0.5f);
        const float dza = (fz + 0.5f) - gzf, dzb = ((fz + 1.0f) + 0.5f) - gzf;  // (float)gz + 0.5f, gz1 = gz0 + 1
        // v_sqrt_f32 (1 ulp; argument >= 1e-8, no denormals).  Both taps lie within one cell of the
        // sample by construction (dza in (-1, 0], dzb in (0, 1]), so dz^2 + eps rounds to <= 1.0f and
        // the reference's `abs_dx > 1 ? 0 : dx / abs_dx` (numerics.h:116-126) takes its zero branch
        // only for wild guides whose f32 offsets round to 2; for every guide with an exact gzf a
        // 1-ulp sqrt cannot flip it (sqrt(1.0f) is exact).  The derivative is dz / s with v_rcp_f32.
        // With dguide fused the smoothed |dz| and its reciprocal (the derivative is dz / s) come from ONE
        // v_rsq_f32 per tap -- s = q * rsq(q), 1.5 ulp instead of v_sqrt_f32's 1 -- instead of a v_sqrt_f32 and a
        // v_rcp_f32 (each a quarter-rate instruction); rsq(1.0f) is exact, so is s there.
        const float qza = fmaf(dza, dza, kSmoothEps), qzb = fmaf(dzb, dzb, kSmoothEps);
        // (round 5: only the upper tap's reciprocal is still needed -- the lower tap's derivative enters dguide through
        //  the cancellation-free sum below)
        [[maybe_unused]] const float rzb = WG ? __builtin_amdgcn_rsqf(qzb) : 0.0f;
        const float sza = __builtin_amdgcn_sqrtf(qza);
        const float szb = WG ? qzb * rzb : __builtin_amdgcn_sqrtf(qzb);
        // U[i] = dout_i * [in; 1]: the rows of V the dgrid contraction stages AND the vectors the fused dguide
        // contracts with, as column pairs (CJ = 4 shapes; two packed multiplies per output channel)
        constexpr bool UPAIRS = APPLY && CJ == 4 && CFULL <= 16;
        [[maybe_unused]] f32x2 U01[COUT], U23[COUT];
        if constexpr (UPAIRS) {
          const f32x2 i01 = {cur.in[cb][0], CIN > 1 ? cur.in[cb][CIN > 1 ? 1 : 0] : 1.0f};
          const f32x2 i23 = {CIN > 2 ? cur.in[cb][CIN > 2 ? 2 : 0] : 1.0f, CIN > 3 ? cur.in[cb][CIN > 3 ? 3 : 0] : 1.0f};
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
            const f32x2 di = {cur.d[cb][i], cur.d[cb][i]};
            U01[i] = i01 * di;
            U23[i] = i23 * di;
          }
        }
        if constexpr (FUSED) {
          // per-pixel VJPs from the row's coefficient image: vectors at (x corner, plane iz + 1 + tap)
          const int iz = (int)__builtin_amdgcn_fmed3f(fz, -1.0f, zhi);
          const int a0 = (int)__umul24((unsigned)(iz + 1), (unsigned)CB);
          // GD * SmoothedLerpWeightGrad (:186-187); the s > 1 branch binds only for wild guides (see above)
          // (the branch is taken on q = s^2: sqrt is monotone and sqrt(1.0f) == 1.0f, so q > 1 <=> the reference's
          //  correctly rounded s > 1 except for q within 2 ulp above 1, which no guide produces -- dz is an exact
          //  difference in [-1, 1] or, for wild guides, +-2 and beyond -- while q * rsq(q) may round to 1 + ulp
          //  for q just BELOW 1, i.e. a guide within 1e-7 of a cell boundary: ~16 pixels of a random 4K frame)
          const float dw1 = (qzb > 1.0f) ? 0.0f : gd_f * (dzb * (WG ? rzb : __builtin_amdgcn_rcpf(szb)));
          // (round 5) dguide = dw0 <G0, U> + dw1 <G1, U> with dw0 ~ -GD, dw1 ~ +GD and dot products up to ~10: terms of
          // ~80 cancel, and both the 1.5-ulp error of dw and the rounding of the two dot products survive the
          // cancellation (the reference's own float32 evaluation carries the same noise, 1.5e-5 on the suite's data;
          // this kernel's was 1.2-1.9 x that).  Evaluated instead as  dw1 <G1 - G0, U> + (dw0 + dw1) <G0, U>:
          //   * G1 - G0 comes from the image's z-difference planes (differences of the raw grid values): the
          //     cancellation happens per coefficient, before the contraction;
          //   * dw0 + dw1 = GD (dza / sza + dzb / szb) without subtracting two numbers near GD: with s^2 = dz^2 + eps,
          //     dz / s = sign(dz) (1 - e), e = eps / (s (s + |dz|)), so the sum is GD (e_a - e_b)  (dza <= 0 < dzb).
          [[maybe_unused]] float dwsum = 0.0f;
          if constexpr (WG) {
            // e_a - e_b = eps (D_b - D_a) / (D_a D_b), D = s (s + |dz|): one reciprocal for both taps
            const float Da = sza * (sza + fabsf(dza)), Db = szb * (szb + fabsf(dzb));
            dwsum = (gd_f * kSmoothEps) * ((Db - Da) * __builtin_amdgcn_rcpf(Da * Db));
            if (__builtin_expect(__ballot(qza > 1.0f || qzb > 1.0f) != 0ull, 0)) {  // wave-uniform; wild guides only:
              // a tap past its cell has derivative 0 (numerics.h:116-126), the sum is the direct one
              const float dw0 = (qza > 1.0f) ? 0.0f : gd_f * (dza * __builtin_amdgcn_rcpf(sza));
              if (qza > 1.0f || qzb > 1.0f) dwsum = dw0 + dw1;
            }
          }
          // Direct form (no 2 x C blended-coefficient accumulators: 4 scalars instead of 24 registers
          // live, which is what keeps this kernel at 4 waves per SIMD).  With U[c] = dout_i * [in; 1]_j
          // -- the SAME per-pixel products the dgrid contraction uses -- and G_v the coefficient
          // vector of corner v = (x corner, z tap):
          //   dguide   = sum_v (wx dw)_v <G_v, U>                 (bilateral_slice_apply.cc:140-206)
          //   dinput_j = sum_v (wx wz)_v sum_i dout_i G_v[i, j]   (:208-259)
          float dgv = 0.0f;
          f32x2 div01 = {0.0f, 0.0f}, div23 = {0.0f, 0.0f};  // dinput columns (0, 1), (2, 3)
          {
            // The image holds column g and the DIFFERENCE to column g + 1, so the x blend of a z tap is one FMA
            // per coefficient -- G_t = A_t + w1 (B_t - A_t); the two x weights sum to 1 up to rounding (w0 =
            // 1 - |dx|, w1 = 1 - |dx + 1|, dx in (-1, 0]; bilateral_slice_apply.cc:58-68), so this moves a
            // coefficient by <= 1 ulp of max(|A|, |B|).  Then, with U[c] = dout_i * [in; 1]_j -- the SAME
            // per-pixel products the dgrid contraction uses -- per tap t:
            //   dguide   += dw_t <G_t, U>                 (bilateral_slice_apply.cc:140-206)
            //   dinput_j += wz_t sum_i dout_i G_t[i, j]   (:208-259)
            // 2 x (C + C + 3 CIN) FMAs per pixel instead of the four-corner form's 4 x (C + 3 CIN).
            const float wz0 = __builtin_amdgcn_fmed3f(1.0f - sza, 0.0f, 1.0f);  // max(1 - s, 0): s > 0
            const float wz1 = __builtin_amdgcn_fmed3f(1.0f - szb, 0.0f, 1.0f);
            // "tap" 0 = the lower plane's vector G0, "tap" 1 = the z difference G1 - G0:
            //   dguide = (dw0 + dw1) <G0, U> + dw1 <G1 - G0, U>;  dinput = (wz0 + wz1) T(G0) + wz1 T(G1 - G0)
            const float wzt[2] = {wz0 + wz1, wz1};
            const float dwt[2] = {dwsum, dw1};
            const f32x4 wx1 = {wb, wb, wb, wb};
            // One float4 of a tap at a time (row i of [COUT][CJ = 4] for APPLY, channels 4 q .. 4 q + 3 for a
            // slice), the next one's two reads issued under this one's math: a whole tap in registers (A and the
            // difference: 2 C floats) costs the fourth wave per SIMD.  The z-difference planes ("tap" 1) lie two
            // columns behind the planes themselves.
            constexpr int NQ = C / 4, NS = 2 * NQ;
            const char* ibase = reinterpret_cast<const char*>(img) + a0;
            const int dzb_off = 2 * colb - NQ * 16;  // vector st >= NQ: (st - NQ) * 16 + 2 * colb
            f32x4 nA = *reinterpret_cast<const f32x4*>(ibase), nD = *reinterpret_cast<const f32x4*>(ibase + colb);
            f32x2 acc = {0.0f, 0.0f}, t01 = {0.0f, 0.0f}, t23 = {0.0f, 0.0f};
#pragma unroll
            for (int st = 0; st < NS; ++st) {
              const int t = st / NQ, q = st % NQ;
              const f32x4 G = __builtin_elementwise_fma(wx1, nD, nA);
              if (st + 1 < NS) {
                const int o = (st + 1) * 16 + ((st + 1) >= NQ ? dzb_off : 0);
                nA = *reinterpret_cast<const f32x4*>(ibase + o);
                nD = *reinterpret_cast<const f32x4*>(ibase + colb + o);
              }
              if (q == 0) acc = t01 = t23 = f32x2{0.0f, 0.0f};
              if constexpr (WG) {
                if constexpr (APPLY) {  // <G[i, :], dout_i * [in; 1]> two columns at a time
                  static_assert(CJ == 4, "fused apply shapes have 4 grid columns per output channel");
                  acc = __builtin_elementwise_fma(f32x2{G.x, G.y}, U01[q], acc);
                  acc = __builtin_elementwise_fma(f32x2{G.z, G.w}, U23[q], acc);
                } else {
                  acc = __builtin_elementwise_fma(f32x2{G.x, G.y}, f32x2{cur.d[cb][4 * q], cur.d[cb][4 * q + 1]}, acc);
                  acc = __builtin_elementwise_fma(f32x2{G.z, G.w}, f32x2{cur.d[cb][4 * q + 2], cur.d[cb][4 * q + 3]}, acc);
                }
                if (q == NQ - 1) dgv = fmaf(dwt[t], acc.x + acc.y, dgv);
              }
              if constexpr (WI) {
                // t_j = sum_i G[i, j] dout_i: columns (0, 1) packed, column 2 (and 3 for CIN = 4) beside them
                const f32x2 di = {cur.d[cb][q], cur.d[cb][q]};
                t01 = __builtin_elementwise_fma(f32x2{G.x, G.y}, di, t01);
                if constexpr (CIN > 3) t23 = __builtin_elementwise_fma(f32x2{G.z, G.w}, di, t23);
                else if constexpr (CIN > 2) t23.x = fmaf(G.z, cur.d[cb][q], t23.x);
                if (q == NQ - 1) {
                  const f32x2 wv = {wzt[t], wzt[t]};
                  div01 = __builtin_elementwise_fma(wv, t01, div01);
                  if constexpr (CIN > 3) div23 = __builtin_elementwise_fma(wv, t23, div23);
                  else if constexpr (CIN > 2) div23.x = fmaf(wzt[t], t23.x, div23.x);
                }
              }
              __builtin_amdgcn_sched_barrier(0);  // keeps the read-ahead at one vector
            }
          }
          {  // nontemporal buffer stores; descriptors end at the interval, so dead lanes are dropped
            const unsigned px = (unsigned)(x0 + lane);
            if constexpr (WG) {
              const __amdgpu_buffer_rsrc_t rs = rows::make_rsrc_uniform(p.dguide + prow_out, (unsigned)x_hi * 4u);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dgv), rs, px * 4u, 0, rows::kAuxStream);
            }
            if constexpr (WI) {
              const __amdgpu_buffer_rsrc_t rs = rows::make_rsrc_uniform(p.dinput + prow_out * CIN, (unsigned)x_hi * (4u * CIN));
              if constexpr (CIN == 3) {
                const u32x3 v = {__float_as_uint(div01.x), __float_as_uint(div01.y), __float_as_uint(div23.x)};
                __builtin_amdgcn_raw_buffer_store_b96(v, rs, __umul24(px, 12u), 0, rows::kAuxStream);
              } else if constexpr (CIN == 4) {
                const u32x4 v = {__float_as_uint(div01.x), __float_as_uint(div01.y), __float_as_uint(div23.x), __float_as_uint(div23.y)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, px * 16u, 0, rows::kAuxStream);
              } else {
                const float dv[4] = {div01.x, div01.y, div23.x, div23.y};
#pragma unroll
                for (int j = 0; j < CIN; ++j)
                  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv[j]), rs, (__umul24(px, CIN) + j) * 4u, 0, rows::kAuxStream);
              }
            }
          }
        }
        // P = (plane of the lower tap, its weight), Q = the upper tap.  In the outermost half cells
        // (gzf < .5, gzf > GD - .5; :121-125) the one live plane is forced to weight 1: there the lower tap's
        // clamped plane IS that plane (fz = -1 -> 0; fz >= GD - 1 -> GD - 1) and Q carries weight 0 -- where Q
        // clamps onto P's slot it is written first, so P's value wins.
        const bool edge = __builtin_amdgcn_fmed3f(gzf, 0.5f, gd_f - 0.5f) != gzf;
        const float wP = edge ? 1.0f : __builtin_amdgcn_fmed3f(1.0f - sza, 0.0f, 1.0f);
        const float wQ = edge ? 0.0f : __builtin_amdgcn_fmed3f(1.0f - szb, 0.0f, 1.0f);
        const int zP = (int)__builtin_amdgcn_fmed3f(fz, 0.0f, zhi), zQ = min(zP + 1, p.GD - 1);
        // (NH = 2: plane z lives in row z & 7 of half z >> 3's tile; the scatter itself happens per live half, below)
        float* aP = at + __umul24((unsigned)(NH == 1 ? zP : (zP & 7)), (unsigned)kTStride) + lane;
        float* aQ = at + __umul24((unsigned)(NH == 1 ? zQ : (zQ & 7)), (unsigned)kTStride) + lane;
        auto enc2 = [](f32x2 v) { return SPLIT == 1 ? f32x2(split_pack2(v)) : SPLIT == 2 ? f32x2(split_pack2_f16(v)) : v; };
        auto enc = [&](float v) { return SPLIT ? enc2(f32x2{v, v}).x : v; };
        const f32x2 q2 = enc2(f32x2{w0, w1} * f32x2{wQ, wQ}), p2 = enc2(f32x2{w0, w1} * f32x2{wP, wP});
        if constexpr (NH == 1) {
          aQ[0] = q2.x;
          aQ[8 * kTStride] = q2.y;
          aP[0] = p2.x;
          aP[8 * kTStride] = p2.y;
        }
        // V^T[c][px]: dout x [in; 1] (slice: dout)
        if constexpr (UPAIRS) {
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
            const f32x2 e01 = enc2(U01[i]), e23 = enc2(U23[i]);
            vt[(i * CJ + 0) * kTStride + lane] = e01.x;
            vt[(i * CJ + 1) * kTStride + lane] = e01.y;
            vt[(i * CJ + 2) * kTStride + lane] = e23.x;
            vt[(i * CJ + 3) * kTStride + lane] = e23.y;
          }
        } else if constexpr (APPLY) {
#pragma unroll
          for (int i = 0; i < COUT; ++i) {
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
              const int c = i * CJ + j;  // (compile-time after unrolling: only this window's channels are staged)
              if (c >= C0 && c < C0 + C)
                vt[(c - C0) * kTStride + lane] =
                    enc((j < CIN) ? cur.d[cb][i] * cur.in[cb][j < CIN ? j : 0] : cur.d[cb][i]);
            }
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) vt[c * kTStride + lane] = enc(cur.d[cb][c]);
        }
        wave_lds_order();
        // D[k, c] += sum_px A[k, px] * V[px, c]; two accumulators break the dependent-issue chain.
        // Two half-chunks: 8 operand reads (A, V pairs in MFMA order) -> 8 MFMAs, twice, through the SAME 16
        // registers (the second half's reads are tied to the first half's variables, so they issue after the
        // first half's MFMAs have read them).
        f32x2 av[4], bv[4];
#define HDRNET_GG_READS(OFF, TIE)                                                                          \
  asm volatile("ds_read_b64 %0, %8 offset:" #OFF "+0\n\tds_read_b64 %4, %9 offset:" #OFF "+0\n\t"           \
               "ds_read_b64 %1, %8 offset:" #OFF "+16\n\tds_read_b64 %5, %9 offset:" #OFF "+16\n\t"         \
               "ds_read_b64 %2, %8 offset:" #OFF "+32\n\tds_read_b64 %6, %9 offset:" #OFF "+32\n\t"         \
               "ds_read_b64 %3, %8 offset:" #OFF "+48\n\tds_read_b64 %7, %9 offset:" #OFF "+48\n\t"         \
               "s_waitcnt lgkmcnt(0)"                                                                      \
               : TIE(av[0]), TIE(av[1]), TIE(av[2]), TIE(av[3]), TIE(bv[0]), TIE(bv[1]), TIE(bv[2]),       \
                 TIE(bv[3])                                                                                \
               : "v"(a_addr), "v"(v_addr)                                                                  \
               : "memory")
#define HDRNET_GG_OUT(x) "=&v"(x)
#define HDRNET_GG_INOUT(x) "+v"(x)
        auto contract = [&](f32x4& dacc, f32x4& dacc2) {
          if constexpr (SPLIT != 0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
              const f32x4 a4 = {av[2 * q][0], av[2 * q][1], av[2 * q + 1][0], av[2 * q + 1][1]};
              const f32x4 b4 = {bv[2 * q][0], bv[2 * q][1], bv[2 * q + 1][0], bv[2 * q + 1][1]};
              const u32x4_t vb = __builtin_bit_cast(u32x4_t, b4);
              // [a_hi, a_lo] x [v_hi, v_lo] = hi hi + lo lo, then x the ROTATED dword [v_lo, v_hi] = the two cross terms:
              // all four products, one v_alignbit per B dword (round 2: three products, a perm and a shift per dword)
              u32x4_t br;
#pragma unroll
              for (int e = 0; e < 4; ++e) br[e] = __builtin_amdgcn_alignbit(vb[e], vb[e], 16);
              if constexpr (SPLIT == 1) {
                const bf16x8 a8 = __builtin_bit_cast(bf16x8, a4);
                dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, vb), dacc, 0, 0, 0);
                dacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, br), dacc2, 0, 0, 0);
              } else {
                // f16 hi / lo': dacc takes hi hi alone ([a_hi, 0]: the lo' lo' product has the wrong scale), dacc2 the
                // cross terms (scale 2^-11, applied at the row fold)
                const u32x4_t ua = __builtin_bit_cast(u32x4_t, a4);
                const u32x4_t ah = {ua[0] & 0xffffu, ua[1] & 0xffffu, ua[2] & 0xffffu, ua[3] & 0xffffu};
                dacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, vb), dacc, 0, 0, 0);
                dacc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, br), dacc2, 0, 0, 0);
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (ABL == 2 || ABL == 4) {  // tools ablation: no MFMAs
                dacc[0] += av[e][0] * bv[e][0];
                dacc2[0] += av[e][1] * bv[e][1];
                continue;
              }
              dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e][0], bv[e][0], dacc, 0, 0, 0);
              dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e][1], bv[e][1], dacc2, 0, 0, 0);
            }
          }
        };
        if constexpr (ABL == 6) ts1 = clock64();  // VALU phase + staging writes issued
        if constexpr (NH == 1) {
          HDRNET_GG_READS(0, HDRNET_GG_OUT);
          if constexpr (ABL == 6) ts2 = clock64();  // first half's operands have arrived
          contract(dacc[0], dacc2[0]);
          HDRNET_GG_READS(64, HDRNET_GG_INOUT);
          if constexpr (ABL == 6) ts3 = clock64();  // first half's MFMAs issued, second half's operands arrived
          contract(dacc[0], dacc2[0]);
          wave_lds_order();
          aQ[0] = 0.0f;
          aQ[8 * kTStride] = 0.0f;
          aP[0] = 0.0f;
          aP[8 * kTStride] = 0.0f;
        } else {
          // which plane halves hold a tap of this chunk (zQ >= zP: half 0 is live iff some zP < 8, half 1 iff some
          // zQ >= 8); a pixel's tap outside the half being contracted is staged as 0 (a zero written into a zeroed
          // slab: rows z & 7 of the two taps differ, or Q's slot is P's and P is written last)
          const bool live_h[2] = {__ballot(zP < 8) != 0ull, __ballot(zQ >= 8) != 0ull};
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (live_h[h]) {  // wave-uniform
              const bool inP = (zP >> 3) == h, inQ = (zQ >> 3) == h;
              aQ[0] = inQ ? q2.x : 0.0f;
              aQ[8 * kTStride] = inQ ? q2.y : 0.0f;
              aP[0] = inP ? p2.x : 0.0f;
              aP[8 * kTStride] = inP ? p2.y : 0.0f;
              wave_lds_order();
              HDRNET_GG_READS(0, HDRNET_GG_OUT);
              contract(dacc[h], dacc2[h]);
              HDRNET_GG_READS(64, HDRNET_GG_INOUT);
              contract(dacc[h], dacc2[h]);
              wave_lds_order();
              aQ[0] = 0.0f;
              aQ[8 * kTStride] = 0.0f;
              aP[0] = 0.0f;
              aP[8 * kTStride] = 0.0f;
            }
          }
        }
#undef HDRNET_GG_READS
#undef HDRNET_GG_OUT
#undef HDRNET_GG_INOUT
        if constexpr (ABL == 6) {
          const int nch = t * kBatch + cb;  // chunk ordinal of this wave
          if (wave == 0 && lane == 0 && nch < kTraceChunks) {
            long long* tr = p.trace + ((size_t)task * kTraceChunks + nch) * 5;
            tr[0] = ts0; tr[1] = ts1; tr[2] = ts2; tr[3] = ts3; tr[4] = clock64();
          }
        }
        if constexpr (WI) __builtin_amdgcn_sched_barrier(0);  // chunk by chunk: keeps the live set of one
      }
    }
    if (bi == nbr - 1) {
      // last batch of the row: fold the row's 16x16 result, scaled by its two y weights
      // (bilateral_slice_apply.cc:42,47,55-56; weights un-clamped, indices clamped), into the
      // register tiles of the (<= 3) grid rows the group touches.
      const int gy0 = row_gy0;
      const float wy0 = row_wy0, wy1 = row_wy1;
      const int rel0 = clamp_index(gy0, 0, p.GH - 1) - gy_base;
      const int rel1 = clamp_index(gy0 + 1, 0, p.GH - 1) - gy_base;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        if constexpr (SPLIT == 2) dacc[h] += 0x1p-11f * dacc2[h];  // hi hi + 2^-11 (hi lo' + lo' hi)
        else dacc[h] += dacc2[h];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const float sr = (rel0 == rr ? wy0 : 0.0f) + (rel1 == rr ? wy1 : 0.0f);
          acc[h][rr] += sr * dacc[h];
        }
        dacc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        dacc2[h] = dacc[h];
      }
    }
  };
  if constexpr (ABL == 1 || ABL == 4) {
    for (int t = 0; t < nbt; ++t) process(t, ring[0], ring[0]);
  } else if constexpr (kAhead == 1) {
    for (int t = 0; t < nbt; t += 2) {
      process(t, ring[0], ring[1]);
      process(t + 1, ring[1], ring[0]);
    }
  } else {
    for (int t = 0; t < nbt; t += 3) {
      process(t, ring[0], ring[2]);
      process(t + 1, ring[1], ring[0]);
      process(t + 2, ring[2], ring[1]);
    }
  }
  // Sum the four waves' register tiles in fixed order (wave 0 + 1 + 2 + 3) through LDS -- the
  // operand slabs are free now -- and write one partial tile per workgroup.
  __syncthreads();
  float* red = lds + wave * kSlab;
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int q = 0; q < 4; ++q) red[((h * 3 + r) * 16 + 4 * sub + q) * 16 + bc] = acc[h][r][q];
    }
  }
  __syncthreads();
  float* dst = p.partial + (size_t)task * (NH * kTileFloats);  // [half][rel 3][k 16][c 16]
  for (int e = threadIdx.x; e < NH * kTileFloats; e += kWaves * 64) {
    float sum = lds[e];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += lds[w * kSlab + e];
    dst[e] = sum;
  }
}

// Stage 2.  One 512-thread workgroup per grid column (b, gy, gx) -- a 3-D launch grid, no index divisions:
// lane (part, z, c) adds the partial tiles of row groups yg = yg_lo + part, + 4, ... for plane z, channel c --
// for each group the interval g = gx (its x-corner-0 rows) and the interval g = gx - 1 (its x-corner-1 rows) --
// then the 4 partial sums are added in fixed order.  A wave reads 256-B runs (4 planes x 16 channels of one tile);
// the result is deterministic.  (Round 2: one 256-thread workgroup per (column, plane), 16 parts: 5.6 us of device
// time per call at 4K, profiles/r03/bwd_kernel_stats.csv.)
// NH = 2 (9 .. 16 planes): 16 plane slots x 2 parts; a task's tile is [half][rel][k][c], plane z in half z >> 3, row z & 7.
template <int NH>
__global__ __launch_bounds__(512) void grid_grad_stage2(const float* __restrict__ partial,
                                                        float* __restrict__ dgrid, int GH, int GW,
                                                        int GD, int C, int rg, int nyg,
                                                        float scale_y, int c0 = 0, int Cw = 16) {
  constexpr int kNZ = 8 * NH, kS2Parts = 32 / kNZ, kTile = NH * kTileFloats;
  __shared__ float red[kS2Parts][kNZ * 16 + 1];
  const int c = threadIdx.x & 15, z = (threadIdx.x >> 4) & (kNZ - 1), part = threadIdx.x / (16 * kNZ);
  const int gx = blockIdx.x, gy = blockIdx.y;
  const long long b = blockIdx.z;
  const int nint = GW + 1;
  const int zrow = (z >> 3) * 3 * 16 + (z & 7);  // + rel * 16: row of plane z, x corner 0
  // Conservative window of row groups that can touch gy; exact membership is `rel`.
  const int yg_lo = max(0, (int)floorf((gy - 2.0f) / scale_y) / rg - 1);
  const int yg_hi = min(nyg, (int)ceilf((gy + 2.5f) / scale_y) / rg + 2);
  float s = 0.0f;
  if (z < GD) {
    for (int yg = yg_lo + part; yg < yg_hi; yg += kS2Parts) {
      const int rel = gy - gy_base_of(yg * rg, scale_y, GH);
      if (rel < 0 || rel > 2) continue;
      const size_t t0 = ((size_t)b * nyg + yg) * nint;
      float v0 = partial[(t0 + gx + 1) * kTile + (rel * 16 + zrow) * 16 + c];
      float v1 = partial[(t0 + gx) * kTile + (rel * 16 + 8 + zrow) * 16 + c];
      // clamp-to-edge: interval g = -1's corner 0 and interval g = GW - 1's corner 1 land on the edge columns
      float v2 = 0.0f, v3 = 0.0f;
      if (gx == 0) v2 = partial[t0 * kTile + (rel * 16 + zrow) * 16 + c];
      if (gx == GW - 1) v3 = partial[(t0 + GW) * kTile + (rel * 16 + 8 + zrow) * 16 + c];
      s += v0;
      s += v1;
      if (gx == 0) s += v2;
      if (gx == GW - 1) s += v3;
    }
  }
  red[part][z * 16 + c] = s;
  __syncthreads();
  if (part == 0 && z < GD && c < Cw && c0 + c < C) {  // (C: the grid's channels; this launch's window: c0 .. c0 + Cw - 1)
    float t = red[0][z * 16 + c];
#pragma unroll
    for (int q = 1; q < kS2Parts; ++q) t += red[q][z * 16 + c];
    dgrid[((((size_t)b * GH + gy) * GW + gx) * GD + z) * C + c0 + c] = t;
  }
}

struct GGPlan {
  int rg, nyg;
  long long ntasks;
  size_t ws_bytes;
};

// Rows per workgroup task (4 waves take alternate rows).  The grid's fit to the machine is a first-order
// effect: a wave's start-up (prologue + the one pixel load nothing hides) is paid per task, and a last,
// partly filled round of workgroups runs the chip below its occupancy.  At 4K, dgrid + dguide: 16 rows
// per task = 2295 tasks over 1024 resident workgroups (2.24 rounds) 99.5 us; 36 rows = 1020 tasks, one
// exact round, 90.9 us; 34 rows = 1088 tasks, one round and a sliver, 109 us (profiles/r02/exp27).  So
// rg is chosen per launch from the kernel's resident-workgroup count (`slots`): the number of rounds
// R <= 6 whose exact-fit rg fills them best, rg within [4, cell height] (a task's rows may span at most 3
// clamped grid rows).  slots <= 0: the smallest rg any launch may pick -- the workspace bound.
constexpr int kMinRg = 4, kMaxRounds = 6;
constexpr int kMaxOcc = 8;  // workgroups per CU a stage-1 kernel can have (8 waves per SIMD, 4-wave workgroups)

bool gg_plan(int B, int H, int W, int GH, int GW, int GD, int C, long long slots, GGPlan* pl) {
  if (GD > 16 || C > 32 || C < 1) return false;
  const int nh = GD > 8 ? 2 : 1;  // plane halves: a task writes one [3][16][16] tile per half
  const int nw = C > 16 ? 2 : 1;  // channel windows: one stage-1 launch and one set of partial tiles per window
  const int cell = H / GH > 1 ? H / GH : 1;
  const int rg_lo = cell < kMinRg ? cell : kMinRg;
  const long long cols = (long long)B * (GW + 1);
  int rg = rg_lo;
  if (slots > 0) {
    double best = -1.0;
    for (int R = 1; R <= kMaxRounds; ++R) {
      const long long per_col = (slots * R) / cols;  // row groups a column of tasks may have
      if (per_col < 1) continue;
      int cand = (int)((H + per_col - 1) / per_col);
      if (cand > cell) continue;  // would need more rows per task than a cell is high
      if (cand < rg_lo) cand = rg_lo;
      const long long ntasks = cols * ((H + cand - 1) / cand);
      const long long rounds = (ntasks + slots - 1) / slots;
      const double eff = (double)ntasks / (double)(rounds * slots);
      if (eff > best + 1e-9) {
        best = eff;
        rg = cand;
      }
    }
    if (best < 0.0) rg = cell;  // fewer tasks than slots even at the largest rg a cell allows
  }
#ifdef HDRNET_TOOLS_BUILD
  if (const char* e = getenv("HDRNET_GG_RG")) rg = atoi(e) < cell ? atoi(e) : cell;  // experiments only
#endif
  if (rg < 1) rg = 1;
  pl->rg = rg;
  pl->nyg = (H + rg - 1) / rg;
  pl->ntasks = cols * pl->nyg;
  if (pl->ntasks > 0x7fffffffLL || (long long)B * GH * GW * GD > 0x7fffffffLL || pl->nyg > 65535 || B > 65535 || GH > 65535) return false;
  pl->ws_bytes = (size_t)pl->ntasks * nh * nw * kTileFloats * sizeof(float);
  return true;
}

// Workspace bound for a shape, whatever gradients the call will ask for: the launch plans its rows per task
// from the resident-workgroup count of the kernel variant it picks (occupancy x CUs of the current device);
// a variant's occupancy is 1 .. 8 workgroups per CU (8 waves per SIMD, 4-wave workgroups), so the largest
// plan over those eight counts bounds every launch.  (Round 2 returned the plan for the smallest rows-per-
// task any launch may pick -- 28 MB at 4K where a launch uses 7, hundreds of MB for batched training.)
bool gg_ws_bound(int B, int H, int W, int GH, int GW, int GD, int C, size_t* bytes) {
  // asked for on every backward call (the *_supported checks): the last answer per thread is kept, keyed on the
  // shape and the CU count it was planned for
  struct Key {
    int B, H, W, GH, GW, GD, C;
    long long cus;
    bool operator==(const Key& o) const {
      return B == o.B && H == o.H && W == o.W && GH == o.GH && GW == o.GW && GD == o.GD && C == o.C && cus == o.cus;
    }
  };
  thread_local Key last{};
  thread_local size_t last_bytes = 0;
  thread_local bool last_ok = false, have = false;
  const long long cus = rows::num_cus();
  const Key k{B, H, W, GH, GW, GD, C, cus};
  if (!have || !(last == k)) {
    size_t worst = 0;
    bool ok = true;
    for (int occ = 1; occ <= kMaxOcc && ok; ++occ) {
      GGPlan pl;
      ok = gg_plan(B, H, W, GH, GW, GD, C, occ * cus, &pl);
      if (ok && pl.ws_bytes > worst) worst = pl.ws_bytes;
    }
    last = k;
    last_bytes = worst;
    last_ok = ok;
    have = true;
  }
  *bytes = last_bytes;
  return last_ok;
}

// The caller's workspace is smaller than this device's bound (queried on another device, or with an older
// library): the call then runs on the generic gather kernel, ~100x slower.  Say so once instead of silently.
void warn_small_workspace(size_t have, size_t need) {
  static std::atomic<bool> said{false};
  if (have == 0 || said.exchange(true)) return;
  fprintf(stderr, "hdrnet_amd: gradient workspace of %zu bytes is smaller than the %zu this device needs "
          "(hdrnet_bilateral_slice*_grad_workspace_bytes, queried with the device current); using the generic "
          "grid-gradient kernel, which is ~100x slower\n", have, need);
}

long long* g_gg_trace = nullptr;  // tools: phase-trace buffer (grid_grad_set_trace)

// Workgroups of `kfn` resident on the device at once (occupancy x CUs).  Queried once per kernel.
typedef void (*Stage1Fn)(GGParams);

long long resident_slots(Stage1Fn kfn, std::atomic<int>* cache, size_t dyn_lds = 0) {
  if (dyn_lds) cache = nullptr;  // (tools: an occupancy experiment -- never cached)
  int occ = cache ? cache->load(std::memory_order_relaxed) : 0;
  if (occ <= 0) {
    occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kfn), kWaves * 64, dyn_lds) !=
            hipSuccess || occ <= 0)
      occ = 1;
    if (occ > kMaxOcc) occ = kMaxOcc;  // the workspace bound (gg_ws_bound) covers 1 .. kMaxOcc
    if (cache) cache->store(occ, std::memory_order_relaxed);
  }
  return (long long)occ * rows::num_cus();  // CU count of the CURRENT device (cached per ordinal)
}

struct GGPtrs {
  const float *guide, *input, *dout, *grid;
  float *dgrid, *dguide, *dinput;
};

template <int CIN, int COUT, bool OFFSET, bool APPLY>
hipError_t gg_launch(const GGPtrs& q, int B, int H, int W, int GH, int GW, int GD, void* ws, size_t ws_bytes,
                     hipStream_t s, int split, int ablate = 0) {
  constexpr int C = APPLY ? COUT * (CIN + (OFFSET ? 1 : 0)) : COUT;
  const bool wg = q.dguide != nullptr, wi = q.dinput != nullptr;
  Stage1Fn kfn = nullptr;
  std::atomic<int>* occ = nullptr;
  static std::atomic<int> occ_cache[16];  // per (plane halves, split, dguide, dinput) of this shape
#ifdef HDRNET_TOOLS_BUILD
  if constexpr (APPLY && CIN == 3 && COUT == 3 && OFFSET) {  // ablations (tools variants 4 .. 8): timing only
    if (ablate >= 1 && ablate <= 6) {
      if (ablate == 6 && !g_gg_trace) return hipErrorInvalidValue;
#define GG_ABL(A)                                                                              \
  (wg && wi ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0, true, true, A>       \
            : wg ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0, true, false, A> \
                 : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0, false, false, A>)
      kfn = ablate == 1 ? GG_ABL(1) : ablate == 2 ? GG_ABL(2) : ablate == 3 ? GG_ABL(3) : ablate == 4 ? GG_ABL(4) : ablate == 5 ? GG_ABL(5) : GG_ABL(6);
#undef GG_ABL
    }
  }
#endif
  const bool two = GD > 8;  // NH = 2: planes 8 .. 15 in a second tile per task
  if (two && (split || kfn)) return hipErrorNotSupported;  // the tools variants exist for GD <= 8 only
  if (!kfn) {
    occ = &occ_cache[(two ? 12 : 4 * split) + (wg ? 2 : 0) + (wi ? 1 : 0)];
    if constexpr (C % 4 == 0) {
      constexpr bool CAN_WI = APPLY && CIN > 0;
      if (wi && !CAN_WI) return hipErrorInvalidValue;
#define GG_PICK(SPL, NH)                                                                              \
  (wg && wi ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, SPL, true, CAN_WI, 0, NH>          \
            : wg ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, SPL, true, false, 0, NH>      \
                 : wi ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, SPL, false, CAN_WI, 0, NH> \
                      : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, SPL, false, false, 0, NH>)
#ifdef HDRNET_TOOLS_BUILD  // the split contractions are experiments: not in the product library
      kfn = two ? GG_PICK(0, 2) : split == 1 ? GG_PICK(1, 1) : split == 2 ? GG_PICK(2, 1) : GG_PICK(0, 1);
#else
      if (split) return hipErrorNotSupported;
      kfn = two ? GG_PICK(0, 2) : GG_PICK(0, 1);
#endif
#undef GG_PICK
    } else {
      if (wg || wi) return hipErrorInvalidValue;  // fused VJPs read the coefficient image as float4
#ifdef HDRNET_TOOLS_BUILD
      kfn = two ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0, false, false, 0, 2>
                : split == 1 ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 1>
                : split == 2 ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 2>
                             : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0>;
#else
      if (split) return hipErrorNotSupported;
      kfn = two ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0, false, false, 0, 2>
                : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, APPLY, 0>;
#endif
    }
  }
#ifdef HDRNET_TOOLS_BUILD
  // tools knob 1: bytes of UNUSED dynamic LDS per workgroup -- lowers the resident workgroups per CU (and re-fits the
  // row groups to the new round size) to measure what a resident wave is worth (profiles/r05/bwd_steps.md)
  const size_t dyn_lds = (size_t)(tools_knob(1) > 0 ? tools_knob(1) : 0);
#else
  const size_t dyn_lds = 0;
#endif
  GGPlan pl;
  if (!gg_plan(B, H, W, GH, GW, GD, C, resident_slots(kfn, occ, dyn_lds), &pl) || pl.ws_bytes > ws_bytes)
    return hipErrorInvalidValue;
  GGParams p{q.guide, q.input, q.dout, q.grid, q.dguide, q.dinput, static_cast<float*>(ws), H, W, GH, GW, GD,
             pl.rg, pl.nyg, pl.ntasks, (float)GW / W, (float)GH / H, g_gg_trace};
  const dim3 nblocks((unsigned)(GW + 1), (unsigned)pl.nyg, (unsigned)B);
  kfn<<<nblocks, kWaves * 64, dyn_lds, s>>>(p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const dim3 ncols((unsigned)GW, (unsigned)GH, (unsigned)B);
  if (two)
    grid_grad_stage2<2><<<ncols, 512, 0, s>>>(static_cast<const float*>(ws), q.dgrid, GH, GW, GD, C, pl.rg, pl.nyg,
                                              (float)GH / H);
  else
    grid_grad_stage2<1><<<ncols, 512, 0, s>>>(static_cast<const float*>(ws), q.dgrid, GH, GW, GD, C, pl.rg, pl.nyg,
                                              (float)GH / H);
  return hipGetLastError();
}

// 17 .. 32 grid channels (4 -> 4 with offset): dgrid as two channel windows of 16 columns (grid_grad_stage1's CW) --
// stage 1 and stage 2 once per window, the second window's partial tiles behind the first's in the workspace.
template <int CIN, int COUT, bool OFFSET>
hipError_t gg_launch_windows(const GGPtrs& q, int B, int H, int W, int GH, int GW, int GD, void* ws, size_t ws_bytes,
                             hipStream_t s) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  static_assert(C > 16 && C <= 32, "two windows");
  if (q.dguide || q.dinput) return hipErrorInvalidValue;  // the per-pixel VJPs of these shapes run in apply_vjp_seg
  const bool two = GD > 8;
  static std::atomic<int> occ_cache[2];
  const Stage1Fn k0 = two ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, true, 0, false, false, 0, 2, 0>
                          : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, true, 0, false, false, 0, 1, 0>;
  const Stage1Fn k1 = two ? (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, true, 0, false, false, 0, 2, 1>
                          : (Stage1Fn)grid_grad_stage1<CIN, COUT, OFFSET, true, 0, false, false, 0, 1, 1>;
  GGPlan pl;
  if (!gg_plan(B, H, W, GH, GW, GD, C, resident_slots(k0, &occ_cache[two ? 1 : 0]), &pl) || pl.ws_bytes > ws_bytes)
    return hipErrorInvalidValue;
  const size_t window_floats = (size_t)pl.ntasks * (two ? 2 : 1) * kTileFloats;
  const dim3 nblocks((unsigned)(GW + 1), (unsigned)pl.nyg, (unsigned)B);
  const dim3 ncols((unsigned)GW, (unsigned)GH, (unsigned)B);
  for (int w = 0; w < 2; ++w) {
    float* part = static_cast<float*>(ws) + w * window_floats;
    GGParams p{q.guide, q.input, q.dout, q.grid, nullptr, nullptr, part, H, W, GH, GW, GD,
               pl.rg, pl.nyg, pl.ntasks, (float)GW / W, (float)GH / H, nullptr};
    (w == 0 ? k0 : k1)<<<nblocks, kWaves * 64, 0, s>>>(p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int c0 = 16 * w, cw = w == 0 ? 16 : C - 16;
    if (two)
      grid_grad_stage2<2><<<ncols, 512, 0, s>>>(part, q.dgrid, GH, GW, GD, C, pl.rg, pl.nyg, (float)GH / H, c0, cw);
    else
      grid_grad_stage2<1><<<ncols, 512, 0, s>>>(part, q.dgrid, GH, GW, GD, C, pl.rg, pl.nyg, (float)GH / H, c0, cw);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// the shared fast-shape table (launch.hip.h) restricted to what the contraction pass covers: one 16-column tile, or two
// channel windows for dgrid alone
bool apply_shape_ok(int Cin, int Cout, bool off) {
  return apply_fast_shape(Cin, Cout, off) && Cout * (Cin + (off ? 1 : 0)) <= 32;
}

bool slice_c_ok(int C) { return C == 1 || C == 2 || C == 4 || C == 8 || C == 12 || C == 16; }

}  // namespace

#ifdef HDRNET_TOOLS_BUILD
void grid_grad_set_trace(long long* device_buf) { g_gg_trace = device_buf; }
#endif

size_t apply_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                                      bool has_offset) {
  size_t bytes = 0;
  if (!apply_shape_ok(Cin, Cout, has_offset)) return 0;
  if (!gg_ws_bound(B, H, W, GH, GW, GD, Cout * (Cin + (has_offset ? 1 : 0)), &bytes)) return 0;
  return bytes;
}

bool apply_grid_grad_mfma_supported(const ApplyGradArgs& a) {
  size_t bytes = 0;
  if (!apply_shape_ok(a.Cin, a.Cout, a.has_offset) || !gg_ws_bound(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.Cout * a.Cj, &bytes))
    return false;
  if (a.workspace != nullptr && a.workspace_bytes >= bytes) return true;
  warn_small_workspace(a.workspace ? a.workspace_bytes : 0, bytes);
  return false;
}

// variant (tools A/B): 2 = bf16-split contraction.  fused: also write a.dguide / a.dinput.
static hipError_t apply_gg(const ApplyGradArgs& a, bool fused, hipStream_t s) {
  const GGPtrs q{a.guide, a.input, a.dout, a.grid, a.dgrid, fused ? a.dguide : nullptr,
                 fused ? a.dinput : nullptr};
  const int split = a.variant == 2 ? 1 : a.variant == 10 ? 2 : 0;  // tools: bf16 split / f16 hi-lo' split
#define HDRNET_CASE(CI, CO, OFF)                                                                  \
  if constexpr (CO * (CI + (OFF ? 1 : 0)) <= 16) {                                                \
    if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF)                                       \
      return gg_launch<CI, CO, OFF, true>(q, a.B, a.H, a.W, a.GH, a.GW, a.GD, a.workspace, a.workspace_bytes, s, \
                                          split, (a.variant >= 4 && a.variant <= 9) ? a.variant - 3 : 0); \
  } else if constexpr (CO * (CI + (OFF ? 1 : 0)) <= 32) {                                         \
    if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) {                                     \
      if (a.variant != 0 && a.variant != 3) return hipErrorNotSupported;                          \
      return gg_launch_windows<CI, CO, OFF>(q, a.B, a.H, a.W, a.GH, a.GW, a.GD, a.workspace, a.workspace_bytes, s); \
    }                                                                                             \
  }
  HDRNET_APPLY_FAST_SHAPES(HDRNET_CASE)
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_apply_grid_grad_mfma(const ApplyGradArgs& a, hipStream_t s, const char** name) {
  *name = a.variant == 2 ? "grid_grad_mfma/bf16x2" : a.variant == 10 ? "grid_grad_mfma/f16hilo" : "grid_grad_mfma";
  return apply_gg(a, false, s);
}

// Fused backward: dgrid AND dguide / dinput in one pass over the pixels (C % 4 == 0 shapes).
bool apply_bwd_fused_supported(const ApplyGradArgs& a) {
  if (!a.dgrid || !(a.dguide || a.dinput) || !a.grid) return false;
  if ((a.Cout * a.Cj) % 4 != 0 || a.Cout * a.Cj > 16 || a.Cj != 4 || ((uintptr_t)a.grid & 15u)) return false;
  if (a.dinput && a.Cin == 0) return false;
  return apply_grid_grad_mfma_supported(a);
}

hipError_t launch_apply_bwd_fused(const ApplyGradArgs& a, hipStream_t s, const char** name) {
  *name = a.variant == 2 ? "apply_bwd_fused/mfma-bf16x2" : a.variant == 10 ? "apply_bwd_fused/mfma-f16hilo" : "apply_bwd_fused/mfma";
  return apply_gg(a, true, s);
}

size_t slice_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int C) {
  size_t bytes = 0;
  if (!slice_c_ok(C) || !gg_ws_bound(B, H, W, GH, GW, GD, C, &bytes)) return 0;
  return bytes;
}

bool slice_grid_grad_mfma_supported(const SliceGradArgs& a) {
  size_t bytes = 0;
  if (!slice_c_ok(a.C) || !gg_ws_bound(a.B, a.H, a.W, a.GH, a.GW, a.GD, a.C, &bytes)) return false;
  if (a.workspace != nullptr && a.workspace_bytes >= bytes) return true;
  warn_small_workspace(a.workspace ? a.workspace_bytes : 0, bytes);
  return false;
}

static hipError_t slice_gg(const SliceGradArgs& a, bool fused, hipStream_t s) {
  const GGPtrs q{a.guide, nullptr, a.dout, a.grid, a.dgrid, fused ? a.dguide : nullptr, nullptr};
  const int split = a.variant == 2 ? 1 : a.variant == 10 ? 2 : 0;
#define HDRNET_CASE(CC)                                                                            \
  if (a.C == CC)                                                                                   \
  return gg_launch<0, CC, false, false>(q, a.B, a.H, a.W, a.GH, a.GW, a.GD, a.workspace, a.workspace_bytes, s, split)
  HDRNET_CASE(1);
  HDRNET_CASE(2);
  HDRNET_CASE(4);
  HDRNET_CASE(8);
  HDRNET_CASE(12);
  HDRNET_CASE(16);
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

hipError_t launch_slice_grid_grad_mfma(const SliceGradArgs& a, hipStream_t s, const char** name) {
  *name = a.variant == 2 ? "grid_grad_mfma/bf16x2" : "grid_grad_mfma";
  return slice_gg(a, false, s);
}

bool slice_bwd_fused_supported(const SliceGradArgs& a) {
  if (!a.dgrid || !a.dguide || !a.grid) return false;
  if (a.C % 4 != 0 || ((uintptr_t)a.grid & 15u)) return false;
  return slice_grid_grad_mfma_supported(a);
}

hipError_t launch_slice_bwd_fused(const SliceGradArgs& a, hipStream_t s, const char** name) {
  *name = "slice_bwd_fused/mfma";
  return slice_gg(a, true, s);
}

}  // namespace hdrnet_amd
