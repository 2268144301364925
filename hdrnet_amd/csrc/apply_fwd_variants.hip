// BENCHMARK-ONLY alternatives to the shipped forward kernel (apply_fwd_rows.hip).
//
// Nothing here is reachable with flags == 0: the C-ABI's *_ex entry points route a non-zero
// variant number (flags bits 8..15) to launch_apply_fwd_variant, and tools/ab_bench.py times
// the variants interleaved with the product kernel in one process.  They stay in the tree
// because DESIGN.md section 4 quotes their timings as the evidence for the design:
//
//   2        one wavefront per tile (no workgroup barrier, per-wave LDS image)
//   3..6     persistent software-pipelined stream, {4,5,6,3} workgroups per CU
//   7        product kernel with naive per-lane strided stores (no LDS transpose)
//   8        product kernel with nontemporal lane-contiguous input loads + LDS transpose
//   9-11     Q = 2, 3, 4 quads per thread, all loads issued first (more bytes in flight per wave)
//   101      memory skeleton: the product kernel's loads / stores / launch shape, no slicing
//   103-105  memory skeletons with lane-contiguous accesses on {both, loads only, stores only}
//   106      103 with nontemporal input loads (the shipped kernel's access pattern)
//   107      an EMPTY kernel launched with the product's geometry (dispatch + inter-kernel gap alone)
//   108      skeleton 106 on flat 1024-pixel tasks (row boundaries ignored)
//
// Skeletons do NOT compute the op (their name says ABLATION); the others are checked for
// parity by tests/test_gpu_parity.py like the product kernel.
#include <hip/hip_runtime.h>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

constexpr int kVariantWave = 2;
constexpr int kVariantStream = 3;  // .. 6
constexpr int kVariantDirectStores = 7;
constexpr int kVariantNtLoads = 8;

// Variant 107: a launch of the product's geometry (workgroups, threads, LDS per workgroup) that does nothing --
// what a launch costs before it moves a byte: dispatch of every workgroup + the gap between dependent kernels.
__global__ __launch_bounds__(256) void apply_fwd_empty(float* out) {
  extern __shared__ float lds_empty[];
  if (out == nullptr) lds_empty[threadIdx.x] = 0.f;  // never taken; keeps the LDS allocation
}

// Variant 108: skeleton 106's accesses on FLAT tasks -- workgroup b moves pixels [1024 b, 1024 b + 1024) of the
// image taken as one run of B * H * W pixels, whatever rows they fall in: 1920 x 1080 = 2025 four-wave workgroups
// against the chip's 2048 slots, where row segments need 2160.  What a flattened decomposition of a one-round frame
// could gain, measured before building it.
__global__ __launch_bounds__(256) void apply_fwd_skeleton_flat(const float* __restrict__ guide,
                                                               const float* __restrict__ input,
                                                               float* __restrict__ out, long long npx) {
  const long long p0 = (long long)blockIdx.x * 1024;
  const int n = (int)min((long long)1024, npx - p0);  // pixels of this task (multiple of 4)
  const int t = threadIdx.x;
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (4 * t < n) g4 = load_stream4(guide + p0 + 4 * t);
  const float4* ip = reinterpret_cast<const float4*>(input + p0 * 3);
  float4* op = reinterpret_cast<float4*>(out + p0 * 3);
  const int nq = n * 3 / 4;
  float4 v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = t + 256 * k;
    if (e < nq) v[k] = load_stream4(reinterpret_cast<const float*>(ip + e));
  }
  const float gq = g4.x + g4.y + g4.z + g4.w;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = t + 256 * k;
    if (e < nq) {
      v[k].x *= gq; v[k].y *= gq; v[k].z *= gq; v[k].w *= gq;
      op[e] = v[k];
    }
  }
}

// ---- memory skeletons ---------------------------------------------------------------------
// MODE 1: thread = 4 consecutive pixels, exactly the product kernel's global accesses.
// MODE 3 / 4 / 5: thread t touches float4 number t + k * blockDim of the segment's input /
// output (lane-contiguous) for {loads and stores, loads only, stores only}; the other side keeps
// the per-lane 48-B stride.  The guide is read as in the product kernel in every mode.
template <int MODE>
__global__ __launch_bounds__(256) void apply_fwd_skeleton(
    const float* __restrict__ guide, const float* __restrict__ input, float* __restrict__ out,
    int H, int W, int nseg, int seg) {
  constexpr int CIN = 3, COUT = 3;
  const int bid = blockIdx.x;
  const int segi = bid % nseg;
  const int row = bid / nseg;
  const int xs = segi * seg;
  const int xe = min(xs + seg, W);
  const int x = xs + kPxPerThread * threadIdx.x;
  const bool active = x < xe;
  const size_t p = (size_t)row * W + x;
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) g4 = (MODE == 6) ? load_stream4(guide + p) : *reinterpret_cast<const float4*>(guide + p);
  if constexpr (MODE == 1) {
    if (!active) return;
    const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
    float4* op = reinterpret_cast<float4*>(out + p * COUT);
    float4 v[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = ip[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      v[q].x *= g4.x; v[q].y *= g4.y; v[q].z *= g4.z; v[q].w *= g4.w;
      op[q] = v[q];
    }
  } else {
    const int nthreads = blockDim.x;
    const size_t seg_p = (size_t)row * W + xs;
    const int nq = (xe - xs) * CIN / 4;  // float4 count of the segment's input
    const float4* ip = reinterpret_cast<const float4*>(input + seg_p * CIN);
    float4* op = reinterpret_cast<float4*>(out + seg_p * COUT);
    float4 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = (MODE == 5) ? (int)threadIdx.x * 3 + k : (int)threadIdx.x + k * nthreads;
      if (e < nq) {
        if constexpr (MODE == 6) v[k] = load_stream4(reinterpret_cast<const float*>(ip + e));
        else v[k] = ip[e];
      }
    }
    const float gq = g4.x + g4.y + g4.z + g4.w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = (MODE == 4) ? (int)threadIdx.x * 3 + k : (int)threadIdx.x + k * nthreads;
      if (e < nq) {
        v[k].x *= gq; v[k].y *= gq; v[k].z *= gq; v[k].w *= gq;
        op[e] = v[k];
      }
    }
  }
}

// ---- Q quads per thread: more bytes in flight per wave ---------------------------------------------
// The shipped kernel holds 4 KB of loads per wave and is bound by bytes in flight (10 workgroups per
// CU have to cover HBM latency + their own compute).  Here a thread owns Q quads -- quad q of thread t
// is pixels xs + 4 (q * blockDim + t) .. +3, so quad q of a wave is still one dense 256-pixel run --
// issues ALL its loads first and then slices / stores quad by quad through the wave's slab.
template <int CIN, int COUT, bool OFFSET, int Q>
__global__ __launch_bounds__(256) void apply_fwd_rows_multiquad(
    const float* __restrict__ grid, const float* __restrict__ guide, const float* __restrict__ input,
    float* __restrict__ out, int H, int W, int GH, int GW, int GD, int nseg, int seg,
    int slab_offset_floats, float scale_x, float scale_y) {
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
  const int nthreads = blockDim.x;

  float4 g4[Q], iv[Q][CIN];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int x = xs + kPxPerThread * (q * nthreads + (int)threadIdx.x);
    g4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < xe) {
      const size_t p = (size_t)row * W + x;
      g4[q] = *reinterpret_cast<const float4*>(guide + p);
      const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
      for (int k = 0; k < CIN; ++k) iv[q][k] = ip[k];
    }
  }

  const RowCtx r = stage_row<C, false>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);

  float4* slab = reinterpret_cast<float4*>(colY + slab_offset_floats) + (threadIdx.x >> 6) * (64 * COUT);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int x = xs + kPxPerThread * (q * nthreads + (int)threadIdx.x);
    const bool active = x < xe;
    const float gs[4] = {g4[q].x, g4[q].y, g4[q].z, g4[q].w};
    const float xf0 = (float)x + 0.5f;
    const float* inf = reinterpret_cast<const float*>(iv[q]);
    float4 ov[COUT];
    float* of = reinterpret_cast<float*>(ov);
    if (active) {
#pragma unroll
      for (int k = 0; k < kPxPerThread; ++k) {
        float in[CIN], o[COUT];
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
        slice_apply_pixel<CIN, COUT, OFFSET>(r, xf0 + (float)k, gs[k], in, o);
#pragma unroll
        for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
      }
#pragma unroll
      for (int k = 0; k < COUT; ++k) slab[lane * COUT + k] = ov[k];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int wave_x0 = xs + kPxPerThread * (q * nthreads + (int)(threadIdx.x & ~63u));
    const int nvalid = (min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0) * COUT / 4;  // float4s
    float4* gp = reinterpret_cast<float4*>(out + ((size_t)row * W + wave_x0) * COUT);
#pragma unroll
    for (int k = 0; k < COUT; ++k) {
      const int e = lane + 64 * k;
      if (e < nvalid) gp[e] = slab[e];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

template <int Q>
hipError_t launch_multiquad(const ApplyArgs& a, hipStream_t s, const char** name) {
  // segments of T * 4 * Q pixels, balanced over the row like make_row_plan
  int best_T = 0, best_nseg = 0;
  long long best_waste = -1;
  for (int T : {256, 192, 128, 64}) {
    const int span = T * kPxPerThread * Q;
    const int nseg = (a.W + span - 1) / span;
    const long long waste = (long long)nseg * span - a.W;
    if (best_waste < 0 || waste < best_waste) {
      best_waste = waste;
      best_T = T;
      best_nseg = nseg;
    }
  }
  const int seg = round_up((a.W + best_nseg - 1) / best_nseg, 4);
  int threads = round_up((seg + kPxPerThread * Q - 1) / (kPxPerThread * Q), 64);
  if (threads > 256) threads = 256;
  (void)best_T;
  const int max_cols = max_cols_for(seg, a.GW, a.W);
  const int slab_off = round_up(max_cols * a.GD * 12, 4);
  const size_t lds = ((size_t)slab_off + (size_t)(threads / 64) * 64 * kPxPerThread * 3) * sizeof(float);
  if (lds > 64 * 1024) return hipErrorNotSupported;
  const long long nblocks = (long long)a.B * a.H * best_nseg;
  apply_fwd_rows_multiquad<3, 3, true, Q><<<(unsigned)nblocks, threads, lds, s>>>(
      a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, best_nseg, seg, slab_off,
      (float)a.GW / a.W, (float)a.GH / a.H);
  *name = Q == 2 ? "apply_fwd_rows/multiquad2" : (Q == 3 ? "apply_fwd_rows/multiquad3" : "apply_fwd_rows/multiquad4");
  return hipGetLastError();
}

// ---- one wavefront per output tile ------------------------------------------------------
// A tile is `tile_w` consecutive pixels of one image row (tile_w <= 256, 4 per lane).
// Each wave stages its own y-pre-lerped columns (<= ~5 of them) in a private LDS
// region and then slices its pixels: no workgroup barrier, waves are independent, and a
// workgroup is just `waves_per_block` consecutive tiles.
template <int CIN, int COUT, bool OFFSET>
__global__ __launch_bounds__(256) void apply_fwd_wave_vec4(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, float* __restrict__ out, int H, int W, int GH, int GW,
    int GD, int tiles_per_row, int tile_w, long long ntiles, int lds_floats_per_wave,
    float scale_x, float scale_y) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  extern __shared__ __attribute__((aligned(16))) float colY_all[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const long long tile = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  if (tile >= ntiles) return;
  float* colY = colY_all + wave * lds_floats_per_wave;
  const int ti = (int)(tile % tiles_per_row);
  const long long row = tile / tiles_per_row;  // = b * H + y
  const int y = (int)(row % H);
  const long long b = row / H;
  const int xs = ti * tile_w;
  const int xe = min(xs + tile_w, W);
  const float* grid_b = grid + (size_t)b * GH * GW * GD * C;

  const int x = xs + kPxPerThread * lane;
  const bool active = x < xe;
  const size_t p = (size_t)row * W + x;

  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 iv[(CIN * kPxPerThread) / 4];
  if (active) {
    g4 = *reinterpret_cast<const float4*>(guide + p);
    const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
    for (int q = 0; q < (CIN * kPxPerThread) / 4; ++q) iv[q] = ip[q];
  }

  const RowCtx r = stage_row<C, true>(colY, grid_b, y, xs, xe, GH, GW, GD, scale_x, scale_y);
  if (!active) return;

  const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
  const float xf0 = (float)x + 0.5f;
  const float* inf = reinterpret_cast<const float*>(iv);
  float4 ov[(COUT * kPxPerThread) / 4];
  float* of = reinterpret_cast<float*>(ov);
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) {
    float in[CIN], o[COUT];
#pragma unroll
    for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
    slice_apply_pixel<CIN, COUT, OFFSET>(r, xf0 + (float)k, gs[k], in, o);
#pragma unroll
    for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
  }
  float4* op = reinterpret_cast<float4*>(out + p * COUT);
#pragma unroll
  for (int q = 0; q < (COUT * kPxPerThread) / 4; ++q) op[q] = ov[q];
}

// ---- persistent, balanced, software-pipelined streaming variant ---------------------------
// The launch is sized to what the chip holds at once (CUs x blocks_per_cu workgroups of 4
// waves).  Every WAVE owns one contiguous, equal share of the image's pixel quads and walks
// it in chunks of <= 64 quads (256 pixels, never across a row end).  The loop is software-
// pipelined around the in-order vmcnt counter of CDNA (which counts stores as well): per
// chunk i the wave
//   a. waits for G_i, the two grid-row slices of chunk i (issued one iteration ago), blends
//      them into its private LDS column image,
//   b. waits for P_i, the chunk's guide/input quads (also issued one iteration ago),
//   c. issues G_{i+1}, then d. P_{i+1}  -- BEFORE chunk i's stores, so that the waits of
//      the next iteration never sit behind a store or a younger load,
//   e. slices chunk i, f. transposes through LDS and stores.
// Every wave therefore always has the next chunk's 4 KiB in flight while it computes, all
// waves finish together (no partially filled last round of workgroups), and the only ramp
// left is one load latency at the start and one chunk of compute at the end of the launch.
// LDS traffic of ONE wave needs no fence: the LDS executes a wave's instructions in order, so
// a ds_read issued after a ds_write of the same wave sees all 64 lanes' data.  (A
// `fence(release, "wavefront")` would cost an s_waitcnt vmcnt(0), i.e. drain the prefetch.)
// The scheduling barrier only keeps the compiler from moving LDS accesses across.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

constexpr int kStageRegs = 2;  // float4 per grid row per lane held in flight (<= 128 float4 / row)

template <int CIN>
struct QuadData {
  float4 g;
  float4 in[CIN];
};

// Unconditional: the caller clamps `quad` to a valid quad (idle lanes re-load the chunk's last
// quad), so the loop body has no exec-masked branch around a VMEM instruction and the
// compiler's s_waitcnt vmcnt(N) counts stay exact.
template <int CIN>
__device__ __forceinline__ QuadData<CIN> load_quad(const float* __restrict__ guide,
                                                   const float* __restrict__ input,
                                                   long long quad) {
  QuadData<CIN> d;
  d.g = reinterpret_cast<const float4*>(guide)[quad];
  const float4* ip = reinterpret_cast<const float4*>(input) + quad * CIN;
#pragma unroll
  for (int q = 0; q < CIN; ++q) d.in[q] = ip[q];
  return d;
}

// Where a chunk sits and which grid data it needs (all wave-uniform).
struct ChunkGeom {
  int y, xs, len;        // image row, first pixel, quads
  long long b;           // image
  int gy0c, gy1c, gxlo, n4;  // clamped grid rows, first column, float4 count of the column image
  float wy0, wy1;
};

template <int C>
__device__ __forceinline__ ChunkGeom chunk_geom(long long b, int y, int xq, int len, int GH, int GW,
                                                int GD, float scale_x, float scale_y) {
  ChunkGeom c;
  c.b = b;
  c.y = y;
  c.xs = xq * 4;
  c.len = len;
  const float gyf = mul_rn(y + 0.5f, scale_y);
  const int gy0 = floor_to_int(gyf - 0.5f);
  c.wy0 = tent_weight(gy0 + 0.5f, gyf);
  c.wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
  c.gy0c = clamp_index(gy0, 0, GH - 1);
  c.gy1c = clamp_index(gy0 + 1, 0, GH - 1);
  const int xe = c.xs + len * 4;
  c.gxlo = clamp_index(floor_to_int(mul_rn(c.xs + 0.5f, scale_x) - 0.5f), 0, GW - 1);
  const int gxhi = clamp_index(floor_to_int(mul_rn(xe - 1 + 0.5f, scale_x) - 0.5f) + 1, 0, GW - 1);
  c.n4 = (gxhi - c.gxlo + 1) * GD * C / 4;
  return c;
}

struct StageRegs {
  float4 a[kStageRegs], b[kStageRegs];
};

template <int C>
__device__ __forceinline__ StageRegs stage_issue(const float* __restrict__ grid, const ChunkGeom& c,
                                                 int GH, int GW, int GD, int lane) {
  StageRegs r;
  const float* gb = grid + (size_t)c.b * GH * GW * GD * C;
  const float4* a4 = reinterpret_cast<const float4*>(gb + ((size_t)(c.gy0c * GW + c.gxlo) * GD) * C);
  const float4* b4 = reinterpret_cast<const float4*>(gb + ((size_t)(c.gy1c * GW + c.gxlo) * GD) * C);
#pragma unroll
  for (int k = 0; k < kStageRegs; ++k) {
    const int e = min(lane + 64 * k, c.n4 - 1);  // clamped: idle lanes duplicate the last element
    r.a[k] = a4[e];
    r.b[k] = b4[e];
  }
  return r;
}

template <int C>
__device__ __forceinline__ RowCtx stage_commit(float* __restrict__ colY, const StageRegs& r,
                                               const ChunkGeom& c, int GW, int GD, float scale_x,
                                               int lane) {
  float4* d4 = reinterpret_cast<float4*>(colY);
#pragma unroll
  for (int k = 0; k < kStageRegs; ++k) {
    const int e = min(lane + 64 * k, c.n4 - 1);  // duplicates write the same value
    const float4 a = r.a[k], b = r.b[k];
    d4[e] = make_float4(c.wy0 * a.x + c.wy1 * b.x, c.wy0 * a.y + c.wy1 * b.y,
                        c.wy0 * a.z + c.wy1 * b.z, c.wy0 * a.w + c.wy1 * b.w);
  }
  wave_lds_order();
  const int col_bytes = GD * C * (int)sizeof(float);
  return RowCtx{colY, scale_x, (float)GD, c.gxlo, col_bytes, (0 - c.gxlo) * col_bytes,
                (GW - 1 - c.gxlo) * col_bytes, (GD - 1) * C * (int)sizeof(float)};
}

template <int CIN, int COUT, bool OFFSET>
__global__ __launch_bounds__(256) void apply_fwd_stream_vec4(
    const float* __restrict__ grid, const float* __restrict__ guide,
    const float* __restrict__ input, float* __restrict__ out, int H, int Wq, int GH, int GW,
    int GD, long long nquads, long long quads_per_wave, int lds_floats_per_wave, float scale_x,
    float scale_y) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  static_assert(C % 4 == 0, "float4 column image");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float* colY = lds_all + wave * lds_floats_per_wave;
  float4* slab = reinterpret_cast<float4*>(colY + (lds_floats_per_wave - 64 * kPxPerThread * COUT));
  const long long wid = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
  long long pos = wid * quads_per_wave;
  const long long end = min(pos + quads_per_wave, nquads);
  if (pos >= end) return;
  // (b, y, xq) of `pos`, advanced incrementally; everything here is wave-uniform.
  long long row = pos / Wq;
  int xq = (int)(pos - row * Wq);
  int y = (int)(row % H);
  long long b = row / H;
  int len = (int)min((long long)min(64, Wq - xq), end - pos);

  ChunkGeom cg = chunk_geom<C>(b, y, xq, len, GH, GW, GD, scale_x, scale_y);
  StageRegs sr = stage_issue<C>(grid, cg, GH, GW, GD, lane);                      // G_0
  QuadData<CIN> cur = load_quad<CIN>(guide, input, pos + min(lane, len - 1));      // P_0

  while (true) {
    // a. G_i -> this wave's LDS column image (frees the staging registers)
    const RowCtx r = stage_commit<C>(colY, sr, cg, GW, GD, scale_x, lane);
    // b./c. next chunk: geometry, then its grid rows and its guide / input quads go out NOW --
    //       before chunk i is sliced and before its stores -- so they fly during the slicing
    //       and no later wait sits behind a store.  (Past the end: re-load this chunk, unused.)
    const long long npos = pos + len;
    int nxq = xq + len, ny = y;
    long long nb = b;
    if (nxq == Wq) {
      nxq = 0;
      if (++ny == H) {
        ny = 0;
        ++nb;
      }
    }
    const int nlen = npos < end ? (int)min((long long)min(64, Wq - nxq), end - npos) : 0;
    const bool more = nlen > 0;
    const ChunkGeom ncg = more ? chunk_geom<C>(nb, ny, nxq, nlen, GH, GW, GD, scale_x, scale_y) : cg;
    sr = stage_issue<C>(grid, ncg, GH, GW, GD, lane);
    const QuadData<CIN> nxt =
        load_quad<CIN>(guide, input, more ? npos + min(lane, nlen - 1) : pos + min(lane, len - 1));
    // d. slice chunk i (idle lanes slice a duplicate of the last quad; their result is unused)
    float4 ov[COUT];
    {
      const float gs[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w};
      const float xf0 = (float)(cg.xs + 4 * min(lane, len - 1)) + 0.5f;
      const float* inf = reinterpret_cast<const float*>(cur.in);
      float* of = reinterpret_cast<float*>(ov);
#pragma unroll
      for (int k = 0; k < kPxPerThread; ++k) {
        float in[CIN], o[COUT];
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
        slice_apply_pixel<CIN, COUT, OFFSET>(r, xf0 + (float)k, gs[k], in, o);
#pragma unroll
        for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
      }
    }
    // e. transpose through the wave's slab; lane-contiguous 16-B stores.  Idle lanes write /
    //    store a duplicate of the last valid element (same value, same address).
    {
      const int wl = min(lane, len - 1);
#pragma unroll
      for (int q = 0; q < COUT; ++q) slab[wl * COUT + q] = ov[q];
      wave_lds_order();
      const int nvalid = len * COUT;
      float4* gp = reinterpret_cast<float4*>(out) + pos * COUT;
      // all slab reads into distinct registers first: re-using one register quad for the three
      // stores would make each store wait (vmcnt) for the previous one -- and, the counter
      // being in-order, for the prefetch issued before it.
      float4 tv[COUT];
#pragma unroll
      for (int k = 0; k < COUT; ++k) tv[k] = slab[min(lane + 64 * k, nvalid - 1)];
#pragma unroll
      for (int k = 0; k < COUT; ++k) gp[min(lane + 64 * k, nvalid - 1)] = tv[k];
      wave_lds_order();
    }
    if (!more) break;
    cur = nxt;
    cg = ncg;
    pos = npos;
    len = nlen;
    xq = nxq;
    y = ny;
    b = nb;
  }
}

template <int CIN, int COUT, bool OFFSET>
hipError_t launch_variant_t(const ApplyArgs& a, const Plan& pl, hipStream_t s, const char** name) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  const float sx = (float)a.GW / a.W, sy = (float)a.GH / a.H;
  if (a.variant == kVariantWave) {
    // One wavefront per tile of <= 256 pixels, balanced over the row.
    const int tiles_per_row = (a.W + 64 * kPxPerThread - 1) / (64 * kPxPerThread);
    const int tile_w = round_up((a.W + tiles_per_row - 1) / tiles_per_row, 4);
    const long long ncol = ((long long)(tile_w - 1) * a.GW) / a.W + 4;
    const int cols = (int)(ncol < a.GW ? ncol : a.GW);
    const int lds_floats = round_up(cols * a.GD * C, 4);
    const long long ntiles = (long long)a.B * a.H * tiles_per_row;
    const int waves = 4;
    apply_fwd_wave_vec4<CIN, COUT, OFFSET>
        <<<(unsigned)((ntiles + waves - 1) / waves), waves * 64,
           (size_t)waves * lds_floats * sizeof(float), s>>>(
            a.grid, a.guide, a.input, a.out, a.H, a.W, a.GH, a.GW, a.GD, tiles_per_row, tile_w,
            ntiles, lds_floats, sx, sy);
    *name = "apply_fwd_wave/vec4";
    return hipGetLastError();
  }
  if constexpr (C % 4 == 0) {
    if (a.variant >= kVariantStream && a.variant < kVariantStream + 4) {
      // Persistent balanced stream: CUs x blocks_per_cu workgroups of 4 waves.
      const int bpc_table[4] = {4, 5, 6, 3};
      const int blocks_per_cu = bpc_table[(a.variant - kVariantStream) & 3];
      const int waves = 4;
      const long long nquads = (long long)a.B * a.H * (a.W / 4);
      long long nwaves = (long long)num_cus() * blocks_per_cu * waves;
      if (nwaves > (nquads + 63) / 64) nwaves = (nquads + 63) / 64;  // small images: 1 chunk each
      nwaves = (nwaves + waves - 1) / waves * waves;
      const long long qpw = (nquads + nwaves - 1) / nwaves;
      const int cols = max_cols_for(64 * kPxPerThread, a.GW, a.W);
      if (cols * a.GD * C / 4 <= 64 * kStageRegs) {
        const int lds_floats = round_up(cols * a.GD * C, 4) + 64 * kPxPerThread * COUT;
        apply_fwd_stream_vec4<CIN, COUT, OFFSET>
            <<<(unsigned)(nwaves / waves), waves * 64, (size_t)waves * lds_floats * sizeof(float), s>>>(
                a.grid, a.guide, a.input, a.out, a.H, a.W / 4, a.GH, a.GW, a.GD, nquads, qpw,
                lds_floats, sx, sy);
        *name = "apply_fwd_stream/vec4";
        return hipGetLastError();
      }
    }
  }
  if constexpr (CIN == 3 && COUT == 3 && OFFSET) {
    const unsigned nblocks = (unsigned)((long long)a.B * a.H * pl.nseg);
    switch (a.variant) {
      case 101:
        apply_fwd_skeleton<1><<<nblocks, pl.threads, 0, s>>>(a.guide, a.input, a.out, a.H, a.W, pl.nseg, pl.seg);
        *name = "ABLATION/memory-skeleton";
        return hipGetLastError();
      case 103:
        apply_fwd_skeleton<3><<<nblocks, pl.threads, 0, s>>>(a.guide, a.input, a.out, a.H, a.W, pl.nseg, pl.seg);
        *name = "ABLATION/skeleton ld-contig st-contig";
        return hipGetLastError();
      case 104:
        apply_fwd_skeleton<4><<<nblocks, pl.threads, 0, s>>>(a.guide, a.input, a.out, a.H, a.W, pl.nseg, pl.seg);
        *name = "ABLATION/skeleton ld-contig st-strided";
        return hipGetLastError();
      case 105:
        apply_fwd_skeleton<5><<<nblocks, pl.threads, 0, s>>>(a.guide, a.input, a.out, a.H, a.W, pl.nseg, pl.seg);
        *name = "ABLATION/skeleton ld-strided st-contig";
        return hipGetLastError();
      case 106:
        apply_fwd_skeleton<6><<<nblocks, pl.threads, 0, s>>>(a.guide, a.input, a.out, a.H, a.W, pl.nseg, pl.seg);
        *name = "ABLATION/skeleton nt-ld-contig st-contig";
        return hipGetLastError();
      case 108: {
        const long long npx = (long long)a.B * a.H * a.W;
        apply_fwd_skeleton_flat<<<(unsigned)((npx + 1023) / 1024), 256, 0, s>>>(a.guide, a.input, a.out, npx);
        *name = "ABLATION/skeleton flat 1024-px tasks";
        return hipGetLastError();
      }
      case 107: {
        // LDS per workgroup as the product kernel's: padded image + one (guide + in / out) slab per wave
        const size_t lds = ((size_t)round_up(((pl.seg - 1) * a.GW / a.W + 4) * (a.GD + 2) * C, 4) +
                            (size_t)(pl.threads / 64) * 64 * kPxPerThread * 4) * sizeof(float);
        apply_fwd_empty<<<nblocks, pl.threads, lds, s>>>(a.out);
        *name = "ABLATION/empty launch, product geometry";
        return hipGetLastError();
      }
      default:
        break;
    }
  }
  return hipErrorNotSupported;
}

}  // namespace

// Experiment knobs of the tools build (hdrnet_tools_set_knob; include/hdrnet_amd_tools.h).
static int g_knob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void tools_set_knob(int idx, int value) {
  if (idx >= 0 && idx < 8) g_knob[idx] = value;
}
int tools_knob(int idx) { return idx >= 0 && idx < 8 ? g_knob[idx] : 0; }

// hipErrorNotSupported: no such variant for this shape -- the caller launches the product kernel.
hipError_t launch_apply_fwd_variant(const ApplyArgs& a, hipStream_t s, const char** name) {
  const bool aligned = (((uintptr_t)a.guide | (uintptr_t)a.input | (uintptr_t)a.out |
                         (uintptr_t)a.grid) & 15u) == 0;
  const Plan pl = make_row_plan(a.W, a.GW, aligned);
  if (!pl.vec4) return hipErrorNotSupported;
  // (variants 20 .. 72 -- the product kernel's load / store / pixel-phase flavours, the ticketed tail, the timeline
  //  trace -- were removed in round 5: profiles/r02 .. r04 hold their measurements, the history their code)
  if (a.variant == kVariantDirectStores) return launch_apply_fwd_rows_direct_stores(a, s, name, 0);
  if (a.variant == kVariantNtLoads) return launch_apply_fwd_rows_direct_stores(a, s, name, 1);
  if (a.Cin == 3 && a.Cout == 3 && a.has_offset) {
    if (a.variant == 9) return launch_multiquad<2>(a, s, name);
    if (a.variant == 10) return launch_multiquad<3>(a, s, name);
    if (a.variant == 11) return launch_multiquad<4>(a, s, name);
  }
#define HDRNET_CASE(CI, CO, OFF) \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) return launch_variant_t<CI, CO, OFF>(a, pl, s, name)
  HDRNET_CASE(3, 3, true);
  HDRNET_CASE(3, 3, false);
  HDRNET_CASE(3, 4, true);
  HDRNET_CASE(1, 1, true);
  HDRNET_CASE(1, 1, false);
  HDRNET_CASE(1, 3, true);
  HDRNET_CASE(4, 4, true);
#undef HDRNET_CASE
  return hipErrorNotSupported;
}

}  // namespace hdrnet_amd
