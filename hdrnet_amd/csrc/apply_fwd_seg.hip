// BilateralSliceApply forward for gfx950 -- the north-star kernel.
//
// Reference semantics: hdrnet/ops/bilateral_slice_apply.cc:24-82 (the CUDA twin,
// bilateral_slice_apply.cu.cc:36-126, assigns one thread per output CHANNEL and re-derives all
// weights for each of its 32 scattered grid loads).
//
// The op is an HBM stream -- 4 B guide + 4*Cin B input in, 4*Cout B out per pixel -- next to a
// 96 KiB grid that never leaves L2.  Decomposition:
//
//   * a workgroup owns one SEGMENT OF ONE IMAGE ROW (<= 1024 px, 4 px per lane).  For a row,
//     gy0 / gy1 and both y weights are wave-uniform, so the workgroup first blends the two grid
//     rows it needs into LDS ("y-pre-lerp"); a pixel then gathers 2(x) x 2(z) coefficient vectors
//     instead of 8 and computes 2 sqrt instead of the reference's 96.
//   * PADDED LDS image.  Column j of the image is grid column clamp(cmin + j, 0, GW-1) and carries
//     GD + 2 z-planes, plane p = grid plane clamp(p - 1, 0, GD-1).  The reference clamps INDICES
//     but not weights (bilateral_slice_apply.cc:58-68); with the clamped copies materialised once
//     per workgroup a pixel's four coefficient vectors sit at a0, a0 + CB, a0 + colb,
//     a0 + colb + CB: one address, immediate offsets, no per-pixel min / max.  One vector is C
//     contiguous floats read as ds_read_b128; the 48-B stride (C = 12) keeps the data-dependent
//     z gather at its conflict floor.
//   * PIXEL LOADS are LDS-DMA (`buffer_load_dwordx4 ... lds`, nontemporal): each wave streams its
//     256-pixel run (1 KiB of guide, Cin KiB of input) lane-contiguously straight into its LDS slab,
//     no VGPRs held while in flight; a lane then reads its own 4 pixels back with ds_read_b128.  A
//     grid of about one round of workgroups (a single 1080p frame) takes per-lane loads instead.
//   * STORES leave through the same slab, transposed to lane-contiguous 16-B runs, as
//     `buffer_store_dwordx4` on a descriptor that covers exactly the row segment (lanes past the run
//     are dropped by the bounds check -- no predicate): write-through (sc0 sc1) where segments are
//     whole 128-B lines, nontemporal where not; the flavour is chosen per launch
//     (launch_apply_fwd_seg below; rows_common.hip.h).
//   * the per-pixel code is the LEAN form of seg_common.hip.h (v_fract x weight, float byte addresses,
//     clamp-modifier tents); the plain forward ships its SCALAR blend (kPixLeanScalar), the
//     instruction-bound guide-network / wire-format kernels the packed one; the segment's grid-column
//     window comes from a host-side table.
//   * 3-D launch grid (segment, row, batch): no integer division in the kernel; the resident waves
//     per CU are CAPPED (resident_cap_lds below).
//
// Numerics: the coordinate and weight expressions of the reference in the reference's order
// (products (x+.5)*scale_x, guide*GD explicitly rounded, see numerics.hip.h: mul_rn); wy is folded
// into the LDS image, the x weights come as wx1 = fract(gxf - .5) and w(x0, .) = wz - wz * wx1 (exact
// but for 1 ulp in the first half cell), max(., 0) of the z tent is the clamp of the subtraction,
// v_sqrt_f32 (1 ulp) stands in for sqrtf; differences stay at the 1e-7 level (tests/test_gpu_parity.py
// and test_gpu_fullsize.py hold rtol = atol = 1e-5 against the oracle and report the reference's own
// 1e-6 bar: worst / bar <= 0.25 at every config size).
//
// This file holds the PRODUCT's configurations only; the alternatives it was chosen against and their
// measurements: docs/EXPERIMENTS.md section 4.1 and "Kernel-file lab notes".
#include <hip/hip_runtime.h>

#include <cstdio>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
#include "seg_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

// Pixel loads: per-lane 16-B loads of the lane's own 4 pixels (grids of about one round of workgroups: a single 1080p
// frame), or LDS-DMA, nontemporal, as buffer_load ... lds: ONE per-lane offset register for all pieces and the run's
// end enforced by the descriptor (everything deeper).
constexpr int kLoadsLane = 0, kLoadsBufDmaNt = 4;
// Pixel phase (seg_common.hip.h): lean with the packed blend (the instruction-bound guide-network kernels), lean with
// a scalar blend (the plain forward).
constexpr int kPixLean = 1, kPixLeanScalar = 2;
// Output stores (lane-contiguous 16 B after the per-wave LDS transpose): buffer_store_dwordx4 ... nt, or sc0 sc1
// (write-through: the line does not stay dirty in L2) where every row segment is whole 128-B lines.
constexpr int kStoresBufNt = 2, kStoresBufSc01 = 4;

struct SegParams {
  const float* grid;
  const float* guide;
  const float* input;
  float* out;
  int H, W, GH, GW, GD;
  int y0;          // first frame row of a row-split launch (buffers hold rows y0 .. y0 + H - 1); else 0
  int grid_image;  // floats per image of the grid: GH * GW * GD * C (< 2^31, capi.hip check_common)
  SegTab tab;      // (cmin, ncols) of every segment of a row, from the host
  int seg;         // pixels per segment, multiple of 4
  int slab_off;    // float offset of the per-wave slabs in dynamic LDS (= size of the image)
  float scale_x, scale_y;
  float inv_col;   // 1 / (GD * C / VEC): column of a staging element by float multiply
  GuideNN gn;        // GUIDE_NN: the folded point-wise guide network (rows_common.hip.h)
  UpAdd up;          // UPADD: the coarser pyramid level to up-sample and add
};

template <int STORES>
constexpr int store_aux() {
  return STORES == kStoresBufSc01 ? kAuxSc0Sc1 : kAuxNt;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// N 1-KiB pieces of one buffer: piece K lands at dst_wave_base + 1 KiB * K.  The instruction's immediate
// offset (a compile-time constant, hence the recursion) is added to the memory address AND to the LDS
// address (LDS = M0 + offset + 16 * lane), so the LDS base stays the slab's for every piece.
template <int K, int N>
__device__ __forceinline__ void buf_dma_pieces(__amdgpu_buffer_rsrc_t rs, float* dst_wave_base, unsigned voff) {
  if constexpr (K < N) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst_wave_base, 16, voff, 0, 1024 * K, kAuxNt);
    buf_dma_pieces<K + 1, N>(rs, dst_wave_base, voff);
  }
}

// GUIDE_NN / UPADD: the guide is computed in registers from the input run
// the wave has just streamed in (no guide DMA: 24 instead of 28 B/px), and / or the coarser pyramid
// level is up-sampled and added before the store -- HDRNetPointwiseNNGuide / HDRNetGaussianPyrNN.
template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool GUIDE_NN, bool UPADD, int PIX>
__device__ __forceinline__ void seg_task(const SegParams& p, float* __restrict__ lds, const int segi, const int y,
                                         const int b) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  constexpr int SLABW = 64 * kPxPerThread * (CIN > COUT ? CIN : COUT);  // floats: in / out run of a wave
  constexpr bool DMA = LOADS == kLoadsBufDmaNt;
  constexpr int SLAB = SLABW + ((DMA && !GUIDE_NN) ? 64 * kPxPerThread : 0);  // + the guide run

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
  const int xs = segi * p.seg;
  const int xe = min(xs + p.seg, p.W);
  const float* grid_b = p.grid + (size_t)b * (unsigned)p.grid_image;

  const int x = xs + kPxPerThread * tid;
  const bool active = x < xe;
  const int wave_x0 = xs + kPxPerThread * 64 * wave;
  const int wave_px = min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0;  // <= 0: idle wave
  float4* slab = reinterpret_cast<float4*>(lds + p.slab_off + wave * SLAB);
  [[maybe_unused]] float4* gslab = slab + SLABW / 4;  // DMA only

  // Grid columns of this segment, unclamped: gx0 of the first pixel .. gx0 + 1 of the last.
  const SegCols sc = seg_cols_tab(p.tab, segi, xs, xe, p.scale_x);
  const int cmin = sc.cmin, ncols = sc.ncols;
  const int colb = (p.GD + 2) * CB;
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);

  // Addressing: a wave-uniform 64-bit segment base (SGPRs) + a 32-bit per-lane element offset.
  const size_t row = (unsigned)b * (unsigned)p.H + (unsigned)y;  // B, H <= 65535 (seg_geom)
  const unsigned lpx = kPxPerThread * (unsigned)tid;         // this lane's first pixel in the segment
  const unsigned wpx = kPxPerThread * 64u * (unsigned)wave;  // this wave's first pixel in the segment
  [[maybe_unused]] const float* gseg = GUIDE_NN ? nullptr : p.guide + (row * p.W + xs);
  const float* iseg = p.input + (row * p.W + xs) * CIN;

  // Pixel loads go out first: their HBM latency overlaps the (L2-resident) staging.
  float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 iv[CIN > 0 ? CIN : 1];
  if constexpr (LOADS == kLoadsLane) {
    if (active) {
      if constexpr (!GUIDE_NN) g4 = *reinterpret_cast<const float4*>(gseg + lpx);
#pragma unroll
      for (int q = 0; q < CIN; ++q) iv[q] = *reinterpret_cast<const float4*>(iseg + (lpx * CIN + 4 * q));
    }
  } else {
    static_assert(LOADS == kLoadsBufDmaNt, "per-lane loads or the buffer-form LDS-DMA");
    // LDS-DMA through buffer descriptors that end with the wave's run: every piece uses the same
    // per-lane offset (16 * lane) + an immediate; lanes past the run read zeros (never used).
    if (wave_px > 0) {
      const unsigned voff = 16u * (unsigned)lane;
      if constexpr (!GUIDE_NN) {
        const __amdgpu_buffer_rsrc_t grs = make_rsrc_uniform(gseg + wpx, (unsigned)wave_px * 4u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lptr_t) reinterpret_cast<float*>(gslab), 16, voff, 0, 0, kAuxNt);
      }
      const __amdgpu_buffer_rsrc_t irs = make_rsrc_uniform(iseg + wpx * CIN, (unsigned)wave_px * (4u * CIN));
      buf_dma_pieces<0, CIN>(irs, reinterpret_cast<float*>(slab), voff);
    }
  }

  stage_image<C>(lds, grid_b, y + p.y0, cmin, ncols, p.GH, p.GW, p.GD, p.scale_y, p.inv_col, tid, (int)blockDim.x);

  // x-only terms of this thread's 4 pixels (the lean pixel phase of seg_common.hip.h)
  XTermLean xl[kPxPerThread];
  const float xf0 = (float)x + 0.5f;  // (x + k) + 0.5f == xf0 + k exactly (x < 2^23)
  {
    const float colb_f = (float)colb, xbase_f = (float)(CB - cmin * colb);  // exact: |.| < 2^24
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) xl[k] = x_term_lean(xf0 + (float)k, p.scale_x, colb_f, xbase_f);
  }

  __syncthreads();  // image complete; with LDS-DMA in flight the compiler drains vmcnt here too
  // ---- this lane's 4 pixels ---------------------------------------------------------------------
  if constexpr (DMA) {
    if (active) {
      if constexpr (!GUIDE_NN) g4 = gslab[lane];
#pragma unroll
      for (int q = 0; q < CIN; ++q) iv[q] = slab[lane * CIN + q];
    }
  }
  // CIN == COUT: a lane reads and writes only ITS slab entries -- in place, no cross-lane hazard;
  // otherwise the input reads of all lanes must precede the output writes below.
  if constexpr (LOADS != kLoadsLane && CIN != COUT) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

  float gs[4] = {g4.x, g4.y, g4.z, g4.w};
  const float* inf = reinterpret_cast<const float*>(iv);
  float4 ov[COUT];
  float* of = reinterpret_cast<float*>(ov);
  if (active) {
    if constexpr (GUIDE_NN) {
      guide_nn_quad<CIN>(p.gn, inf, gs);
      if (p.gn.guide_out)  // wave-uniform
        *reinterpret_cast<float4*>(p.gn.guide_out + (row * p.W + xs) + lpx) = make_float4(gs[0], gs[1], gs[2], gs[3]);
    }
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) {
      float in[CIN > 0 ? CIN : 1], o[COUT];
#pragma unroll
      for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
      seg_pixel_lean<CIN, COUT, OFFSET, PIX == kPixLean>(lds, gd_f, zhi, colb, xl[k], gs[k], in, o);
#pragma unroll
      for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
    }
    if constexpr (UPADD) upadd_quad<COUT>(p.up, b, y, x, of);
#pragma unroll
    for (int q = 0; q < COUT; ++q) slab[lane * COUT + q] = ov[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // Store phase: lane l stores float4 number l + 64 k of the wave's output run -- every store
  // instruction covers one dense 1 KiB (per-lane 16-B stores at a 16*COUT-byte stride cost 6.5 us
  // per 4K frame, profiles/r01).
  float* oseg = p.out + (row * p.W + xs) * COUT;  // uniform
  {
    // descriptor over exactly this row segment: lanes past the run are dropped by the bounds check
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc_uniform(oseg, (unsigned)(xe - xs) * COUT * 4u);
#pragma unroll
    for (int k = 0; k < COUT; ++k)
      buf_store16<store_aux<STORES>()>(slab[lane + 64 * k], orsrc, (wpx * COUT + 4u * (unsigned)(lane + 64 * k)) * 4u);
  }
}

// The 3-D launch grid (segment, row, image): one task per workgroup, no index arithmetic.
template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool GUIDE_NN = false, bool UPADD = false,
          int PIX = kPixLeanScalar>
__global__ __launch_bounds__(256) void apply_fwd_seg(const SegParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  seg_task<CIN, COUT, OFFSET, LOADS, STORES, GUIDE_NN, UPADD, PIX>(p, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

struct SegGeom {
  Plan pl;
  int max_cols, slab_off;
  size_t lds;
  bool ok;
};

constexpr size_t kMaxLdsBytes = 64 * 1024;  // keep >= 2 workgroups per CU

SegGeom seg_geom(const ApplyArgs& a, bool dma, bool guide_map = true) {
  const int C = a.Cout * a.Cj;
  SegGeom g{};
  const bool aligned = (((guide_map ? (uintptr_t)a.guide : 0) | (uintptr_t)a.input | (uintptr_t)a.out |
                         (uintptr_t)a.grid) & 15u) == 0;
  g.pl = make_row_plan(a.W, a.GW, aligned);
  g.max_cols = (int)(((long long)(g.pl.seg - 1) * a.GW) / a.W + 4);
  g.slab_off = round_up(g.max_cols * (a.GD + 2) * C, 4);
  const int slabw = 64 * kPxPerThread * (a.Cin > a.Cout ? a.Cin : a.Cout) + ((dma && guide_map) ? 64 * kPxPerThread : 0);
  g.lds = ((size_t)g.slab_off + (size_t)(g.pl.threads / 64) * slabw) * sizeof(float);
  const long long nstage = (long long)g.max_cols * a.GD * C;
  g.ok = g.pl.vec4 && g.lds <= kMaxLdsBytes && nstage < (1 << 20) && a.B <= 65535 && a.H <= 65535 &&
         (long long)a.W * a.Cout * 4 < (1LL << 31);
  return g;
}

// RESIDENT-WAVE CAP (round 4).  The kernel needs only ~40 KB of loads in flight per CU to saturate its share of the HBM
// (Little's law on 26 KB/us per CU and ~1.5 us of fixed latency; 9 -> 7 -> 6 workgroups per CU: 39.4 / 39.3 / 39.7 us
// per 4K launch, profiles/r04/fwd_launch_shape.md), but what it does with more resident waves matters to the POWER
// MANAGER: with 27 waves per CU (9 three-wave workgroups, what the LDS footprint allows) a third of the boxes of the
// pool flip, for 50-200 ms at a time, into a state where the launch takes 42-46 us instead of 39 (higher clock,
// compute side power-braked: DESIGN.md section 5); with 24 waves the flips become rare, with 21 they are gone --
// 39.0-39.3 us on the boxes where the uncapped kernel runs 40.6-44.3, and THE SAME 38.0-39.5 us as uncapped on every
// steady box (20 boxes interleaved, profiles/r04/power_state/).  The cap is a floor on the dynamic LDS a workgroup
// asks for: the smallest allocation of which kResidentWaves / waves + 1 no longer fit into a CU's 160 KiB.
inline int resident_cap_wgs(int waves) {
  // 3-wave workgroups (4K): 7 = 21 waves -- free on steady boxes, no flips.  4-wave workgroups (4 x 1080p, 4000 x 3000;
  // 7 fit by LDS = 28 waves): 6 = 24 waves costs 0.0-0.2 us per launch, 5 costs 1 us at 4 x 1080p (profiles/r04/
  // power_state/occ2_*).  Other widths: 21 waves.
  return waves == 3 ? 7 : waves == 4 ? 6 : waves > 0 ? (21 / waves > 0 ? 21 / waves : 1) : 0;
}
inline size_t resident_cap_lds(size_t lds, int threads) {
  const int wgs = resident_cap_wgs(threads / 64);
  if (wgs < 1) return lds;
  const size_t floor_bytes = ((size_t)160 * 1024 / (size_t)(wgs + 1) / 256 + 1) * 256;
  return lds > floor_bytes ? lds : floor_bytes;
}

template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool GUIDE_NN = false, bool UPADD = false,
          int PIX = kPixLeanScalar>
hipError_t launch_seg_t(const ApplyArgs& a, hipStream_t s, const GuideNN& gn = GuideNN{nullptr, nullptr, nullptr, 0},
                        const UpAdd& up = UpAdd{nullptr, 0, 0, 0.f, 0.f}) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  SegGeom g = seg_geom(a, LOADS == kLoadsBufDmaNt, !GUIDE_NN);
  if (!g.ok) return hipErrorNotSupported;
  // (not for the per-lane-load flavour, which serves grids of about ONE round of workgroups -- a single 1080p frame:
  //  there every resident slot counts, 11.7 us uncapped / 12.0 at 7 workgroups per CU / 13.1 at 6;
  //  nor with the guide network fused: that kernel is VALU-bound and wants its waves, 44.8 -> 48.2 us capped at 4K)
  if constexpr (LOADS == kLoadsBufDmaNt && !GUIDE_NN) g.lds = resident_cap_lds(g.lds, g.pl.threads);
  SegParams p;
  p.gn = gn;
  p.up = up;
  p.grid = a.grid;
  p.guide = a.guide;
  p.input = a.input;
  p.out = a.out;
  p.H = a.H;
  p.W = a.W;
  p.GH = a.GH;
  p.GW = a.GW;
  p.GD = a.GD;
  p.y0 = a.y0;
  p.grid_image = a.GH * a.GW * a.GD * C;
  p.tab = make_seg_tab(a.W, g.pl.seg, g.pl.nseg, (float)a.GW / a.W);
  p.seg = g.pl.seg;
  p.slab_off = g.slab_off;
  p.scale_x = (float)a.GW / a.W;
  p.scale_y = (float)a.GH / a.frame_rows();
  p.inv_col = 1.0f / (float)(a.GD * (C / VEC));
  const dim3 grid3((unsigned)g.pl.nseg, (unsigned)a.H, (unsigned)a.B);
  apply_fwd_seg<CIN, COUT, OFFSET, LOADS, STORES, GUIDE_NN, UPADD, PIX><<<grid3, g.pl.threads, g.lds, s>>>(p);
  return hipGetLastError();
}

bool seg_shape(const ApplyArgs& a) { return apply_fast_shape(a.Cin, a.Cout, a.has_offset); }

}  // namespace

// The product configurations (measured per frame size across boxes, profiles/r02/d_ab_variants_*.txt):
//   * grids deeper than ~1.25 rounds of resident workgroups (4K, batched 1080p): LDS-DMA nontemporal loads;
//     stores write-through (sc0 sc1) where every row segment is whole 128-B lines (4K: 38.9-39.4 us, the
//     tightest spread of all flavours), else nontemporal (a write-through of a partial line costs a
//     read-modify-write: 4000-px rows 67 vs 58.6 us);
//   * a grid of about one round (one 1080p frame): per-lane loads + nontemporal stores -- with no second
//     round to overlap, the DMA's longer path to first use costs more than its registers save
//     (11.3-12.2 vs 12.5-13.7 us).
bool apply_fwd_seg_supported(const ApplyArgs& a) {
  if (!seg_shape(a)) return false;
  // stage_image reads the grid as float4 when C % 4 == 0.
  if ((a.Cout * a.Cj) % 4 == 0 && ((uintptr_t)a.grid & 15u)) return false;
  return seg_geom(a, true).ok && seg_geom(a, false).ok;
}

// The per-launch flavour choice for one shape (PIX: the pixel phase, DMAL: the DMA form).
template <int CI, int CO, bool OFF, int PIX, int DMAL>
hipError_t launch_seg_pick(const ApplyArgs& a, hipStream_t s) {
  const SegGeom g = seg_geom(a, true);
  const long long nblocks = (long long)g.pl.nseg * a.H * a.B;
  const long long one_round = (long long)num_cus() * (32 / (g.pl.threads / 64));  // workgroups resident at once
  const bool small = 4 * nblocks <= 5 * one_round;
  const bool whole_lines = ((uintptr_t)a.out % 128 == 0) && ((long long)g.pl.seg * a.Cout * 4) % 128 == 0 &&
                           ((long long)a.W * a.Cout * 4) % 128 == 0;
  const GuideNN gn{nullptr, nullptr, nullptr, 0};
  const UpAdd up{nullptr, 0, 0, 0.f, 0.f};
  // The per-launch flavour choice was measured on, and is instantiated for, the shape every BASELINE.json config has
  // (3 -> 3 with offset); the other fast shapes take the one flavour that is never far off (LDS-DMA loads,
  // nontemporal stores) -- a third of the instantiations for shapes no config names (hygiene).
  if constexpr (CI == 3 && CO == 3 && OFF) {
    if (small) return launch_seg_t<CI, CO, OFF, kLoadsLane, kStoresBufNt, false, false, PIX>(a, s, gn, up);
    if (whole_lines)
      return launch_seg_t<CI, CO, OFF, DMAL, kStoresBufSc01, false, false, PIX>(a, s, gn, up);
  }
  return launch_seg_t<CI, CO, OFF, DMAL, kStoresBufNt, false, false, PIX>(a, s, gn, up);
}

// Round 3 (profiles/r03/ab_variants_4k.txt, two boxes, interleaved): lean pixel phase 40.4 -> 39.6-39.8 us,
// buffer-form DMA 40.4 -> 39.5 us, both 39.3-39.5 us next to the no-compute skeleton's 39.0.  The SCALAR blend
// (v_fma_f32 for v_pk_fma_f32: 100 more instructions, bit-identical results) times the same as the packed one on
// steady boxes (a v_pk_fma_f32 issues as two passes on gfx950) and is the product because of the other boxes:
// where the power management falls into its high-clock state -- the compute side power-braked, 45-47 us per
// frame (profiles/r03/power/summary.txt) -- it spends less time there: 42.7 vs 44.6 us mean over four
// alternations on one such box, 41.4-41.5 vs 44.5-47.5 on two others, equal on a fourth (power/scalar_blend.txt).
constexpr int kProductPix = kPixLeanScalar, kProductDma = kLoadsBufDmaNt;
// The guide-network forwards are VALU-bound (16 features x 5 operations per pixel on top of the pixel phase): there
// the 100 extra instructions of the scalar blend cost 0.7-4 % (power/scalar_blend.txt), so they keep the packed one.
constexpr int kGuideNNPix = kPixLean;

hipError_t launch_apply_fwd_seg(const ApplyArgs& a, hipStream_t s, const char** name) {
  *name = "apply_fwd_seg/vec4";
#define HDRNET_CASE(CI, CO, OFF) \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) return launch_seg_pick<CI, CO, OFF, kProductPix, kProductDma>(a, s);
  HDRNET_APPLY_FAST_SHAPES(HDRNET_CASE)
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

// Fused guide network (+ optional guide copy): Cin = Cout in {3, 1} as the round-1 kernel offered.
bool apply_fwd_seg_nnguide_supported(const ApplyArgs& a, const float* guide_out) {
  if (!((a.Cin == 3 && a.Cout == 3) || (a.Cin == 1 && a.Cout == 1))) return false;
  if (((uintptr_t)guide_out & 15u) || ((a.Cout * a.Cj) % 4 == 0 && ((uintptr_t)a.grid & 15u))) return false;
  return seg_geom(a, true, false).ok;
}

hipError_t launch_apply_fwd_seg_nnguide(const ApplyArgs& a, const float* conv1, const float* conv2, int n_feats,
                                        float* guide_out, hipStream_t s, const char** name) {
  const GuideNN gn{conv1, conv2, guide_out, n_feats, a.fast_sigmoid, a.guide_prescaled};
  *name = "apply_fwd_seg/vec4+nnguide";
  // (stores: nontemporal.  The write-through form the plain 4K forward uses was timed here in round 5, interleaved:
  //  41.92 vs 41.74-41.99 us -- no difference on an instruction-bound kernel; profiles/r05/f2_prescale.md)
#define HDRNET_CASE(CI, CO, OFF)                          \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) \
    return launch_seg_t<CI, CO, OFF, kProductDma, kStoresBufNt, true, false, kGuideNNPix>(a, s, gn)
  HDRNET_CASE(3, 3, true);
  HDRNET_CASE(3, 3, false);
  HDRNET_CASE(1, 1, true);
  HDRNET_CASE(1, 1, false);
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

// Slice-apply (+ optional fused guide network) + bilinear up-add of the coarser pyramid level:
// Cin = Cout = 3 with offset (the reference's pyramid model).
bool apply_fwd_seg_upadd_supported(const ApplyArgs& a, const float* coarse, bool guide_nn) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset) || ((uintptr_t)coarse & 3u) || ((uintptr_t)a.grid & 15u))
    return false;
  return seg_geom(a, true, !guide_nn).ok;
}

hipError_t launch_apply_fwd_seg_upadd(const ApplyArgs& a, const float* coarse, int Hc, int Wc, const float* conv1,
                                      const float* conv2, int n_feats, hipStream_t s, const char** name) {
  const UpAdd up{coarse, Hc, Wc, resize_scale(Hc, a.H), resize_scale(Wc, a.W)};
  if (conv1) {
    *name = "apply_fwd_seg/vec4+nnguide+upadd";
    return launch_seg_t<3, 3, true, kProductDma, kStoresBufNt, true, true, kGuideNNPix>(
        a, s, GuideNN{conv1, conv2, nullptr, n_feats, a.fast_sigmoid, a.guide_prescaled}, up);
  }
  *name = "apply_fwd_seg/vec4+upadd";
  return launch_seg_t<3, 3, true, kProductDma, kStoresBufNt, false, true, kProductPix>(
      a, s, GuideNN{nullptr, nullptr, nullptr, 0}, up);
}

}  // namespace hdrnet_amd
