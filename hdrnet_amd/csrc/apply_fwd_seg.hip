// BilateralSliceApply forward for gfx950 -- the north-star kernel (second generation).
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
//     z gather bank-conflict-free.
//   * the z-corner chain (offsets, squares, 1 - sqrt) runs on 2-wide vectors, both corners at
//     once; the blend is 24 v_pk_fma_f32 / v_pk_mul_f32 per pixel.
//   * PIXEL LOADS are LDS-DMA (`global_load_lds_dwordx4 ... nt`): each wave streams its 256-pixel
//     run (1 KiB of guide, Cin KiB of input) lane-contiguously straight into its LDS slab with the
//     nontemporal policy, no VGPRs held while in flight; a lane then reads its own 4 pixels back
//     with ds_read_b128.  Nontemporal loads lower the no-compute floor of this byte mix from
//     40.8 to 39.4 us per 4K frame, but only as dense per-instruction runs (a per-pixel 48-B
//     stride re-fetches lines).
//   * STORES leave through the same slab, transposed to lane-contiguous 16-B runs, as
//     `buffer_store_dwordx4` with a streaming cache policy on a descriptor that covers exactly the
//     row segment (lanes past the run are dropped by the bounds check -- no predicate).  Plain stores
//     leave up to an L2's worth of dirty lines for the end-of-kernel write-back; streaming them out is
//     worth 1.5 us per 4K frame and 0.7 us per 1080p frame.  Write-through (sc0 sc1) is the fastest
//     where segments are whole 128-B lines and 10 % worse than `nt` where they are not, so the
//     flavour is chosen per launch (launch_apply_fwd_seg below; rows_common.hip.h; profiles/r02/).
//   * 3-D launch grid (segment, row, batch): no integer division in the kernel.
//   * ROUND 3 -- a shorter pixel phase (it, not the memory system, is what slows down in the power management's
//     slow state -- at an unchanged REPORTED clock: profiles/r03/power/summary.txt; the kernel runs at the
//     1400-W package cap, 1.74-1.9 GHz): the LEAN per-pixel code of seg_common.hip.h
//     (v_fract x weight, float byte addresses, clamp-modifier tents), the pixel runs fetched as
//     `buffer_load_dwordx4 ... lds` (one per-lane offset register for all four 1-KiB pieces, the run's end
//     enforced by the descriptor instead of four clamps + 64-bit address adds), the segment's grid-column
//     window from a host-side table in the kernel arguments, 32-bit row arithmetic.  40.4 -> 39.4 us interleaved
//     on two boxes, the no-compute skeleton at 39.0 (profiles/r03/ab_variants_4k_*.txt).  The BLEND the product
//     ships is the SCALAR one (kPixLeanScalar: 48 v_fma_f32 per pixel, 419 VALU + 113 SALU instructions per wave;
//     the packed form, 24 v_pk_fma_f32, is 323 + 114): same time on steady boxes, bit-identical results, less time in
//     the power manager's braked state on the boxes that fall into it (launch_apply_fwd_seg below).  The guide-network
//     / wire-format kernels, which are VALU-bound, keep the packed blend.
//   * ROUND 4 -- the launch shape (profiles/r04/fwd_launch_shape.md).  ONE thing changed: the resident waves per CU are
//     CAPPED (resident_cap_lds below: 7 three-wave workgroups per CU instead of the 9 the LDS footprint allows) -- the
//     same 38-39.5 us per 4K launch on steady boxes, and the end of the power manager's 42-46-us state on the boxes that
//     had it (profiles/r04/power_state/).  Everything else was measured and left as it is: fewer resident
//     workgroups shorten a workgroup's life (9 -> 6 per CU: 5.1 -> 4.0 us) and leave the launch where it was (39.4 vs
//     39.7 us): it is throughput-, not latency-bound; flat 1024-px tasks that ignore rows (1080p: 2025 workgroups for
//     2048 slots) do not beat row segments in a no-compute skeleton (11.5 vs 11.3 us); a ticketed tail (SCHED = 1 below:
//     the last ~2000 tasks drawn from counters so that the XCDs finish together) narrows the XCDs' finish from 3.2 to
//     1.5 us but nets 0.3 us (0.8 %) at 4K and 1.1 us (1.8 %) at 4000x3000 -- tools variants 70 / 71, not the product.
//     Per launch ~2.5-3 us lie outside the workgroups' span (38.8 us per launch, 35.9 us from the first workgroup's
//     start to the last one's end): the gap between dependent kernels of one stream.
//
// Numerics: the coordinate and weight expressions of the reference in the reference's order
// (products (x+.5)*scale_x, guide*GD explicitly rounded, see numerics.hip.h: mul_rn); wy is folded
// into the LDS image, the x weights come as wx1 = fract(gxf - .5) and w(x0, .) = wz - wz * wx1 (exact
// but for 1 ulp in the first half cell), max(., 0) of the z tent is the clamp of the subtraction,
// v_sqrt_f32 (1 ulp) stands in for sqrtf; differences stay at the 1e-7 level (tests/test_gpu_parity.py
// and test_gpu_fullsize.py hold rtol = atol = 1e-5 against the oracle and report the reference's own
// 1e-6 bar: worst / bar <= 0.25 at every config size).
//
// The TOOLS build (HDRNET_TOOLS_BUILD) also instantiates the load / store flavours this design
// was chosen against (per-lane loads, nontemporal lane-contiguous register loads, plain DMA; plain /
// nt / sc1 stores) and a per-workgroup timeline trace; tools/ab_bench.py times them interleaved.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
#include "seg_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

constexpr int kLoadsLane = 0;      // per-lane 16-B loads of the lane's own 4 pixels
constexpr int kLoadsNtContig = 1;  // nontemporal, lane-contiguous, transposed through the slab
constexpr int kLoadsDma = 2;       // LDS-DMA (global_load_lds_dwordx4), default cache policy
constexpr int kLoadsDmaNt = 3;     // LDS-DMA, nontemporal                         <- product (round 2)
constexpr int kLoadsBufDmaNt = 4;  // LDS-DMA, nontemporal, as buffer_load ... lds: ONE per-lane offset register for
                                   // all pieces and the run's end enforced by the descriptor (no per-piece
                                   // clamp + 64-bit address arithmetic)

// Pixel phase: 0 = round 2's (seg_pixel), 1 = lean with the packed blend, 2 = lean with a scalar blend
// (seg_common.hip.h).
constexpr int kPixR02 = 0, kPixLean = 1, kPixLeanScalar = 2, kPixLeanAllScalar = 3;  // 3: z taps scalar too

// Output stores (all lane-contiguous 16 B after the per-wave LDS transpose):
constexpr int kStoresGlobal = 0;   // global_store_dwordx4, predicated on the run length
constexpr int kStoresBufNt = 2;    // buffer_store_dwordx4 ... nt                  <- product
constexpr int kStoresBufSc1 = 3;   // ... sc1 (write-through: the line does not stay dirty in L2)
constexpr int kStoresBufSc01 = 4;  // ... sc0 sc1
// (1 = buffer_store_dwordx4 with the default policy)

// ---- ticketed tail (SCHED = 1) ----------------------------------------------------------------------------
// The launch grid is flat: workgroups 0 .. S-1 run task = blockIdx.x; the E workgroups behind them draw ONE
// ticket each for the remaining D tasks (E > D: an XCD that is ahead reaches its ticketed workgroups earlier and
// takes more than an eighth of them; a workgroup whose ticket is past its pool's tasks exits).
//
// Device-scope atomics are the scarce resource here: one word serves ~60-90 returning atomics per us on this
// fabric and the requests queued on it hold up the memory channel they sit on (round 4, gpurun_out/r04a: a
// `done` word incremented by every ticketed workgroup stretched a 39-us launch to 55-95 us).  So: kPools
// counter words, each on a line of its own 4.25 KiB apart; a ticketed workgroup draws exactly ONCE, from the pool
// of its position (the 8 consecutive workgroups the dispatcher spreads over the 8 XCDs share a pool, the next 8
// take the next pool: every pool is drawn by all XCDs alike); dynamic task j belongs to pool j % kPools.  Every
// pool receives exactly E / kPools draws per launch -- the workgroup that draws the last one zeroes the word: a
// launch finds the words zero and leaves them zero, with no second counter.
// CONTRACT: two launches that use the same counter block must not execute concurrently.
constexpr int kPools = 32;
constexpr int kPoolStride = 1088;  // unsigned per pool (4352 B)
constexpr unsigned kNoTask = 0xffffffffu;
constexpr size_t kSchedWords = (size_t)kPools * kPoolStride;

struct DynSched {
  unsigned* ctr;  // [kPools][kPoolStride]: word 0 of pool p = its tickets
  unsigned S, D, E;  // E % (8 * kPools) == 0
  unsigned nseg, nseg_magic, h_magic;  // t / nseg = umulhi(t, nseg_magic), r / H = umulhi(r, h_magic)
};

__device__ __forceinline__ unsigned dyn_draw(const DynSched& d, unsigned i) {
  const unsigned pool = (i >> 3) & (kPools - 1);
  unsigned* w = d.ctr + pool * kPoolStride;
  const unsigned per = (d.D + kPools - 1 - pool) / kPools;  // tasks j < D with j % kPools == pool
  const unsigned t = __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t == d.E / kPools - 1) __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the pool's last draw
  return t < per ? d.S + t * kPools + pool : kNoTask;
}

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
  long long* trace;  // TRACE: [nblocks][3] wall-clock ticks (start, end), XCC id; else unused
  GuideNN gn;        // GUIDE_NN: the folded point-wise guide network (rows_common.hip.h)
  UpAdd up;          // UPADD: the coarser pyramid level to up-sample and add
  DynSched dyn;      // SCHED != 0: flat launch grid with a ticketed tail (below)
#ifdef HDRNET_TOOLS_BUILD
  int stagger, stagger_rows;  // experiment (knob 7): workgroups of the first rows sleep (index % 8) * stagger x 64
                              // cycles before their pixel phase -- de-synchronises the launch's first round
#endif
};

template <int STORES>
constexpr int store_aux() {
  return STORES == kStoresBufNt ? kAuxNt : STORES == kStoresBufSc1 ? kAuxSc1 : STORES == kStoresBufSc01 ? kAuxSc0Sc1 : kAuxPlain;
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at `src` land at `dst_wave_base + 16 l`.
template <bool NT>
__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst_wave_base, 16, 0, NT ? 2 : 0);
}

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
template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool TRACE, bool GUIDE_NN, bool UPADD, int PIX>
__device__ __forceinline__ void seg_task(const SegParams& p, float* __restrict__ lds, const int segi, const int y,
                                         const int b, [[maybe_unused]] const size_t trace_slot) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  constexpr int SLABW = 64 * kPxPerThread * (CIN > COUT ? CIN : COUT);  // floats: in / out run of a wave
  constexpr bool DMA = LOADS >= kLoadsDma;
  constexpr int SLAB = SLABW + ((DMA && !GUIDE_NN) ? 64 * kPxPerThread : 0);  // + the guide run

  long long t_start = 0;
  if constexpr (TRACE) t_start = wall_clock64();

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
  } else if constexpr (LOADS == kLoadsNtContig) {
    if constexpr (!GUIDE_NN) {
      if (active) g4 = load_stream4(gseg + lpx);
    }
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const int e = lane + 64 * k;
      if (e < wave_px * CIN / 4) iv[k] = load_stream4(iseg + (wpx * CIN + 4u * (unsigned)e));
    }
  } else if constexpr (LOADS == kLoadsBufDmaNt) {
    // LDS-DMA through buffer descriptors that end with the wave's run: every piece uses the same
    // per-lane offset (16 * lane) + an immediate; lanes past the run read zeros (never used).
    if (wave_px > 0) {
      const unsigned voff = 16u * (unsigned)lane;
      if constexpr (!GUIDE_NN) {
        const __amdgpu_buffer_rsrc_t grs = make_rsrc(gseg + wpx, (unsigned)wave_px * 4u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lptr_t) reinterpret_cast<float*>(gslab), 16, voff, 0, 0, kAuxNt);
      }
      const __amdgpu_buffer_rsrc_t irs = make_rsrc(iseg + wpx * CIN, (unsigned)wave_px * (4u * CIN));
      buf_dma_pieces<0, CIN>(irs, reinterpret_cast<float*>(slab), voff);
    }
  } else {
    // LDS-DMA: every lane issues (the LDS side is base + 16 * lane); lanes past the run re-read
    // its last float4.  An idle wave issues nothing.
    if (wave_px > 0) {
      if constexpr (!GUIDE_NN)
        dma16<LOADS == kLoadsDmaNt>(gseg + (wpx + 4u * (unsigned)min(lane, wave_px / 4 - 1)),
                                    reinterpret_cast<float*>(gslab));
      const int last = wave_px * CIN / 4 - 1;
#pragma unroll
      for (int k = 0; k < CIN; ++k)
        dma16<LOADS == kLoadsDmaNt>(iseg + (wpx * CIN + 4u * (unsigned)min(lane + 64 * k, last)),
                                    reinterpret_cast<float*>(slab + 64 * k));
    }
  }

  stage_image<C>(lds, grid_b, y + p.y0, cmin, ncols, p.GH, p.GW, p.GD, p.scale_y, p.inv_col, tid, (int)blockDim.x);

  // x-only terms of this thread's 4 pixels
  constexpr bool LEAN = PIX != kPixR02;
  XTerm xt[LEAN ? 1 : kPxPerThread];
  XTermLean xl[LEAN ? kPxPerThread : 1];
  const float xf0 = (float)x + 0.5f;  // (x + k) + 0.5f == xf0 + k exactly (x < 2^23)
  if constexpr (LEAN) {
    const float colb_f = (float)colb, xbase_f = (float)(CB - cmin * colb);  // exact: |.| < 2^24
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) xl[k] = x_term_lean(xf0 + (float)k, p.scale_x, colb_f, xbase_f);
  } else {
#pragma unroll
    for (int k = 0; k < kPxPerThread; ++k) xt[k] = x_term(xf0 + (float)k, p.scale_x, cmin, colb, CB);
  }

  __syncthreads();  // image complete; with LDS-DMA in flight the compiler drains vmcnt here too
#ifdef HDRNET_TOOLS_BUILD
  if (p.stagger > 0 && y < p.stagger_rows) {  // uniform
    const int n = ((y * 5 + segi) & 7) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
#endif

  // ---- this lane's 4 pixels ---------------------------------------------------------------------
  if constexpr (LOADS == kLoadsNtContig) {
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
  } else if constexpr (DMA) {
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
      if constexpr (LEAN)
        seg_pixel_lean<CIN, COUT, OFFSET, PIX == kPixLean, PIX != kPixLeanAllScalar>(lds, gd_f, zhi, colb, xl[k], gs[k], in, o);
      else
        seg_pixel<CIN, COUT, OFFSET>(lds, gd_f, zhi, colb, xt[k], gs[k], in, o);
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
  if constexpr (STORES == kStoresGlobal) {
    const int nvalid = wave_px * COUT / 4;  // float4s
#pragma unroll
    for (int k = 0; k < COUT; ++k) {
      const int e = lane + 64 * k;
      if (e < nvalid) *reinterpret_cast<float4*>(oseg + (wpx * COUT + 4u * (unsigned)e)) = slab[e];
    }
  } else {
    // descriptor over exactly this row segment: lanes past the run are dropped by the bounds check
    const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(oseg, (unsigned)(xe - xs) * COUT * 4u);
#pragma unroll
    for (int k = 0; k < COUT; ++k)
      buf_store16<store_aux<STORES>()>(slab[lane + 64 * k], orsrc, (wpx * COUT + 4u * (unsigned)(lane + 64 * k)) * 4u);
  }

  if constexpr (TRACE) {
    if (tid == 0) {
      p.trace[3 * trace_slot] = t_start;
      p.trace[3 * trace_slot + 1] = wall_clock64();
      p.trace[3 * trace_slot + 2] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf;  // HW_REG_XCC_ID
    }
  }
}

// SCHED = 0: the 3-D launch grid (segment, row, image) -- one task per workgroup, no index arithmetic.
// SCHED = 1: a FLAT grid whose last workgroups take their task from ticket counters (DynSched): the XCDs of a
//            launch do not finish together (1350 workgroups each at 4K, ends 2-3.4 us apart: the last 1 % of the
//            workgroups retires over the last 2 us of a 39-us launch, profiles/r03/ab_variants_4k.txt), and a
//            statically assigned tail cannot move to the XCDs that are ahead.
template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool TRACE, bool GUIDE_NN = false,
          bool UPADD = false, int PIX = kPixR02, int SCHED = 0>
__global__ __launch_bounds__(256) void apply_fwd_seg(const SegParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if constexpr (SCHED == 0) {
    seg_task<CIN, COUT, OFFSET, LOADS, STORES, TRACE, GUIDE_NN, UPADD, PIX>(
        p, lds, blockIdx.x, blockIdx.y, blockIdx.z,
        ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  } else {
    __shared__ unsigned s_task;
    const DynSched& d = p.dyn;
    unsigned t = blockIdx.x;
    const bool dynamic = t >= d.S;  // uniform
    if (dynamic) {
      if (threadIdx.x == 0) s_task = dyn_draw(d, t - d.S);
      __syncthreads();
      t = s_task;
    }
    if (t != kNoTask) {
      const unsigned r = __umulhi(t, d.nseg_magic);  // t / nseg: row index over all images
      const unsigned segi = t - r * d.nseg;
      const unsigned b = __umulhi(r, d.h_magic);     // r / H
      const unsigned y = r - b * (unsigned)p.H;
      seg_task<CIN, COUT, OFFSET, LOADS, STORES, TRACE, GUIDE_NN, UPADD, PIX>(p, lds, (int)segi, (int)y, (int)b, t);
    }
  }
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

#ifdef HDRNET_TOOLS_BUILD
// Experiment knobs of the tools build (hdrnet_tools_set_knob): 0 extra dynamic LDS bytes per workgroup (caps the
// workgroups per CU), 1 D = ticketed tasks, 2 surplus workgroups E - D.
int g_knob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
unsigned* g_sched_words = nullptr;  // kSchedWords zeroed words, allocated on first use
#endif

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

inline unsigned div_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d) + 1u; }

// Flat-grid schedule of M tasks with a ticketed tail of D tasks served by E workgroups.  False: the task count is
// out of the magic division's exact range (the caller launches the 3-D grid instead).
inline bool make_dyn_sched(DynSched& d, unsigned* words, long long M, int nseg, int H, long long D, long long E) {
  const long long dmax = nseg > H ? nseg : H;
  if (M <= 0 || M * dmax >= (1ll << 32) || M + E >= (1ll << 31) || nseg < 2 || H < 2) return false;
  if (D > M) D = M;
  E = (E + 8 * kPools - 1) / (8 * kPools) * (8 * kPools);
  if (D == 0) E = 0;
  d.ctr = words;
  d.S = (unsigned)(M - D);
  d.D = (unsigned)D;
  d.E = (unsigned)E;
  d.nseg = (unsigned)nseg;
  d.nseg_magic = div_magic((unsigned)nseg);
  d.h_magic = div_magic((unsigned)H);
  return true;
}

template <int CIN, int COUT, bool OFFSET, int LOADS, int STORES, bool TRACE, bool GUIDE_NN = false,
          bool UPADD = false, int PIX = kPixR02, int SCHED = 0>
hipError_t launch_seg_t(const ApplyArgs& a, hipStream_t s, long long* trace,
                        const GuideNN& gn = GuideNN{nullptr, nullptr, nullptr, 0},
                        const UpAdd& up = UpAdd{nullptr, 0, 0, 0.f, 0.f}, const DynSched* dyn = nullptr) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  SegGeom g = seg_geom(a, LOADS >= kLoadsDma, !GUIDE_NN);
  if (!g.ok) return hipErrorNotSupported;
  // (not for the per-lane-load flavour, which serves grids of about ONE round of workgroups -- a single 1080p frame:
  //  there every resident slot counts, 11.7 us uncapped / 12.0 at 7 workgroups per CU / 13.1 at 6;
  //  nor with the guide network fused: that kernel is VALU-bound and wants its waves, 44.8 -> 48.2 us capped at 4K)
  if constexpr (LOADS >= kLoadsDma && !GUIDE_NN) g.lds = resident_cap_lds(g.lds, g.pl.threads);
#ifdef HDRNET_TOOLS_BUILD
  if (g_knob[0] < 0) g.lds = seg_geom(a, LOADS >= kLoadsDma, !GUIDE_NN).lds;  // knob 0 < 0: no cap (round 3's residency)
  else g.lds += (size_t)g_knob[0];
#endif
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
  p.trace = trace;
#ifdef HDRNET_TOOLS_BUILD
  p.stagger = g_knob[7];
  p.stagger_rows = (2304 + g.pl.nseg - 1) / g.pl.nseg;  // the first round of resident workgroups
#endif
  if constexpr (SCHED != 0) {
    if (!dyn) return hipErrorInvalidValue;
    p.dyn = *dyn;
    apply_fwd_seg<CIN, COUT, OFFSET, LOADS, STORES, TRACE, GUIDE_NN, UPADD, PIX, SCHED>
        <<<dim3(p.dyn.S + p.dyn.E), g.pl.threads, g.lds, s>>>(p);
  } else {
    p.dyn = DynSched{};
    const dim3 grid3((unsigned)g.pl.nseg, (unsigned)a.H, (unsigned)a.B);
    apply_fwd_seg<CIN, COUT, OFFSET, LOADS, STORES, TRACE, GUIDE_NN, UPADD, PIX><<<grid3, g.pl.threads, g.lds, s>>>(p);
  }
  return hipGetLastError();
}

bool seg_shape(const ApplyArgs& a) { return apply_fast_shape(a.Cin, a.Cout, a.has_offset); }

#ifdef HDRNET_TOOLS_BUILD
long long* g_trace = nullptr;  // device buffer of [nblocks][3]
#endif

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

// The per-launch flavour choice for one shape, with the pixel phase PIX and the DMA form DMAL as parameters
// (the tools build times the alternatives against each other, launch_apply_fwd_seg_pix).
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
    if (small) return launch_seg_t<CI, CO, OFF, kLoadsLane, kStoresBufNt, false, false, false, PIX>(a, s, nullptr, gn, up);
    if (whole_lines)
      return launch_seg_t<CI, CO, OFF, DMAL, kStoresBufSc01, false, false, false, PIX>(a, s, nullptr, gn, up);
  }
  return launch_seg_t<CI, CO, OFF, DMAL, kStoresBufNt, false, false, false, PIX>(a, s, nullptr, gn, up);
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
  const GuideNN gn{conv1, conv2, guide_out, n_feats, a.fast_sigmoid};
  *name = "apply_fwd_seg/vec4+nnguide";
#define HDRNET_CASE(CI, CO, OFF)                          \
  if (a.Cin == CI && a.Cout == CO && a.has_offset == OFF) \
    return launch_seg_t<CI, CO, OFF, kProductDma, kStoresBufNt, false, true, false, kGuideNNPix>(a, s, nullptr, gn)
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
    return launch_seg_t<3, 3, true, kProductDma, kStoresBufNt, false, true, true, kGuideNNPix>(
        a, s, nullptr, GuideNN{conv1, conv2, nullptr, n_feats, a.fast_sigmoid}, up);
  }
  *name = "apply_fwd_seg/vec4+upadd";
  return launch_seg_t<3, 3, true, kProductDma, kStoresBufNt, false, false, true, kProductPix>(
      a, s, nullptr, GuideNN{nullptr, nullptr, nullptr, 0}, up);
}

#ifdef HDRNET_TOOLS_BUILD
void apply_fwd_seg_set_trace(long long* device_buf) { g_trace = device_buf; }
void apply_fwd_seg_set_knob(int idx, int value) {
  if (idx >= 0 && idx < 8) g_knob[idx] = value;
}
int tools_knob(int idx) { return idx >= 0 && idx < 8 ? g_knob[idx] : 0; }

// variants 70 / 71 (71: with the timeline trace): the product flavour on the flat grid with a ticketed tail;
// knobs 1, 2 = D, surplus.
hipError_t launch_apply_fwd_seg_dyn(const ApplyArgs& a, bool trace, hipStream_t s, const char** name) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return hipErrorNotSupported;
  if (trace && !g_trace) return hipErrorInvalidValue;
  if (!g_sched_words) {
    if (hipMalloc(&g_sched_words, kSchedWords * sizeof(unsigned)) != hipSuccess) return hipErrorOutOfMemory;
    if (hipMemset(g_sched_words, 0, kSchedWords * sizeof(unsigned)) != hipSuccess) return hipErrorUnknown;
  }
  const SegGeom g = seg_geom(a, true);
  if (!g.ok) return hipErrorNotSupported;
  const long long M = (long long)g.pl.nseg * a.H * a.B;
  const long long D = g_knob[1] < M ? g_knob[1] : M;
  DynSched d;
  if (!make_dyn_sched(d, g_sched_words, M, g.pl.nseg, a.H, D, D + g_knob[2])) return hipErrorNotSupported;
  const GuideNN gn{nullptr, nullptr, nullptr, 0};
  const UpAdd up{nullptr, 0, 0, 0.f, 0.f};
  static char namebuf[96];
  snprintf(namebuf, sizeof namebuf, "apply_fwd_seg/dyn D=%u E=%u", d.D, d.E);
  *name = namebuf;
  const bool whole_lines = ((uintptr_t)a.out % 128 == 0) && ((long long)g.pl.seg * a.Cout * 4) % 128 == 0 &&
                           ((long long)a.W * a.Cout * 4) % 128 == 0;
  if (trace) {
    return whole_lines ? launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufSc01, true, false, false, kPixLeanScalar, 1>(a, s, g_trace, gn, up, &d)
                       : launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufNt, true, false, false, kPixLeanScalar, 1>(a, s, g_trace, gn, up, &d);
  }
  return whole_lines ? launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufSc01, false, false, false, kPixLeanScalar, 1>(a, s, nullptr, gn, up, &d)
                     : launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufNt, false, false, false, kPixLeanScalar, 1>(a, s, nullptr, gn, up, &d);
}

// variant 72: the product kernel (3-D grid) with the timeline trace -- the static twin of 71.
hipError_t launch_apply_fwd_seg_product_trace(const ApplyArgs& a, hipStream_t s, const char** name) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset) || !g_trace) return hipErrorNotSupported;
  const SegGeom g = seg_geom(a, true);
  const bool whole_lines = ((uintptr_t)a.out % 128 == 0) && ((long long)g.pl.seg * a.Cout * 4) % 128 == 0 &&
                           ((long long)a.W * a.Cout * 4) % 128 == 0;
  *name = "apply_fwd_seg/product+trace";
  return whole_lines ? launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufSc01, true, false, false, kPixLeanScalar>(a, s, g_trace)
                     : launch_seg_t<3, 3, true, kLoadsBufDmaNt, kStoresBufNt, true, false, false, kPixLeanScalar>(a, s, g_trace);
}

// knob = variant - 60: pixel phase = knob % 4 {0 round 2, 1 lean packed, 2 lean scalar blend}; + 4: the pixel
// loads as buffer_load ... lds instead of global_load ... lds.  Load / store flavour per launch as the product.
hipError_t launch_apply_fwd_seg_pix(const ApplyArgs& a, int knob, hipStream_t s, const char** name) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return hipErrorNotSupported;
  static const char* const nm[8] = {"apply_fwd_seg/pix0", "apply_fwd_seg/pix-lean", "apply_fwd_seg/pix-lean-scalar",
                                    "apply_fwd_seg/pix-lean-allscalar", "apply_fwd_seg/pix0+bufdma",
                                    "apply_fwd_seg/pix-lean+bufdma", "apply_fwd_seg/pix-lean-scalar+bufdma",
                                    "apply_fwd_seg/pix-lean-allscalar+bufdma"};
  if (knob < 0 || knob >= 8) return hipErrorNotSupported;
  *name = nm[knob];
  switch (knob) {
    case 0: return launch_seg_pick<3, 3, true, kPixR02, kLoadsDmaNt>(a, s);
    case 1: return launch_seg_pick<3, 3, true, kPixLean, kLoadsDmaNt>(a, s);
    case 2: return launch_seg_pick<3, 3, true, kPixLeanScalar, kLoadsDmaNt>(a, s);
    case 3: return launch_seg_pick<3, 3, true, kPixLeanAllScalar, kLoadsDmaNt>(a, s);
    case 4: return launch_seg_pick<3, 3, true, kPixR02, kLoadsBufDmaNt>(a, s);
    case 5: return launch_seg_pick<3, 3, true, kPixLean, kLoadsBufDmaNt>(a, s);
    case 6: return launch_seg_pick<3, 3, true, kPixLeanScalar, kLoadsBufDmaNt>(a, s);
    case 7: return launch_seg_pick<3, 3, true, kPixLeanAllScalar, kLoadsBufDmaNt>(a, s);
  }
  return hipErrorNotSupported;
}

// knob = variant - 20:  0..19 loads = knob % 4 {lane, nt-contig, dma, dma-nt}, stores = knob / 4
// {global, buffer, buffer nt, buffer sc1, buffer sc0 sc1};  20..39 the same with the timeline trace.
hipError_t launch_apply_fwd_seg_knob(const ApplyArgs& a, int knob, hipStream_t s, const char** name) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return hipErrorNotSupported;
  static char namebuf[64];
  static const char* const lname[] = {"lane", "ntcontig", "dma", "dma-nt"};
  static const char* const sname[] = {"", "+bufst", "+bufst-nt", "+bufst-sc1", "+bufst-sc0sc1"};
  if (knob < 0 || knob >= 40) return hipErrorNotSupported;
  const bool trace = knob >= 20;
  const int k = knob % 20, L = k % 4, S = k / 4;
  if (trace && !g_trace) return hipErrorInvalidValue;
  snprintf(namebuf, sizeof namebuf, "apply_fwd_seg/%s%s", lname[L], sname[S]);
  *name = namebuf;
#define SEG_CASE(LL, SS)                                                       \
  if (L == LL && S == SS)                                                      \
    return trace ? launch_seg_t<3, 3, true, LL, SS, true>(a, s, g_trace)       \
                 : launch_seg_t<3, 3, true, LL, SS, false>(a, s, nullptr)
  SEG_CASE(0, 0); SEG_CASE(1, 0); SEG_CASE(2, 0); SEG_CASE(3, 0);
  SEG_CASE(0, 1); SEG_CASE(1, 1); SEG_CASE(2, 1); SEG_CASE(3, 1);
  SEG_CASE(0, 2); SEG_CASE(1, 2); SEG_CASE(2, 2); SEG_CASE(3, 2);
  SEG_CASE(0, 3); SEG_CASE(1, 3); SEG_CASE(2, 3); SEG_CASE(3, 3);
  SEG_CASE(0, 4); SEG_CASE(1, 4); SEG_CASE(2, 4); SEG_CASE(3, 4);
#undef SEG_CASE
  return hipErrorNotSupported;
}
#endif  // HDRNET_TOOLS_BUILD

}  // namespace hdrnet_amd
