// BilateralSliceApply forward for gfx950, second generation ("segment" kernel).
//
// Reference semantics: hdrnet/ops/bilateral_slice_apply.cc:24-82 (CUDA twin:
// bilateral_slice_apply.cu.cc:36-126).  Same decomposition as apply_fwd_rows.hip -- a workgroup
// owns a segment of an image row, the two grid rows the image row needs are blended once into an
// LDS image, a thread owns 4 consecutive pixels -- with the per-pixel instruction stream cut
// roughly in half and the memory side decoupled from it:
//
//   * PADDED LDS image.  Column j of the image is grid column clamp(cmin + j, 0, GW-1) and carries
//     GD + 2 z-planes, plane p = grid plane clamp(p - 1, 0, GD-1).  The reference clamps INDICES
//     but not weights (bilateral_slice_apply.cc:58-68); with the clamped copies materialised once
//     per workgroup, a pixel's four coefficient vectors sit at a00, a00 + CB, a00 + colb,
//     a00 + colb + CB -- one address, immediate offsets, no per-pixel min / max.
//   * the z-corner chain (offsets, squares, 1 - sqrt) runs on 2-wide vectors (v_pk_add / v_pk_fma /
//     v_pk_mul), both corners at once.
//   * R > 1: the workgroup keeps its segment for R consecutive rows.  Everything that depends on
//     x only (gxf, floor, both x weights, the column byte offset) is computed once; row r + 1's
//     pixel loads and grid-row loads are issued BEFORE row r is sliced (gfx9 vmcnt is in order
//     and the staging loads are issued ahead of the pixel loads, so waiting for either never
//     drains the younger prefetch), and the LDS image is double-buffered: one barrier per row.
//   * pixel loads: per-lane 16-B (LOADS 0), nontemporal lane-contiguous + LDS transpose (1), or
//     LDS-DMA `global_load_lds_dwordx4` straight into the wave's slab, plain / nontemporal (2 / 3).
//   * 3-D launch grid (segment, row block, batch): no integer division in the kernel.
//
// Numerics: identical expressions to apply_fwd_rows.hip (rows_common.hip.h: slice_terms) for the
// coordinates and weights; only the index clamps moved into the image.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

constexpr int kLoadsLane = 0;      // per-lane 16-B loads of the lane's own 4 pixels
constexpr int kLoadsNtContig = 1;  // nontemporal, lane-contiguous, transposed through the slab
constexpr int kLoadsDma = 2;       // LDS-DMA (global_load_lds_dwordx4), default cache policy
constexpr int kLoadsDmaNt = 3;     // LDS-DMA, nontemporal

// Output stores (all lane-contiguous 16 B after the per-wave LDS transpose):
constexpr int kStoresGlobal = 0;  // global_store_dwordx4, predicated on the run length
constexpr int kStoresBuf = 1;     // buffer_store_dwordx4 on a per-row-segment descriptor (out-of-range lanes dropped by the bounds check)
constexpr int kStoresBufNt = 2;   // ... nt
constexpr int kStoresBufSc1 = 3;  // ... sc1 (write-through: the line does not stay dirty in the XCD's L2)
constexpr int kStoresBufSc01 = 4; // ... sc0 sc1

constexpr int kStageMax = 2;  // staging elements (16 B or 4 B) a thread may hold in registers

struct SegParams {
  const float* grid;
  const float* guide;
  const float* input;
  float* out;
  int H, W, GH, GW, GD;
  int seg;         // pixels per segment, multiple of 4
  int img_floats;  // floats per image buffer (16-B multiple)
  int slab_off;    // float offset of the per-wave slabs in dynamic LDS
  float scale_x, scale_y;
  float inv_col;   // 1 / (GD * C / VEC): column of a staging element by float multiply
  long long* trace;  // TRACE: [nblocks][3] wall-clock ticks (start, end), XCC id; else unused
};

typedef int v4i32 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor over `bytes` bytes at `base` (gfx9 family: dword 3 = 0x00020000).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int STORES>
__device__ __forceinline__ void buf_store16(float4 v, __amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  constexpr int aux = STORES == kStoresBufNt ? 2 : STORES == kStoresBufSc1 ? 16 : STORES == kStoresBufSc01 ? 17 : 0;
  const v4i32 d = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, aux);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at `src` land at `dst_wave_base + 16 l`.
template <bool NT>
__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst_wave_base, 16, 0, NT ? 2 : 0);
}

// x-only terms of a pixel (bilateral_slice_apply.cc:41,46,53-54,61-62).
struct XTerm {
  float wx0, wx1;
  int xbp;  // byte offset of (column gx0, plane 0 + 1) in the image
};

__device__ __forceinline__ XTerm x_term(float xf, float scale_x, int cmin, int colb, int cb) {
#pragma clang fp contract(off)
  XTerm t;
  const float gxf = mul_rn(xf, scale_x);
  const float fxl = floorf(gxf - 0.5f);
  const float dx0 = (fxl + 0.5f) - gxf;  // in (-1, 0]
  t.wx0 = 1.0f + dx0;
  t.wx1 = -dx0;
  t.xbp = __mul24((int)fxl - cmin, colb) + cb;
  return t;
}

// One pixel: z terms, the four-vector blend from the padded image, the affine.
template <int CIN, int COUT, bool OFFSET>
__device__ __forceinline__ void seg_pixel(const float* __restrict__ img, float gd_f, float zhi, int colb,
                                          const XTerm& xt, float g, const float (&in)[CIN],
                                          float (&out)[COUT]) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  f32x2 w0, w1;
  int a0;
  {
#pragma clang fp contract(off)
    const float gzf = mul_rn(g, gd_f);
    const float fzl = floorf(gzf - 0.5f);
    const f32x2 cz = {fzl + 0.5f, fzl + 1.5f};
    const f32x2 gz2 = {gzf, gzf};
    const f32x2 dz = cz - gz2;  // (gz0 + .5) - gzf, (gz0 + 1.5) - gzf
    const f32x2 eps2 = {kSmoothEps, kSmoothEps};
    const f32x2 q = __builtin_elementwise_fma(dz, dz, eps2);
    const f32x2 s = {__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y)};
    const f32x2 one2 = {1.0f, 1.0f};
    const f32x2 wz = one2 - s;
    const f32x2 wx0 = {xt.wx0, xt.wx0}, wx1 = {xt.wx1, xt.wx1};
    w0 = wx0 * wz;
    w1 = wx1 * wz;
    // plane of z index iz is iz + 1; the clamp to [-1, GD-1] only guards wild guides (the
    // padded planes already hold the reference's clamped reads).
    const int iz = (int)__builtin_amdgcn_fmed3f(fzl, -1.0f, zhi);
    a0 = __mul24(iz, CB) + xt.xbp;
  }
  CoefVec<C> coef;
  accum_vec<C, true>(coef, img, a0, w0.x);
  accum_vec<C, false>(coef, img, a0 + CB, w0.y);
  accum_vec<C, false>(coef, img, a0 + colb, w1.x);
  accum_vec<C, false>(coef, img, a0 + colb + CB, w1.y);
#pragma unroll
  for (int i = 0; i < COUT; ++i) {
    float v = OFFSET ? coef.get(i * CJ + CIN) : 0.0f;
#pragma unroll
    for (int j = 0; j < CIN; ++j) v = fmaf(coef.get(i * CJ + j), in[j], v);
    out[i] = v;
  }
}

// Staging of one image row's blended, padded grid image.  `issue` loads this thread's elements of
// the two grid rows into registers; `write` blends and stores them (plus the clamped edge planes).
template <int C>
struct Stager {
  static constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  static constexpr int CV = C / VEC;
  typedef float elem_t __attribute__((ext_vector_type(VEC)));
  elem_t a[kStageMax], b[kStageMax];
  float wy0, wy1;

  // Per-thread decode of its staging elements, x-only (hoisted over rows).
  struct Map {
    int src[kStageMax];   // element offset inside a grid row (units of VEC floats)
    int dst[kStageMax];   // element offset inside the image; < 0: no element
    int edge[kStageMax];  // -1: also plane 0, +1: also plane GD + 1, 0: neither (2: both, GD == 1)
    int n;                // staging elements of the workgroup; element i of a thread exists iff
                          // tid + i * nthreads < n, and the whole slot iff i * nthreads < n (uniform)
    int nthreads;
  };

  static __device__ __forceinline__ Map make_map(int tid, int nthreads, int ncols, int cmin, int GW,
                                                 int GD, float inv_col) {
    Map m;
    const int per_col = GD * CV;
    const int n = ncols * per_col;
    m.n = n;
    m.nthreads = nthreads;
#pragma unroll
    for (int i = 0; i < kStageMax; ++i) {
      m.dst[i] = -1;
      if (i * nthreads >= n) continue;  // wave-uniform: the slot is empty for the whole workgroup
      const int e = tid + i * nthreads;
      const int j = (int)(((float)e + 0.5f) * inv_col);  // e / per_col, exact for e < 2^20
      const int rem = e - j * per_col;
      const int sc = min(max(cmin + j, 0), GW - 1);
      m.src[i] = sc * per_col + rem;
      m.dst[i] = (e < n) ? e + CV * (2 * j + 1) : -1;
      const bool lo = rem < CV, hi = rem >= per_col - CV;
      m.edge[i] = (lo && hi) ? 2 : (lo ? -1 : (hi ? 1 : 0));
    }
    return m;
  }

  __device__ __forceinline__ void issue(const Map& m, const float* __restrict__ grid_b, int y, int GH,
                                        int GW, int GD, float scale_y) {
    // Wave-uniform y terms (bilateral_slice_apply.cc:42,47,55-56).
    const float gyf = mul_rn(y + 0.5f, scale_y);
    const int gy0 = floor_to_int(gyf - 0.5f);
    wy0 = tent_weight(gy0 + 0.5f, gyf);
    wy1 = tent_weight(gy0 + 1 + 0.5f, gyf);
    const int gy0c = clamp_index(gy0, 0, GH - 1);
    const int gy1c = clamp_index(gy0 + 1, 0, GH - 1);
    const elem_t* r0 = reinterpret_cast<const elem_t*>(grid_b + (size_t)gy0c * GW * GD * C);
    const elem_t* r1 = reinterpret_cast<const elem_t*>(grid_b + (size_t)gy1c * GW * GD * C);
#pragma unroll
    for (int i = 0; i < kStageMax; ++i) {
      if (i * m.nthreads >= m.n) continue;
      if (m.dst[i] >= 0) {
        a[i] = r0[m.src[i]];
        b[i] = r1[m.src[i]];
      }
    }
  }

  __device__ __forceinline__ void write(const Map& m, float* __restrict__ img) const {
    elem_t* d = reinterpret_cast<elem_t*>(img);
#pragma unroll
    for (int i = 0; i < kStageMax; ++i) {
      if (i * m.nthreads >= m.n) continue;
      if (m.dst[i] >= 0) {
        const elem_t v = wy0 * a[i] + wy1 * b[i];
        d[m.dst[i]] = v;
        if (m.edge[i] == -1 || m.edge[i] == 2) d[m.dst[i] - CV] = v;
        if (m.edge[i] == 1 || m.edge[i] == 2) d[m.dst[i] + CV] = v;
      }
    }
  }
};

template <int CIN>
struct PixLoads {
  float4 g;
  float4 iv[CIN];
};

template <int CIN, int COUT, bool OFFSET, int R, int LOADS, int STORES, bool TRACE>
__global__ __launch_bounds__(256) void apply_fwd_seg(const SegParams p) {
  constexpr int CJ = CIN + (OFFSET ? 1 : 0);
  constexpr int C = COUT * CJ;
  constexpr int CB = C * (int)sizeof(float);
  constexpr int SLABW = 64 * kPxPerThread * (CIN > COUT ? CIN : COUT);  // floats: in / out run of a wave
  constexpr bool DMA = LOADS >= kLoadsDma;
  constexpr int SLAB = SLABW + (DMA ? 64 * kPxPerThread : 0);           // + the guide run
  static_assert(!DMA || R == 1, "LDS-DMA loads: single-row workgroups only");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  long long t_start = 0;
  if constexpr (TRACE) t_start = wall_clock64();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR)
  const int xs = blockIdx.x * p.seg;
  const int xe = min(xs + p.seg, p.W);
  const int y0 = blockIdx.y * R;
  const int b = blockIdx.z;
  const int nrows = min(R, p.H - y0);
  const float* grid_b = p.grid + (size_t)b * p.GH * p.GW * p.GD * C;

  const int x = xs + kPxPerThread * tid;
  const bool active = x < xe;
  const int wave_x0 = xs + kPxPerThread * 64 * wave;
  const int wave_px = min(xe, wave_x0 + 64 * kPxPerThread) - wave_x0;  // <= 0: idle wave
  float4* slab = reinterpret_cast<float4*>(lds + p.slab_off + wave * SLAB);
  float4* gslab = slab + SLABW / 4;  // DMA only

  // Grid columns of this segment, unclamped: gx0 of the first pixel .. gx0 + 1 of the last.
  const int cmin = floor_to_int(mul_rn(xs + 0.5f, p.scale_x) - 0.5f);
  const int cmax = floor_to_int(mul_rn(xe - 1 + 0.5f, p.scale_x) - 0.5f) + 1;
  const int ncols = cmax - cmin + 1;
  const int colb = (p.GD + 2) * CB;
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);

  // Addressing: a wave-uniform 64-bit segment base (SGPRs) + a 32-bit per-lane element offset,
  // so that loads / stores take the `saddr + voffset` form without 64-bit VALU arithmetic.
  const size_t row0 = (size_t)b * p.H + y0;
  const unsigned lpx = kPxPerThread * (unsigned)tid;         // this lane's first pixel in the segment
  const unsigned wpx = kPxPerThread * 64u * (unsigned)wave;  // this wave's first pixel in the segment
  auto issue_pix = [&](size_t row, PixLoads<CIN>& L) {
    const float* gseg = p.guide + (row * p.W + xs);        // uniform
    const float* iseg = p.input + (row * p.W + xs) * CIN;  // uniform
    if constexpr (LOADS == kLoadsLane) {
      if (active) {
        L.g = *reinterpret_cast<const float4*>(gseg + lpx);
#pragma unroll
        for (int q = 0; q < CIN; ++q) L.iv[q] = *reinterpret_cast<const float4*>(iseg + (lpx * CIN + 4 * q));
      }
    } else if constexpr (LOADS == kLoadsNtContig) {
      if (active) L.g = load_stream4(gseg + lpx);
#pragma unroll
      for (int k = 0; k < CIN; ++k) {
        const int e = lane + 64 * k;
        if (e < wave_px * CIN / 4) L.iv[k] = load_stream4(iseg + (wpx * CIN + 4u * (unsigned)e));
      }
    } else {
      // LDS-DMA: every lane issues (the LDS side is base + 16 * lane); lanes past the run re-read
      // its last float4.  An idle wave issues nothing.
      if (wave_px > 0) {
        dma16<LOADS == kLoadsDmaNt>(gseg + (wpx + 4u * (unsigned)min(lane, wave_px / 4 - 1)),
                                    reinterpret_cast<float*>(gslab));
        const int last = wave_px * CIN / 4 - 1;
#pragma unroll
        for (int k = 0; k < CIN; ++k)
          dma16<LOADS == kLoadsDmaNt>(iseg + (wpx * CIN + 4u * (unsigned)min(lane + 64 * k, last)),
                                      reinterpret_cast<float*>(slab + 64 * k));
      }
    }
  };

  // Row 0's pixel loads go out first: their HBM latency overlaps the (L2-resident) staging.
  PixLoads<CIN> cur;
  cur.g = make_float4(0.f, 0.f, 0.f, 0.f);
  Stager<C> st;
  const typename Stager<C>::Map smap =
      Stager<C>::make_map(tid, (int)blockDim.x, ncols, cmin, p.GW, p.GD, p.inv_col);
  if constexpr (DMA) {
    issue_pix(row0, cur);
    st.issue(smap, grid_b, y0, p.GH, p.GW, p.GD, p.scale_y);
  } else {
    // staging loads first: waiting for them (vmcnt is in order) then leaves the pixel loads in flight
    st.issue(smap, grid_b, y0, p.GH, p.GW, p.GD, p.scale_y);
    issue_pix(row0, cur);
  }

  // x-only terms of this thread's 4 pixels
  XTerm xt[kPxPerThread];
  const float xf0 = (float)x + 0.5f;  // (x + k) + 0.5f == xf0 + k exactly (x < 2^23)
#pragma unroll
  for (int k = 0; k < kPxPerThread; ++k) xt[k] = x_term(xf0 + (float)k, p.scale_x, cmin, colb, CB);

  st.write(smap, lds);
  __syncthreads();

#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < nrows) {
      const float* img = lds + (r & 1) * p.img_floats;
      const size_t row = row0 + r;
      PixLoads<CIN> nxt;
      nxt.g = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool more = (r + 1 < R) && (r + 1 < nrows);
      if constexpr (R > 1) {
        if (more) {
          st.issue(smap, grid_b, y0 + r + 1, p.GH, p.GW, p.GD, p.scale_y);
          issue_pix(row + 1, nxt);
        }
      }

      // ---- this lane's 4 pixels of row r --------------------------------------------------------
      float4 g4 = cur.g;
      float4 iv[CIN];
      if constexpr (LOADS == kLoadsLane) {
#pragma unroll
        for (int q = 0; q < CIN; ++q) iv[q] = cur.iv[q];
      } else if constexpr (LOADS == kLoadsNtContig) {
#pragma unroll
        for (int k = 0; k < CIN; ++k) {
          const int e = lane + 64 * k;
          if (e < wave_px * CIN / 4) slab[e] = cur.iv[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (active) {
#pragma unroll
          for (int q = 0; q < CIN; ++q) iv[q] = slab[lane * CIN + q];
        }
        __builtin_amdgcn_wave_barrier();  // the slab is reused for the output below
      } else {
        if (active) {
          g4 = gslab[lane];
#pragma unroll
          for (int q = 0; q < CIN; ++q) iv[q] = slab[lane * CIN + q];
        }
      }

      const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
      const float* inf = reinterpret_cast<const float*>(iv);
      float4 ov[COUT];
      float* of = reinterpret_cast<float*>(ov);
      if (active) {
#pragma unroll
        for (int k = 0; k < kPxPerThread; ++k) {
          float in[CIN], o[COUT];
#pragma unroll
          for (int j = 0; j < CIN; ++j) in[j] = inf[k * CIN + j];
          seg_pixel<CIN, COUT, OFFSET>(img, gd_f, zhi, colb, xt[k], gs[k], in, o);
#pragma unroll
          for (int i = 0; i < COUT; ++i) of[k * COUT + i] = o[i];
        }
        // CIN == COUT: a lane reads and writes only ITS slab entries here -- in place, no cross-lane
        // hazard; otherwise the input reads of all lanes must precede the output writes.
        if constexpr (LOADS != kLoadsLane && CIN != COUT) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < COUT; ++q) slab[lane * COUT + q] = ov[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      {
        float* oseg = p.out + (row * p.W + xs) * COUT;  // uniform
        if constexpr (STORES == kStoresGlobal) {
          if (wave_px >= 64 * kPxPerThread) {  // full wave (uniform): unpredicated stores
#pragma unroll
            for (int k = 0; k < COUT; ++k)
              *reinterpret_cast<float4*>(oseg + (wpx * COUT + 4u * (unsigned)(lane + 64 * k))) = slab[lane + 64 * k];
          } else {
            const int nvalid = wave_px * COUT / 4;  // float4s
#pragma unroll
            for (int k = 0; k < COUT; ++k) {
              const int e = lane + 64 * k;
              if (e < nvalid) *reinterpret_cast<float4*>(oseg + (wpx * COUT + 4u * (unsigned)e)) = slab[e];
            }
          }
        } else {
          // descriptor over exactly this row segment: lanes past the run are dropped by the bounds check
          const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(oseg, (unsigned)(xe - xs) * COUT * 4u);
#pragma unroll
          for (int k = 0; k < COUT; ++k)
            buf_store16<STORES>(slab[lane + 64 * k], orsrc, (wpx * COUT + 4u * (unsigned)(lane + 64 * k)) * 4u);
        }
      }

      if constexpr (R > 1) {
        if (more) {
          st.write(smap, lds + ((r + 1) & 1) * p.img_floats);
          __syncthreads();
          cur = nxt;
        }
      }
    }
  }

  if constexpr (TRACE) {
    if (tid == 0) {
      const size_t bid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      p.trace[3 * bid] = t_start;
      p.trace[3 * bid + 1] = wall_clock64();
      p.trace[3 * bid + 2] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf;  // HW_REG_XCC_ID
    }
  }
}

struct SegGeom {
  Plan pl;
  int max_cols, img_floats, slab_off;
  size_t lds;
  bool ok;
};

template <int CIN, int COUT, bool OFFSET>
SegGeom seg_geom(const ApplyArgs& a, int R, int loads) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  SegGeom g{};
  const bool aligned =
      (((uintptr_t)a.guide | (uintptr_t)a.input | (uintptr_t)a.out | (uintptr_t)a.grid) & 15u) == 0;
  g.pl = make_row_plan(a.W, a.GW, aligned);
  g.max_cols = (int)(((long long)(g.pl.seg - 1) * a.GW) / a.W + 4);
  g.img_floats = round_up(g.max_cols * (a.GD + 2) * C, 4);
  g.slab_off = (R > 1 ? 2 : 1) * g.img_floats;
  const int slabw = 64 * kPxPerThread * (CIN > COUT ? CIN : COUT) + (loads >= kLoadsDma ? 64 * kPxPerThread : 0);
  g.lds = ((size_t)g.slab_off + (size_t)(g.pl.threads / 64) * slabw) * sizeof(float);
  const long long nstage = (long long)g.max_cols * a.GD * (C / VEC);
  g.ok = g.pl.vec4 && g.lds <= 64 * 1024 && nstage <= (long long)kStageMax * g.pl.threads &&
         nstage < (1 << 20) && a.B <= 65535 && (a.H + R - 1) / R <= 65535 &&
         // a row block may span at most two grid rows ... not required: every row restages.
         true;
  return g;
}

template <int CIN, int COUT, bool OFFSET, int R, int LOADS, int STORES, bool TRACE>
hipError_t launch_seg_t(const ApplyArgs& a, hipStream_t s, long long* trace) {
  constexpr int C = COUT * (CIN + (OFFSET ? 1 : 0));
  constexpr int VEC = (C % 4 == 0) ? 4 : 1;
  const SegGeom g = seg_geom<CIN, COUT, OFFSET>(a, R, LOADS);
  if (!g.ok) return hipErrorNotSupported;
  SegParams p;
  p.grid = a.grid;
  p.guide = a.guide;
  p.input = a.input;
  p.out = a.out;
  p.H = a.H;
  p.W = a.W;
  p.GH = a.GH;
  p.GW = a.GW;
  p.GD = a.GD;
  p.seg = g.pl.seg;
  p.img_floats = g.img_floats;
  p.slab_off = g.slab_off;
  p.scale_x = (float)a.GW / a.W;
  p.scale_y = (float)a.GH / a.H;
  p.inv_col = 1.0f / (float)(a.GD * (C / VEC));
  p.trace = trace;
  const dim3 grid3((unsigned)g.pl.nseg, (unsigned)((a.H + R - 1) / R), (unsigned)a.B);
  apply_fwd_seg<CIN, COUT, OFFSET, R, LOADS, STORES, TRACE><<<grid3, g.pl.threads, g.lds, s>>>(p);
  return hipGetLastError();
}

long long* g_trace = nullptr;  // tools only: device buffer of [nblocks][3]

}  // namespace

void apply_fwd_seg_set_trace(long long* device_buf) { g_trace = device_buf; }

// Tools build: knob = variant - 20.
//   0..19  R = 1: loads = knob % 4 {lane, nt-contig, dma, dma-nt}, stores = knob / 4 {global, buf, buf-nt, buf-sc1, buf-sc0sc1}
//   20..39 the same with the timeline trace
//   40..43 R = 2 {lane, nt-contig}, R = 4 {lane, nt-contig}; 44..47 traced
hipError_t launch_apply_fwd_seg(const ApplyArgs& a, int knob, hipStream_t s, const char** name) {
  if (!(a.Cin == 3 && a.Cout == 3 && a.has_offset)) return hipErrorNotSupported;
  static char namebuf[64];
  static const char* const lname[] = {"lane", "ntcontig", "dma", "dma-nt"};
  static const char* const sname[] = {"", "+bufst", "+bufst-nt", "+bufst-sc1", "+bufst-sc0sc1"};
  if (knob < 0 || knob >= 48) return hipErrorNotSupported;
  if (knob < 40) {
    const bool trace = knob >= 20;
    const int k = knob % 20, L = k % 4, S = k / 4;
    if (trace && !g_trace) return hipErrorInvalidValue;
    snprintf(namebuf, sizeof namebuf, "apply_fwd_seg/R1-%s%s", lname[L], sname[S]);
    *name = namebuf;
#define SEG_CASE(LL, SS)                                                              \
  if (L == LL && S == SS)                                                             \
    return trace ? launch_seg_t<3, 3, true, 1, LL, SS, true>(a, s, g_trace)           \
                 : launch_seg_t<3, 3, true, 1, LL, SS, false>(a, s, nullptr)
    SEG_CASE(0, 0); SEG_CASE(1, 0); SEG_CASE(2, 0); SEG_CASE(3, 0);
    SEG_CASE(0, 1); SEG_CASE(1, 1); SEG_CASE(2, 1); SEG_CASE(3, 1);
    SEG_CASE(0, 2); SEG_CASE(1, 2); SEG_CASE(2, 2); SEG_CASE(3, 2);
    SEG_CASE(0, 3); SEG_CASE(1, 3); SEG_CASE(2, 3); SEG_CASE(3, 3);
    SEG_CASE(0, 4); SEG_CASE(1, 4); SEG_CASE(2, 4); SEG_CASE(3, 4);
#undef SEG_CASE
    return hipErrorNotSupported;
  }
  const bool trace = knob >= 44;
  const int k = (knob - 40) % 4;
  if (trace && !g_trace) return hipErrorInvalidValue;
  snprintf(namebuf, sizeof namebuf, "apply_fwd_seg/R%d-%s", k < 2 ? 2 : 4, lname[k & 1]);
  *name = namebuf;
#define SEG_CASE(K, RR, LL)                                                           \
  if (k == K)                                                                         \
    return trace ? launch_seg_t<3, 3, true, RR, LL, 0, true>(a, s, g_trace)           \
                 : launch_seg_t<3, 3, true, RR, LL, 0, false>(a, s, nullptr)
  SEG_CASE(0, 2, 0);
  SEG_CASE(1, 2, 1);
  SEG_CASE(2, 4, 0);
  SEG_CASE(3, 4, 1);
#undef SEG_CASE
  return hipErrorNotSupported;
}

}  // namespace hdrnet_amd
