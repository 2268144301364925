// TOOLS BUILD ONLY: the multi-scale output of HDRNetGaussianPyrNN in ONE pass over the full-resolution frame.
//
// Reference semantics: HDRNetGaussianPyrNN._output, hdrnet/models.py:277-289 --
//   out = L0(in0) + up(L1(in1) + up(L2(in2))),  L_l = slice-apply of level l's grid with level l's guide network,
//   up = tf.image.resize_images(., BILINEAR, align_corners=True) -- the graph the model is trained with.
// (The reference's GL renderer, benchmark/assets/gpyrnn.frag:65-86, evaluates all three levels at every FULL-RESOLUTION
// pixel instead -- a different function of the coefficients: the slice is not linear in the guide.)
//
// The product computes it per level (apply_fwd_seg.hip UPADD: level 2, level 1 + up-add, level 0 + up-add: three
// launches with the two coarse results in memory).  VERDICT r03 / r04 asked for the single pass to be MEASURED rather
// than costed: a workgroup owns R = 4 full-resolution rows of one row segment and RE-EVALUATES the coarse pixels those
// rows tap -- the (<= 4) level-1 rows x (S / 2 + 2) pixels and the (<= 4) level-2 rows x (S / 4 + 2) pixels -- into LDS,
// then evaluates its own pixels and adds the two bilinear up-samplings from LDS: ~1.56 slice evaluations per output
// pixel (the per-level form: 1.31) and no intermediate level in memory.  The pixel core is the product's
// (seg_common.hip.h: stage_image, x_term_lean, seg_pixel_lean; rows_common.hip.h: guide_nn_quad); loads and stores are
// plain per-lane float4s (the product streams through LDS-DMA; this kernel is instruction-bound either way).
// Timed against the per-level chain by tools/pyramid_onepass_bench.py; result in profiles/r05/pyramid_onepass.md.
#include <hip/hip_runtime.h>

#include "../../include/hdrnet_amd_tools.h"
#include "launch.hip.h"
#include "numerics.hip.h"
#include "rows_common.hip.h"
#include "seg_common.hip.h"

namespace hdrnet_amd {
namespace {

using namespace rows;

constexpr int kR = 4;         // full-resolution rows per workgroup
constexpr int kLvRows = 4;    // rows of a coarser level a workgroup may need (checked on the host)
constexpr int kC = 12, kCB = kC * 4;
constexpr int kThreads = 256;

struct PyrLevel {
  const float* grid;  // [B][GH][GW][GD][12]
  const float* in;    // [B][H][W][3]
  GuideNN gn;
  int H, W;
  float scale_x, scale_y;  // GW / W, GH / H
};

struct PyrParams {
  PyrLevel lv[3];  // 0 = full resolution, 1 = half, 2 = quarter
  float* out;      // [B][H0][W0][3]
  int GH, GW, GD;
  int S;              // full-resolution pixels per segment (multiple of 16)
  float sh[2], sw[2];  // up-sampling scales: [0] level 1 -> 0, [1] level 2 -> 1  ((in - 1) / float(out - 1))
  int img_floats;     // LDS floats per coefficient image (largest level)
  int pitch1, pitch2;  // pixels per row of the two level buffers
  float inv_col;
  int grid_image;
};

// Window of coarse rows / columns that the fine indices [lo, hi] tap: lower = floor(i * s), upper = min(ceil(i * s), n - 1)
// (resize_bilinear_op.cc, legacy path) -- monotone in i.
__device__ __forceinline__ void tap_window(int lo, int hi, float s, int n, int* wlo, int* whi) {
  *wlo = (int)floorf(mul_rn((float)lo, s));
  *whi = min((int)ceilf(mul_rn((float)hi, s)), n - 1);
}

// One level's pixels [rows r_lo .. r_hi] x [c_lo, c_lo + ncol) (c_lo, ncol multiples of 4) as a flat list of
// (row, quad) items over the workgroup: guide network, slice, affine; `sink(ri, x, o[12])` takes the quad's result.
template <typename Sink>
__device__ __forceinline__ void eval_level(const PyrParams& p, const PyrLevel& L, int b, float* img, int r_lo, int nrows,
                                           int c_lo, int ncol, Sink sink) {
  const int tid = threadIdx.x;
  const float* grid_b = L.grid + (size_t)b * (unsigned)p.grid_image;
  const SegCols sc = seg_cols(c_lo, c_lo + ncol, L.scale_x);
  const int colb = (p.GD + 2) * kCB;
  for (int ri = 0; ri < nrows; ++ri)  // one y-pre-lerped coefficient image per row
    stage_image<kC>(img + ri * p.img_floats, grid_b, r_lo + ri, sc.cmin, sc.ncols, p.GH, p.GW, p.GD, L.scale_y,
                    p.inv_col, tid, kThreads);
  __syncthreads();
  const float gd_f = (float)p.GD, zhi = (float)(p.GD - 1);
  const float colb_f = (float)colb, xbase_f = (float)(kCB - sc.cmin * colb);
  const int nq = ncol / 4;
  for (int it = tid; it < nrows * nq; it += kThreads) {
    const int ri = it / nq, x = c_lo + 4 * (it - ri * nq);
    const float4* src = reinterpret_cast<const float4*>(L.in + (((size_t)b * L.H + (r_lo + ri)) * L.W + x) * 3);
    const float4 iv[3] = {src[0], src[1], src[2]};
    const float* inf = reinterpret_cast<const float*>(iv);
    float gs[4];
    guide_nn_quad<3>(L.gn, inf, gs);
    float o[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const XTermLean xl = x_term_lean((float)(x + k) + 0.5f, L.scale_x, colb_f, xbase_f);
      const float in[3] = {inf[3 * k], inf[3 * k + 1], inf[3 * k + 2]};
      float ok[3];
      seg_pixel_lean<3, 3, true, true>(img + ri * p.img_floats, gd_f, zhi, colb, xl, gs[k], in, ok);
      o[3 * k] = ok[0]; o[3 * k + 1] = ok[1]; o[3 * k + 2] = ok[2];
    }
    sink(ri, x, o);
  }
}

// o[3 k + i] += the coarser level's buffer `buf` ([row - r_lo][col - c_lo][3], `pitch` pixels per row), bilinearly
// up-sampled at fine pixel (x + k, y): the arithmetic of rows_common.hip.h upadd_quad.
__device__ __forceinline__ void upadd_from_lds(const float* buf, int pitch, int r_lo, int c_lo, int Hc, int Wc, float sh,
                                               float sw, int y, int x, float* o) {
  const float sy = mul_rn((float)y, sh);
  const float fy = floorf(sy);
  const float ly = sy - fy;
  const int y0 = (int)fy - r_lo, y1 = min((int)ceilf(sy), Hc - 1) - r_lo;
  const float* r0 = buf + y0 * pitch * 3;
  const float* r1 = buf + y1 * pitch * 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float sxf = mul_rn((float)(x + k), sw);
    const float fx = floorf(sxf);
    const float lx = sxf - fx;
    const int x0 = ((int)fx - c_lo) * 3, x1 = (min((int)ceilf(sxf), Wc - 1) - c_lo) * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float tl = r0[x0 + i], tr = r0[x1 + i], bl = r1[x0 + i], br = r1[x1 + i];
      const float top = tl + (tr - tl) * lx;
      const float bot = bl + (br - bl) * lx;
      o[3 * k + i] += top + (bot - top) * ly;
    }
  }
}

__global__ __launch_bounds__(kThreads) void pyramid_onepass_kernel(PyrParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* img = lds;                                        // [kLvRows][img_floats]
  float* buf1 = img + kLvRows * p.img_floats;              // level-1 results [kLvRows][pitch1][3]
  float* buf2 = buf1 + kLvRows * p.pitch1 * 3;             // level-2 results [kLvRows][pitch2][3]
  const int b = blockIdx.z;
  const PyrLevel &L0 = p.lv[0], &L1 = p.lv[1], &L2 = p.lv[2];
  const int xs = blockIdx.x * p.S, xe = min(xs + p.S, L0.W);
  const int y0 = blockIdx.y * kR, y1 = min(y0 + kR, L0.H);
  // the level-1 pixels these rows tap, columns widened to whole quads; then the level-2 pixels THOSE tap
  int r1_lo, r1_hi, c1_lo, c1_hi, r2_lo, r2_hi, c2_lo, c2_hi;
  tap_window(y0, y1 - 1, p.sh[0], L1.H, &r1_lo, &r1_hi);
  tap_window(xs, xe - 1, p.sw[0], L1.W, &c1_lo, &c1_hi);
  c1_lo &= ~3;
  const int n1 = (c1_hi - c1_lo + 4) & ~3;
  tap_window(r1_lo, r1_hi, p.sh[1], L2.H, &r2_lo, &r2_hi);
  tap_window(c1_lo, c1_lo + n1 - 1, p.sw[1], L2.W, &c2_lo, &c2_hi);
  c2_lo &= ~3;
  const int n2 = (c2_hi - c2_lo + 4) & ~3;

  // level 2 -> buf2
  eval_level(p, L2, b, img, r2_lo, r2_hi - r2_lo + 1, c2_lo, n2, [&](int ri, int x, const float* o) {
    float* d = buf2 + (ri * p.pitch2 + (x - c2_lo)) * 3;
#pragma unroll
    for (int e = 0; e < 12; ++e) d[e] = o[e];
  });
  __syncthreads();
  // level 1 + up(level 2) -> buf1
  eval_level(p, L1, b, img, r1_lo, r1_hi - r1_lo + 1, c1_lo, n1, [&](int ri, int x, float* o) {
    upadd_from_lds(buf2, p.pitch2, r2_lo, c2_lo, L2.H, L2.W, p.sh[1], p.sw[1], r1_lo + ri, x, o);
    float* d = buf1 + (ri * p.pitch1 + (x - c1_lo)) * 3;
#pragma unroll
    for (int e = 0; e < 12; ++e) d[e] = o[e];
  });
  __syncthreads();
  // level 0 + up(level 1) -> out
  eval_level(p, L0, b, img, y0, y1 - y0, xs, xe - xs, [&](int ri, int x, float* o) {
    upadd_from_lds(buf1, p.pitch1, r1_lo, c1_lo, L1.H, L1.W, p.sh[0], p.sw[0], y0 + ri, x, o);
    float4* dst = reinterpret_cast<float4*>(p.out + (((size_t)b * L0.H + (y0 + ri)) * L0.W + x) * 3);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
  });
}

// rows of the coarser level that `rows` consecutive fine rows starting anywhere can tap (host-side check)
int max_tap_rows(int fine, int coarse, int rows) {
  const float s = resize_scale(coarse, fine);
  int worst = 0;
  for (int y = 0; y < fine; y += 1) {
    const int hi_y = (y + rows - 1 < fine ? y + rows - 1 : fine - 1);
    volatile float a = (float)y * s, c = (float)hi_y * s;
    const int lo = (int)floorf(a);
    int hi = (int)ceilf(c);
    if (hi > coarse - 1) hi = coarse - 1;
    if (hi - lo + 1 > worst) worst = hi - lo + 1;
  }
  return worst;
}

}  // namespace
}  // namespace hdrnet_amd

extern "C" int hdrnet_tools_pyramid_onepass_f32(const float* const grids[3], const float* const inputs[3],
                                                const float* const conv1[3], const float* const conv2[3], int n_feats,
                                                float* out, int B, int H, int W, int GH, int GW, int GD, int seg,
                                                unsigned flags, void* stream) {
  using namespace hdrnet_amd;
  if (B <= 0 || H < 4 || W < 16 || (W % 16) || (H % 4) || GD > 8 || seg <= 0 || (seg % 16)) return HDRNET_INVALID_ARGUMENT;
  PyrParams p{};
  int h = H, w = W;
  for (int l = 0; l < 3; ++l) {
    p.lv[l] = PyrLevel{grids[l], inputs[l], GuideNN{conv1[l], conv2[l], nullptr, n_feats, (flags & HDRNET_GUIDE_SIGMOID_FAST) != 0},
                       h, w, (float)GW / w, (float)GH / h};
    h /= 2;
    w /= 2;
  }
  p.out = out;
  p.GH = GH; p.GW = GW; p.GD = GD;
  p.S = seg;
  for (int l = 0; l < 2; ++l) {
    p.sh[l] = rows::resize_scale(p.lv[l + 1].H, p.lv[l].H);
    p.sw[l] = rows::resize_scale(p.lv[l + 1].W, p.lv[l].W);
  }
  // every workgroup's coarse windows must fit kLvRows rows
  if (max_tap_rows(H, H / 2, kR) > kLvRows || max_tap_rows(H / 2, H / 4, kLvRows) > kLvRows) return HDRNET_INVALID_ARGUMENT;
  p.pitch1 = seg / 2 + 12;
  p.pitch2 = seg / 4 + 16;
  int max_cols = 0;
  for (int l = 0; l < 3; ++l) {
    const int span = l == 0 ? seg : (l == 1 ? p.pitch1 : p.pitch2);
    const int cols = (int)((double)span * GW / p.lv[l].W) + 3;
    if (cols > max_cols) max_cols = cols;
  }
  p.img_floats = max_cols * (GD + 2) * kC;
  p.inv_col = 1.0f / (float)(GD * (kC / 4));
  p.grid_image = GH * GW * GD * kC;
  const size_t lds_bytes = sizeof(float) * ((size_t)kLvRows * p.img_floats + (size_t)kLvRows * 3 * (p.pitch1 + p.pitch2));
  if (lds_bytes > 64 * 1024) return HDRNET_INVALID_ARGUMENT;
  const dim3 nblocks((unsigned)((W + seg - 1) / seg), (unsigned)(H / kR), (unsigned)B);
  hipLaunchKernelGGL(pyramid_onepass_kernel, nblocks, dim3(kThreads), lds_bytes, static_cast<hipStream_t>(stream), p);
  return hipGetLastError() == hipSuccess ? HDRNET_OK : HDRNET_RUNTIME_FAILURE;
}
