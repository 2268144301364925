// NHWC bilinear resize with align_corners = True: the `tf.image.resize_images(level, sz,
// BILINEAR, align_corners=True)` that builds HDRNetGaussianPyrNN's multi-scale input
// (hdrnet/models.py:253-266) and up-samples its coarse outputs (:283-286).
//
// TensorFlow is a dependency of the reference that is not vendored (hdrnet/requirements.txt:
// tensorflow_gpu==2.12.0); its published algorithm (tensorflow/core/kernels/image/
// resize_bilinear_op.cc, legacy path used when half_pixel_centers = false) is restated here:
//   scale = (in - 1) / float(out - 1)  (in / float(out) when out == 1)
//   src = i * scale;  lower = floor(src);  upper = min(ceil(src), in - 1);  lerp = src - lower
//   out = top + (bottom - top) * y_lerp,  top = tl + (tr - tl) * x_lerp
// One thread per output pixel; a pure gather, HBM-bound (reads every source pixel once when
// down-sampling by two: 4 * C * (Hin*Win + Hout*Wout) bytes).
//
// Training the pyramid model also needs the up-ADD of hdrnet/models.py:283-287 (`current = resize(current) + out_lvl`) and
// its VJP; as torch ops they were 0.46 ms of a 1.32-ms graph-captured step (upsample 78 us, its backward 149 us + fills,
// adds, permute copies).  resize_add: the resize above plus the fine level in the same pass.  resize_bilinear_grad: the
// transpose as a GATHER -- thread = source pixel, which walks the destination pixels whose two taps per axis can touch it
// (the same float arithmetic as the forward decides the taps and weights) and sums in a fixed order: no atomics, so the
// gradient is bit-reproducible.  Entry points in include/hdrnet_amd_train.h.
#include <hip/hip_runtime.h>

#include "../../include/hdrnet_amd_train.h"
#include "launch.hip.h"
#include "numerics.hip.h"

namespace hdrnet_amd {
namespace {

template <int C>
__global__ __launch_bounds__(256) void resize_bilinear_ac(const float* __restrict__ in,
                                                          float* __restrict__ out, int Hin, int Win,
                                                          int Hout, int Wout, int Cdyn, float sh,
                                                          float sw, long long npx) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npx) return;
  const int x = (int)(p % Wout);
  const int y = (int)((p / Wout) % Hout);
  const long long b = p / ((long long)Wout * Hout);
  const int nc = C > 0 ? C : Cdyn;
  const float sy = mul_rn((float)y, sh), sx = mul_rn((float)x, sw);
  const float fy = floorf(sy), fx = floorf(sx);
  const float ly = sy - fy, lx = sx - fx;
  const int y0 = (int)fy, y1 = min((int)ceilf(sy), Hin - 1);
  const int x0 = (int)fx, x1 = min((int)ceilf(sx), Win - 1);
  const float* r0 = in + ((size_t)b * Hin + y0) * Win * nc;
  const float* r1 = in + ((size_t)b * Hin + y1) * Win * nc;
  float* o = out + (size_t)p * nc;
#pragma unroll
  for (int c = 0; c < nc; ++c) {
    const float tl = r0[x0 * nc + c], tr = r0[x1 * nc + c];
    const float bl = r1[x0 * nc + c], br = r1[x1 * nc + c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    o[c] = top + (bot - top) * ly;
  }
}

// out[b, y, x, :] = resize(coarse)[b, y, x, :] + fine[b, y, x, :]
template <int C>
__global__ __launch_bounds__(256) void resize_add_ac(const float* __restrict__ in, const float* __restrict__ fine,
                                                     float* __restrict__ out, int Hin, int Win, int Hout, int Wout,
                                                     int Cdyn, float sh, float sw, long long npx) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npx) return;
  const int x = (int)(p % Wout);
  const int y = (int)((p / Wout) % Hout);
  const long long b = p / ((long long)Wout * Hout);
  const int nc = C > 0 ? C : Cdyn;
  const float sy = mul_rn((float)y, sh), sx = mul_rn((float)x, sw);
  const float fy = floorf(sy), fx = floorf(sx);
  const float ly = sy - fy, lx = sx - fx;
  const int y0 = (int)fy, y1 = min((int)ceilf(sy), Hin - 1);
  const int x0 = (int)fx, x1 = min((int)ceilf(sx), Win - 1);
  const float* r0 = in + ((size_t)b * Hin + y0) * Win * nc;
  const float* r1 = in + ((size_t)b * Hin + y1) * Win * nc;
  const float* f = fine + (size_t)p * nc;
  float* o = out + (size_t)p * nc;
#pragma unroll
  for (int c = 0; c < nc; ++c) {
    const float tl = r0[x0 * nc + c], tr = r0[x1 * nc + c];
    const float bl = r1[x0 * nc + c], br = r1[x1 * nc + c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    o[c] = (top + (bot - top) * ly) + f[c];
  }
}

// The weight with which destination index i reads source index k along one axis of extent `nin`: the forward's
// out = a + (b - a) * l = a * (1 - l) + b * l up to rounding, a = in[floor(s)], b = in[min(ceil(s), nin - 1)].
__device__ __forceinline__ float tap_weight(int i, int k, float scale, int nin) {
  const float s = mul_rn((float)i, scale);
  const float f = floorf(s);
  const float l = s - f;
  const int k0 = (int)f, k1 = min((int)ceilf(s), nin - 1);
  return (k0 == k ? 1.0f - l : 0.0f) + (k1 == k ? l : 0.0f);
}

// din[b, ys, xs, :] = sum over destination pixels (y, x) of wy(y, ys) * wx(x, xs) * dout[b, y, x, :]
// (Hin x Win = the SOURCE of the forward, Hout x Wout its destination; rows then columns, ascending: a fixed order).
template <int C>
__global__ __launch_bounds__(256) void resize_bilinear_grad_ac(const float* __restrict__ dout, float* __restrict__ din,
                                                               int Hin, int Win, int Hout, int Wout, int Cdyn, float sh,
                                                               float sw, float inv_sh, float inv_sw, long long npx) {
  constexpr int kMaxC = C > 0 ? C : 4;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npx) return;
  const int xs = (int)(p % Win);
  const int ys = (int)((p / Win) % Hin);
  const long long b = p / ((long long)Win * Hin);
  const int nc = C > 0 ? C : Cdyn;
  // destination indices whose taps can be ys: floor(s) or ceil(s) == ys  =>  s in (ys - 1, ys + 1); one more on each side
  // for the rounding of the reciprocal
  // (clamped as floats: the reciprocal of a zero scale is huge)
  const int ylo = (int)fmaxf(floorf((float)(ys - 1) * inv_sh) - 1.0f, 0.0f);
  const int yhi = (int)fminf(ceilf((float)(ys + 1) * inv_sh) + 1.0f, (float)(Hout - 1));
  const int xlo = (int)fmaxf(floorf((float)(xs - 1) * inv_sw) - 1.0f, 0.0f);
  const int xhi = (int)fminf(ceilf((float)(xs + 1) * inv_sw) + 1.0f, (float)(Wout - 1));
  // the column weights do not depend on the row: kept for windows of <= 8 columns (up-sampling by two: <= 7)
  constexpr int kW = 8;
  float wxs[kW];
  const bool cached = xhi - xlo + 1 <= kW;
#pragma unroll
  for (int i = 0; i < kW; ++i) wxs[i] = (cached && xlo + i <= xhi) ? tap_weight(xlo + i, xs, sw, Win) : 0.0f;
  for (int c0 = 0; c0 < nc; c0 += kMaxC) {  // (one round unless C is dynamic and > 4)
    float acc[kMaxC];
#pragma unroll
    for (int c = 0; c < kMaxC; ++c) acc[c] = 0.0f;
    for (int y = ylo; y <= yhi; ++y) {
      const float wy = tap_weight(y, ys, sh, Hin);
      if (wy == 0.0f) continue;
      const float* row = dout + ((size_t)b * Hout + y) * Wout * nc;
      if (cached) {
#pragma unroll
        for (int i = 0; i < kW; ++i) {
          const float w = wy * wxs[i];
          if (w != 0.0f) {  // (wxs[i] == 0 beyond the window)
#pragma unroll
            for (int c = 0; c < kMaxC; ++c)
              if (c0 + c < nc) acc[c] = __builtin_fmaf(w, row[(size_t)(xlo + i) * nc + c0 + c], acc[c]);
          }
        }
      } else {
        for (int x = xlo; x <= xhi; ++x) {
          const float w = wy * tap_weight(x, xs, sw, Win);
          if (w == 0.0f) continue;
#pragma unroll
          for (int c = 0; c < kMaxC; ++c)
            if (c0 + c < nc) acc[c] = __builtin_fmaf(w, row[(size_t)x * nc + c0 + c], acc[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxC; ++c)
      if (c0 + c < nc) din[(size_t)p * nc + c0 + c] = acc[c];
  }
}

}  // namespace

hipError_t launch_resize_bilinear(const float* in, float* out, int B, int Hin, int Win, int Hout,
                                  int Wout, int C, hipStream_t s, const char** name) {
  const long long npx = (long long)B * Hout * Wout;
  const long long nblocks = (npx + 255) / 256;
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : (float)Hin / (float)Hout;
  const float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : (float)Win / (float)Wout;
  *name = "resize_bilinear_ac";
  if (C == 3)
    resize_bilinear_ac<3><<<(unsigned)nblocks, 256, 0, s>>>(in, out, Hin, Win, Hout, Wout, C, sh, sw, npx);
  else if (C == 1)
    resize_bilinear_ac<1><<<(unsigned)nblocks, 256, 0, s>>>(in, out, Hin, Win, Hout, Wout, C, sh, sw, npx);
  else
    resize_bilinear_ac<0><<<(unsigned)nblocks, 256, 0, s>>>(in, out, Hin, Win, Hout, Wout, C, sh, sw, npx);
  return hipGetLastError();
}

}  // namespace hdrnet_amd

extern "C" int hdrnet_resize_add_f32(const float* coarse, const float* fine, float* output, int batch, int in_height,
                                     int in_width, int out_height, int out_width, int channels, void* stream) {
  using namespace hdrnet_amd;
  if (!coarse || !fine || !output || batch <= 0 || in_height <= 0 || in_width <= 0 || out_height <= 0 || out_width <= 0 ||
      channels <= 0)
    return 1;
  const long long npx = (long long)batch * out_height * out_width;
  const long long nblocks = (npx + 255) / 256;
  if (nblocks > 0x7fffffffLL) return 1;
  const float sh = out_height > 1 ? (float)(in_height - 1) / (float)(out_height - 1) : (float)in_height / (float)out_height;
  const float sw = out_width > 1 ? (float)(in_width - 1) / (float)(out_width - 1) : (float)in_width / (float)out_width;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels == 3)
    resize_add_ac<3><<<(unsigned)nblocks, 256, 0, s>>>(coarse, fine, output, in_height, in_width, out_height, out_width,
                                                       channels, sh, sw, npx);
  else
    resize_add_ac<0><<<(unsigned)nblocks, 256, 0, s>>>(coarse, fine, output, in_height, in_width, out_height, out_width,
                                                       channels, sh, sw, npx);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int hdrnet_resize_bilinear_grad_f32(const float* doutput, float* dinput, int batch, int in_height, int in_width,
                                               int out_height, int out_width, int channels, void* stream) {
  using namespace hdrnet_amd;
  if (!doutput || !dinput || batch <= 0 || in_height <= 0 || in_width <= 0 || out_height <= 0 || out_width <= 0 ||
      channels <= 0)
    return 1;
  const long long npx = (long long)batch * in_height * in_width;
  const long long nblocks = (npx + 255) / 256;
  if (nblocks > 0x7fffffffLL) return 1;
  const float sh = out_height > 1 ? (float)(in_height - 1) / (float)(out_height - 1) : (float)in_height / (float)out_height;
  const float sw = out_width > 1 ? (float)(in_width - 1) / (float)(out_width - 1) : (float)in_width / (float)out_width;
  // a source index k is touched by destination indices within (k - 1, k + 1) / scale; scale 0 (one source row): all of them
  const float inv_sh = sh > 0.0f ? 1.0f / sh : 3.0e38f, inv_sw = sw > 0.0f ? 1.0f / sw : 3.0e38f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (channels == 3)
    resize_bilinear_grad_ac<3><<<(unsigned)nblocks, 256, 0, s>>>(doutput, dinput, in_height, in_width, out_height, out_width,
                                                                 channels, sh, sw, inv_sh, inv_sw, npx);
  else
    resize_bilinear_grad_ac<0><<<(unsigned)nblocks, 256, 0, s>>>(doutput, dinput, in_height, in_width, out_height, out_width,
                                                                 channels, sh, sw, inv_sh, inv_sw, npx);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
