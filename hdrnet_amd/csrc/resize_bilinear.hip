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
#include <hip/hip_runtime.h>

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
