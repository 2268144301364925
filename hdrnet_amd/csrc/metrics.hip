// The training loop's l2 loss, hdrnet/metrics.py:8-11 (`tf.reduce_mean(tf.square(target - prediction))`), and its
// gradient with respect to the prediction, as two HBM-bound passes over the full-resolution batch:
//   forward   reads prediction and target (8 B per element), writes one partial sum per workgroup + the mean
//   backward  reads both again and writes d prediction = (2 / n) * grad_output * (prediction - target) (12 B per element)
// torch's mse_loss is five launches here -- the squared differences written out (and read back by the mean), a
// zeros_like of the gradient that the backward then overwrites -- 143 us at 4 x 1080p against ~85 us for these two.
// When the prediction wants a gradient the forward also writes the UNIT gradient (2 / n) * (prediction - target) -- it has
// both operands in registers -- and the backward only scales it by grad_output, which is 1 when the loss is the root of
// the backward pass: that kernel reads one scalar and returns (8 + 4 B per element instead of 8 + 12; include/hdrnet_amd_train.h).
//
// Also here: the optimizer update of that loop over ONE flat parameter buffer (include/hdrnet_amd_train.h).
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/hdrnet_amd_train.h"
#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

constexpr int kLossBlocks = 2048;
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool GRAD>
__global__ __launch_bounds__(256) void l2_loss_partial(const float* __restrict__ pred, const float* __restrict__ target,
                                                       long long n, float* __restrict__ partial,
                                                       float* __restrict__ dunit) {
  __shared__ float red[256];
  const long long n4 = n >> 2;
  const v4f* p4 = reinterpret_cast<const v4f*>(pred);
  const v4f* t4 = reinterpret_cast<const v4f*>(target);
  v4f* d4 = reinterpret_cast<v4f*>(dunit);
  const float k = (float)(2.0 / (double)n);
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const v4f a = __builtin_nontemporal_load(p4 + i), b = __builtin_nontemporal_load(t4 + i);
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    if constexpr (GRAD) {
      v4f d;
      d.x = k * dx, d.y = k * dy, d.z = k * dz, d.w = k * dw;
      d4[i] = d;  // plain store: the slice-apply's gradient kernels read it next
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // the tail of a length that is not a multiple of 4
    const float d = pred[(n4 << 2) + threadIdx.x] - target[(n4 << 2) + threadIdx.x];
    acc += d * d;
    if constexpr (GRAD) dunit[(n4 << 2) + threadIdx.x] = k * d;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void l2_loss_final(const float* __restrict__ partial, int nb, long long n,
                                                     float* __restrict__ loss) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

__global__ __launch_bounds__(256) void l2_loss_grad(const float* __restrict__ pred, const float* __restrict__ target,
                                                    const float* __restrict__ grad_output, long long n,
                                                    float* __restrict__ dpred) {
  const float k = grad_output[0] * (float)(2.0 / (double)n);
  const long long n4 = n >> 2;
  const v4f* p4 = reinterpret_cast<const v4f*>(pred);
  const v4f* t4 = reinterpret_cast<const v4f*>(target);
  v4f* d4 = reinterpret_cast<v4f*>(dpred);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const v4f a = __builtin_nontemporal_load(p4 + i), b = __builtin_nontemporal_load(t4 + i);
    d4[i] = k * (a - b);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    dpred[i] = k * (pred[i] - target[i]);
  }
}

// d *= grad_output, skipped altogether when grad_output == 1 (the loss is the root of the backward pass)
__global__ __launch_bounds__(256) void l2_loss_grad_scale(float* __restrict__ d, const float* __restrict__ grad_output,
                                                          long long n) {
  const float g = grad_output[0];
  if (g == 1.0f) return;
  const long long n4 = n >> 2;
  v4f* d4 = reinterpret_cast<v4f*>(d);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) d4[i] = g * d4[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) d[(n4 << 2) + threadIdx.x] *= g;
}

__global__ __launch_bounds__(256) void adam_flat(float* __restrict__ param, const float* __restrict__ grad,
                                                 float* __restrict__ m, float* __restrict__ v, long long n,
                                                 const float* __restrict__ step, float lr, float b1, float b2, float eps_in,
                                                 int eps_hat) {
  const float t = step[0] + 1.0f;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  // TensorFlow's form, lr sqrt(bc2) / bc1 * m / (sqrt(v) + eps), is this one with eps / sqrt(bc2) ("epsilon hat")
  const float eps = eps_hat ? eps_in * rs2 : eps_in;
  const long long n4 = n >> 2;
  v4f* p4 = reinterpret_cast<v4f*>(param);
  const v4f* g4 = reinterpret_cast<const v4f*>(grad);
  v4f* m4 = reinterpret_cast<v4f*>(m);
  v4f* v4 = reinterpret_cast<v4f*>(v);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const v4f g = g4[i];
    const v4f mm = b1 * m4[i] + (1.0f - b1) * g;
    const v4f vv = b2 * v4[i] + (1.0f - b2) * (g * g);
    m4[i] = mm;
    v4[i] = vv;
    v4f d;
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = mm[k] / (sqrtf(vv[k]) * rs2 + eps);
    p4[i] = p4[i] - step_size * d;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float g = grad[i];
    const float mm = b1 * m[i] + (1.0f - b1) * g, vv = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mm;
    v[i] = vv;
    param[i] -= step_size * mm / (sqrtf(vv) * rs2 + eps);
  }
}

__global__ void adam_count(float* step) { step[0] += 1.0f; }

}  // namespace

size_t l2_loss_workspace_bytes(long long n) { return n > 0 ? (size_t)kLossBlocks * sizeof(float) : 0; }

hipError_t launch_l2_loss(const float* pred, const float* target, long long n, float* loss, void* workspace,
                          hipStream_t s) {
  float* partial = static_cast<float*>(workspace);
  l2_loss_partial<false><<<kLossBlocks, 256, 0, s>>>(pred, target, n, partial, nullptr);
  l2_loss_final<<<1, 256, 0, s>>>(partial, kLossBlocks, n, loss);
  return hipGetLastError();
}

hipError_t launch_l2_loss_with_grad(const float* pred, const float* target, long long n, float* loss, float* dunit,
                                    void* workspace, hipStream_t s) {
  float* partial = static_cast<float*>(workspace);
  l2_loss_partial<true><<<kLossBlocks, 256, 0, s>>>(pred, target, n, partial, dunit);
  l2_loss_final<<<1, 256, 0, s>>>(partial, kLossBlocks, n, loss);
  return hipGetLastError();
}

hipError_t launch_l2_loss_grad(const float* pred, const float* target, const float* grad_output, long long n,
                               float* dpred, hipStream_t s) {
  l2_loss_grad<<<kLossBlocks, 256, 0, s>>>(pred, target, grad_output, n, dpred);
  return hipGetLastError();
}

}  // namespace hdrnet_amd

extern "C" int hdrnet_l2_loss_with_grad_f32(const float* prediction, const float* target, long long n, float* loss,
                                            float* dprediction_unit, void* workspace, size_t workspace_bytes,
                                            void* stream) {
  using namespace hdrnet_amd;
  if (n <= 0 || !prediction || !target || !loss || !dprediction_unit || !workspace) return 1;
  if (workspace_bytes < l2_loss_workspace_bytes(n)) return 1;
  if (((uintptr_t)prediction | (uintptr_t)target | (uintptr_t)dprediction_unit) & 15u) return 1;
  return launch_l2_loss_with_grad(prediction, target, n, loss, dprediction_unit, workspace,
                                  static_cast<hipStream_t>(stream)) == hipSuccess ? 0 : 2;
}

extern "C" int hdrnet_l2_loss_grad_scale_f32(float* dprediction, const float* grad_output, long long n, void* stream) {
  using namespace hdrnet_amd;
  if (n <= 0 || !dprediction || !grad_output || ((uintptr_t)dprediction & 15u)) return 1;
  l2_loss_grad_scale<<<kLossBlocks, 256, 0, static_cast<hipStream_t>(stream)>>>(dprediction, grad_output, n);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

namespace {
int adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float* step, float lr,
              float beta1, float beta2, float eps, int eps_hat, void* stream) {
  using namespace hdrnet_amd;
  if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !step) return 1;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) return 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long want = ((n >> 2) + 255) / 256;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
  adam_flat<<<blocks, 256, 0, s>>>(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, eps_hat);
  adam_count<<<1, 1, 0, s>>>(step);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
}  // namespace

extern "C" int hdrnet_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                    float* step, float lr, float beta1, float beta2, float eps, void* stream) {
  return adam_step(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, 0, stream);
}

extern "C" int hdrnet_adam_step_tf_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                       float* step, float lr, float beta1, float beta2, float eps, void* stream) {
  return adam_step(param, grad, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, 1, stream);
}
