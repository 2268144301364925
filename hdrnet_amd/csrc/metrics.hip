// The training loop's l2 loss, hdrnet/metrics.py:8-11 (`tf.reduce_mean(tf.square(target - prediction))`), and its
// gradient with respect to the prediction, as two HBM-bound passes over the full-resolution batch:
//   forward   reads prediction and target (8 B per element), writes one partial sum per workgroup + the mean
//   backward  reads both again and writes d prediction = (2 / n) * grad_output * (prediction - target) (12 B per element)
// torch's mse_loss is five launches here -- the squared differences written out (and read back by the mean), a
// zeros_like of the gradient that the backward then overwrites -- 143 us at 4 x 1080p against ~85 us for these two.
#include <hip/hip_runtime.h>

#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

constexpr int kLossBlocks = 2048;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void l2_loss_partial(const float* __restrict__ pred, const float* __restrict__ target,
                                                       long long n, float* __restrict__ partial) {
  __shared__ float red[256];
  const long long n4 = n >> 2;
  const v4f* p4 = reinterpret_cast<const v4f*>(pred);
  const v4f* t4 = reinterpret_cast<const v4f*>(target);
  float acc = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const v4f a = __builtin_nontemporal_load(p4 + i), b = __builtin_nontemporal_load(t4 + i);
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // the tail of a length that is not a multiple of 4
    const float d = pred[(n4 << 2) + threadIdx.x] - target[(n4 << 2) + threadIdx.x];
    acc += d * d;
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

}  // namespace

size_t l2_loss_workspace_bytes(long long n) { return n > 0 ? (size_t)kLossBlocks * sizeof(float) : 0; }

hipError_t launch_l2_loss(const float* pred, const float* target, long long n, float* loss, void* workspace,
                          hipStream_t s) {
  float* partial = static_cast<float*>(workspace);
  l2_loss_partial<<<kLossBlocks, 256, 0, s>>>(pred, target, n, partial);
  l2_loss_final<<<1, 256, 0, s>>>(partial, kLossBlocks, n, loss);
  return hipGetLastError();
}

hipError_t launch_l2_loss_grad(const float* pred, const float* target, const float* grad_output, long long n,
                               float* dpred, hipStream_t s) {
  l2_loss_grad<<<kLossBlocks, 256, 0, s>>>(pred, target, grad_output, n, dpred);
  return hipGetLastError();
}

}  // namespace hdrnet_amd
