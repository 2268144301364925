// Does fp32 MFMA execution overlap with VALU work (a) of other waves on the same SIMD,
// (b) of the same wave when interleaved in program order?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 mfma only, 1 valu only, 2 segregated (16 mfma then 96 valu), 3 interleaved 1:6
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float v[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
      }
    }
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = __builtin_fmaf(v[q], b, a);
      }
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = __builtin_fmaf(v[q], b, a);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] = __builtin_fmaf(v[q], b, a);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = acc0[0] + acc1[1] + acc0[2] + acc1[3];
  for (int q = 0; q < 6; ++q) s += v[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  const int iters = 2000;
  for (int wpb : {1, 2, 4}) {  // workgroups per CU -> waves per SIMD
    const int blocks = 256 * wpb;
    printf("waves/SIMD=%d  mfma-only %.1f us  valu-only %.1f us  segregated %.1f us  interleaved %.1f us\n", wpb,
           run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters),
           run<3>(out, blocks, iters));
  }
  return 0;
}
