// Issue cost of the VALU instruction classes the slicing kernels are made of (gfx950):
// cycles of SIMD time per wave64 instruction = elapsed s_memtime cycles / (instructions * waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_sqrt_f32, 3 v_cndmask, 4 v_mul_lo_u32, 5 v_fma + 1:1 ds_read_b128
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ float4 lds[256];
  lds[threadIdx.x] = make_float4(1, 2, 3, 4);
  __syncthreads();
  float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f;
  float v[8];
  f32x2 p[8];
  unsigned u[8];
  for (int q = 0; q < 8; ++q) { v[q] = q + 1.f; p[q] = f32x2{q + 1.f, q + 2.f}; u[q] = q + threadIdx.x; }
  float4 l = make_float4(0, 0, 0, 0);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (MODE == 0) v[q] = __builtin_fmaf(v[q], b, a);
        if constexpr (MODE == 1) p[q] = __builtin_elementwise_fma(p[q], f32x2{b, b}, f32x2{a, a});
        if constexpr (MODE == 2) v[q] = __builtin_amdgcn_sqrtf(v[q]) + 0.0f * a;
        if constexpr (MODE == 3) v[q] = (v[q] > a) ? b : v[q];
        if constexpr (MODE == 4) u[q] = u[q] * (unsigned)(threadIdx.x | 3);
        if constexpr (MODE == 5) {
          v[q] = __builtin_fmaf(v[q], b, a);
          if (q == 0) { float4 t = lds[(threadIdx.x + r) & 255]; l.x += t.x; }
        }
      }
    }
  }
  const long long t1 = clock64();
  float s = l.x;
  for (int q = 0; q < 8; ++q) s += v[q] + p[q][0] + p[q][1] + (float)u[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc, int wpb, int iters, int per_iter) {
  const int blocks = 256 * wpb;
  k<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  k<MODE><<<blocks, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long h[2048];
  hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < blocks; ++i) m += h[i];
  m /= blocks;
  printf("%-28s waves/SIMD=%d  %.2f cycles of SIMD time per wave-instruction\n", name, wpb,
         m / ((double)iters * per_iter * wpb));
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 2048 * 256 * sizeof(float));
  hipMalloc(&cyc, 2048 * sizeof(long long));
  for (int wpb : {1, 2, 4}) {
    run<0>("v_fma_f32", out, cyc, wpb, 2000, 64);
    run<1>("v_pk_fma_f32", out, cyc, wpb, 2000, 64);
    run<2>("v_sqrt_f32 (+v_fma)", out, cyc, wpb, 2000, 128);
    run<3>("v_cmp + v_cndmask", out, cyc, wpb, 2000, 128);
    run<4>("v_mul_lo_u32", out, cyc, wpb, 2000, 64);
    run<5>("v_fma_f32 + ds_read_b128 8:1", out, cyc, wpb, 2000, 64);
  }
  return 0;
}
