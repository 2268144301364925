// How does v_cvt_pk_u8_f32 round and saturate?  (Candidate for the uint8 output path: one instruction for
// clamp + convert + pack.)  hipcc --offload-arch=gfx950 cvt_pk_u8.hip -o bin/cvt_pk_u8 && bin/cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned* o, int n) {
  int i = threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1, 0xAA0000BBu);
}
int main() {
  const float h[] = {-3.f, -0.5f, 0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 2.51f, 3.5f, 127.5f, 128.5f, 254.4f, 254.5f, 254.6f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f};
  const int n = sizeof(h) / sizeof(float);
  float* dx; unsigned* d; unsigned out[64];
  hipMalloc(&dx, sizeof(h)); hipMalloc(&d, n * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, d, n);
  hipMemcpy(out, d, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%12g -> byte1 = %3u   (dword %08x)\n", h[i], (out[i] >> 8) & 0xff, out[i]);
  return 0;
}
