// Cache-policy variants of the shipped geometry's no-compute stream (4K frame: 133 MB in, 100 MB out).
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr long long NPX = 2160LL * 3840;

// MODE 0 plain; 1 nontemporal stores; 2 nontemporal loads; 3 both; 4 plain, 2 segments (8 quads) per thread-block pair
template <int MODE>
__global__ __launch_bounds__(192) void k(const f4* __restrict__ g, const f4* __restrict__ in, f4* __restrict__ out) {
  const long long q0 = (long long)blockIdx.x * 192;
  const int t = threadIdx.x;
  f4 gv, v[3];
  if constexpr (MODE == 2 || MODE == 3) {
    gv = __builtin_nontemporal_load(g + q0 + t);
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[k2] = __builtin_nontemporal_load(in + q0 * 3 + t + 192 * k2);
  } else {
    gv = g[q0 + t];
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[k2] = in[q0 * 3 + t + 192 * k2];
  }
  const float s = gv.x + gv.y + gv.z + gv.w;
#pragma unroll
  for (int k2 = 0; k2 < 3; ++k2) {
    v[k2] *= s;
    if constexpr (MODE == 1 || MODE == 3) __builtin_nontemporal_store(v[k2], out + q0 * 3 + t + 192 * k2);
    else out[q0 * 3 + t + 192 * k2] = v[k2];
  }
}

int main() {
  const int NSETS = 3;
  f4 *g[NSETS], *in[NSETS], *out[NSETS];
  for (int s = 0; s < NSETS; ++s) {
    hipMalloc(&g[s], NPX * 4); hipMalloc(&in[s], NPX * 12); hipMalloc(&out[s], NPX * 12);
    hipMemset(g[s], 0, NPX * 4); hipMemset(in[s], 0, NPX * 12);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = NPX * 28.0;
  auto time = [&](auto launch, const char* name) {
    std::vector<float> ts;
    for (int r = 0; r < 5; ++r) {
      for (int i = 0; i < 20; ++i) launch(i % NSETS);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 100; ++i) launch(i % NSETS);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ts.push_back(ms * 1e3f / 100);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-44s median %6.2f us -> %6.1f GB/s (%.1f %% of 8 TB/s)\n", name, ts[2], bytes / ts[2] / 1e3, bytes / ts[2] / 80);
  };
  for (int rep = 0; rep < 2; ++rep) {
    time([&](int s) { k<0><<<10800, 192>>>(g[s], in[s], out[s]); }, "plain loads, plain stores");
    time([&](int s) { k<1><<<10800, 192>>>(g[s], in[s], out[s]); }, "plain loads, nontemporal stores");
    time([&](int s) { k<2><<<10800, 192>>>(g[s], in[s], out[s]); }, "nontemporal loads, plain stores");
    time([&](int s) { k<3><<<10800, 192>>>(g[s], in[s], out[s]); }, "nontemporal loads, nontemporal stores");
  }
  return 0;
}
