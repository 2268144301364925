// A 1080p frame (58 MB) fits the device in ONE round of workgroups, so every workgroup reads at the
// same time and writes at the same time: HBM never sees reads and writes together.  Does splitting
// each thread's work into two sub-steps (loads of B in flight while A is stored) help?
// No compute, the forward's byte volume per pixel (16 B in, 12 B out).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int H = 1080, W = 1920;
constexpr long long NPX = (long long)H * W;

// PHASES quads per thread, processed in order; all loads are issued up front.
template <int PHASES>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ g, const float4* __restrict__ in,
                                         float4* __restrict__ out, int quads_per_block) {
  const long long q0 = (long long)blockIdx.x * quads_per_block;
  const int t = threadIdx.x;
  const int per_phase = quads_per_block / PHASES;  // == blockDim.x
  float4 gv[PHASES], v[PHASES][3];
#pragma unroll
  for (int ph = 0; ph < PHASES; ++ph) {
    const long long q = q0 + ph * per_phase;
    gv[ph] = g[q + t];
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) v[ph][k2] = in[q * 3 + t + per_phase * k2];
  }
#pragma unroll
  for (int ph = 0; ph < PHASES; ++ph) {
    const long long q = q0 + ph * per_phase;
    const float s = gv[ph].x + gv[ph].y + gv[ph].z + gv[ph].w;
#pragma unroll
    for (int k2 = 0; k2 < 3; ++k2) {
      v[ph][k2].x *= s; v[ph][k2].y *= s; v[ph][k2].z *= s; v[ph][k2].w *= s;
      out[q * 3 + t + per_phase * k2] = v[ph][k2];
    }
  }
}

int main() {
  const int NSETS = 6;
  float4 *g[NSETS], *in[NSETS], *out[NSETS];
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
      for (int i = 0; i < 50; ++i) launch(i % NSETS);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) launch(i % NSETS);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      ts.push_back(ms * 1e3f / 200);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-60s median %6.2f us -> %6.1f GB/s\n", name, ts[2], bytes / ts[2] / 1e3);
  };
  const long long nq = NPX / 4;  // 518400 quads
  for (int rep = 0; rep < 2; ++rep) {
    time([&](int s) { k<1><<<nq / 240, 240>>>(g[s], in[s], out[s], 240); }, "1 quad / thread, 240-thread blocks (2160 blocks)");
    time([&](int s) { k<1><<<nq / 120, 120>>>(g[s], in[s], out[s], 120); }, "1 quad / thread, 120-thread blocks (4320 blocks)");
    time([&](int s) { k<2><<<nq / 480, 240>>>(g[s], in[s], out[s], 480); }, "2 quads / thread, 240-thread blocks (1080 blocks)");
    time([&](int s) { k<2><<<nq / 240, 120>>>(g[s], in[s], out[s], 240); }, "2 quads / thread, 120-thread blocks (2160 blocks)");
    time([&](int s) { k<4><<<nq / 480, 120>>>(g[s], in[s], out[s], 480); }, "4 quads / thread, 120-thread blocks (1080 blocks)");
    time([&](int s) { k<4><<<nq / 256, 64>>>(g[s], in[s], out[s], 256); }, "4 quads / thread, 64-thread blocks (2025 blocks)");
  }
  return 0;
}
