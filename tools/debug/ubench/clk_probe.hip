// Effective shader clock of one CU while something else loads the chip: a single wave sleeps through `iters`
// s_sleep periods and reports how many shader-clock ticks (clock64 = s_memtime) and how many 100-MHz wall-clock
// ticks (wall_clock64 = s_memrealtime) passed.  Built as a shared library (make bin/libclkprobe.so) so that
// tools/power_probe.py can launch it on a second stream beside the kernels it times: the firmware's REPORTED
// clock (rocm-smi) and this counted one need not agree (clock stretching / duty-cycle throttles).
#include <hip/hip_runtime.h>

__global__ void clk_probe_kernel(long long* out, int iters) {
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
}

extern "C" int clk_probe_launch(long long* out, int iters, void* stream) {
  hipLaunchKernelGGL(clk_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, iters);
  return static_cast<int>(hipGetLastError());
}
