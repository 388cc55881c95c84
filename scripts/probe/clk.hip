// In-kernel shader clock (s_memtime ticks per 100-MHz s_memrealtime tick) for SHORT kernels launched into an otherwise
// idle GPU vs. a busy one: how much of a small-batch step's kernel time is DVFS state rather than work.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <chrono>
__global__ void k_clk(unsigned long long *out, int iters) {
  float x = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.000001f, 0.5f);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (x == 12345.f) out[2] = 1;
}
__global__ void k_heat(float *p, int n) {
  float x = p[threadIdx.x];
  for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.000001f, 0.5f);
  p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned long long *d, h[3];
  float *heat;
  hipMalloc(&d, 64); hipMalloc(&heat, 1024 * 256 * 4);
  auto probe = [&](const char *what, int iters) {
    hipLaunchKernelGGL(k_clk, dim3(32), dim3(256), 0, 0, d, iters);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-44s %8llu cycles in %7.2f us -> %.3f GHz\n", what, h[0], h[1] / 100.0, h[0] / (h[1] * 10.0));
  };
  usleep(300000);
  probe("idle 300 ms, short kernel (2k fma)", 2000);
  usleep(300000);
  probe("idle 300 ms, 20k fma", 20000);
  for (int gap_us : {1000, 100, 20}) {                  // sparse launches: one short kernel every gap_us
    double t0 = now();
    while (now() - t0 < 0.5) { hipLaunchKernelGGL(k_clk, dim3(32), dim3(256), 0, 0, d, 2000); hipDeviceSynchronize(); usleep(gap_us); }
    char buf[80]; snprintf(buf, sizeof buf, "0.5 s of short kernels every ~%d us", gap_us);
    probe(buf, 2000);
  }
  { double t0 = now(); while (now() - t0 < 0.5) { for (int q = 0; q < 64; ++q) hipLaunchKernelGGL(k_clk, dim3(32), dim3(256), 0, 0, d, 2000); hipDeviceSynchronize(); }
    probe("0.5 s of back-to-back short kernels (32 WG)", 2000); }
  { double t0 = now(); while (now() - t0 < 2.0) { hipLaunchKernelGGL(k_heat, dim3(1024), dim3(256), 0, 0, heat, 200000); hipDeviceSynchronize(); }
    probe("after 2 s of chip-wide busy kernels", 2000); }
  return 0;
}
