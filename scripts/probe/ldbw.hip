// Load-bandwidth probe: what one CU / the chip pulls through the vector-memory path as a function of the bytes kept in
// flight (U 16-B loads per lane), the waves per workgroup, the workgroup count, and where the data lives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int U>
__global__ void k_stream(const uint4 *__restrict__ p, size_t per_wg, int reps, uint4 *__restrict__ out) {
  const uint4 *base = p + (size_t)blockIdx.x * per_wg;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int nt = blockDim.x;
  for (int rep = 0; rep < reps; ++rep) {
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * nt < per_wg; i += (size_t)U * nt) {
      uint4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = base[i + (size_t)u * nt];
#pragma unroll
      for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * nt + threadIdx.x] = acc;   // never true for the fill pattern: keeps the loads
}

template <int U>
float run(const uint4 *p, size_t per_wg, int reps, uint4 *out, int wgs, int threads) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_stream<U>, dim3(wgs), dim3(threads), 0, 0, p, per_wg, reps, out);
  hipEventRecord(a, 0);
  const int L = 5;
  for (int w = 0; w < L; ++w) hipLaunchKernelGGL(k_stream<U>, dim3(wgs), dim3(threads), 0, 0, p, per_wg, reps, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / L;
}

int main() {
  const size_t total = (size_t)1 << 30;   // 1 GiB
  uint4 *p, *out;
  hipMalloc(&p, total);
  hipMalloc(&out, (size_t)1 << 24);
  hipMemset(p, 1, total);
  printf("mode wgs threads U bytes_in_flight_per_wg GBps B_per_clk_per_wg(2.4GHz)\n");
  for (int mode = 0; mode < 2; ++mode) {                   // 0: HBM stream (each WG its own 1 GiB / wgs), 1: 64 KB per WG re-read (L2 / L1)
    for (int wgs : {256, 512, 1024}) {
      for (int threads : {256, 512, 1024}) {
        for (int U : {1, 2, 4, 8, 16}) {
          size_t per_wg = mode == 0 ? total / 16 / wgs : (size_t)(64 * 1024) / 16;
          int reps = mode == 0 ? 1 : 64;
          float ms;
          switch (U) {
            case 1: ms = run<1>(p, per_wg, reps, out, wgs, threads); break;
            case 2: ms = run<2>(p, per_wg, reps, out, wgs, threads); break;
            case 4: ms = run<4>(p, per_wg, reps, out, wgs, threads); break;
            case 8: ms = run<8>(p, per_wg, reps, out, wgs, threads); break;
            default: ms = run<16>(p, per_wg, reps, out, wgs, threads); break;
          }
          double bytes = (double)per_wg * 16 * reps * wgs;
          double gbps = bytes / (ms * 1e-3) / 1e9;
          printf("%s %d %d %d %d %.0f %.1f\n", mode == 0 ? "hbm" : "l2", wgs, threads, U, threads * U * 16, gbps,
                 gbps * 1e9 / wgs / 2.4e9 * (wgs > 256 ? wgs / 256.0 : 1.0));
          fflush(stdout);
        }
      }
    }
  }
  return 0;
}
