// Access-pattern probe: the SAME bytes (a [rows x 8 KB] bf16 matrix) pulled by (a) the MFMA A-fragment pattern of the
// forward projection (per wave instruction: 16 rows x 64 B, rows 8 KB apart), (b) 2 rows x 512 B (backward tile
// pattern), (c) fully contiguous 1 KB per wave instruction.  One workgroup of `threads` per 128 rows, U loads in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int ROWB = 8192;   // bytes per row

template <int U, int PAT>
__global__ void k_pat(const char *__restrict__ p, int rows_per_wave, int reps, uint4 *__restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t wave_row0 = ((size_t)blockIdx.x * nw + w) * rows_per_wave;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int rep = 0; rep < reps; ++rep) {
    for (int rb = 0; rb < rows_per_wave; rb += 16) {                 // 16 rows x 8 KB = 128 KB per block of rows
      const char *base = p + (wave_row0 + rb) * ROWB;
      // one "instruction slot" = 1 KB; a block of 16 rows has 128 slots
      for (int s0 = 0; s0 < 128; s0 += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int s = s0 + u;
          size_t off;
          if (PAT == 0) off = (size_t)(lane & 15) * ROWB + (size_t)s * 64 + (lane >> 4) * 16;          // 16 rows x 64 B
          else if (PAT == 1) off = (size_t)((s >> 4) * 2 + (lane >> 5)) * ROWB + (size_t)(s & 15) * 512 + (lane & 31) * 16;  // 2 rows x 512 B
          else if (PAT == 2) off = (size_t)(s >> 3) * ROWB + (size_t)(s & 7) * 1024 + lane * 16;         // 1 row x 1 KB
          else off = (size_t)(s >> 2) * 4096 + (size_t)(lane & 15) * 256 + (size_t)(s & 3) * 64 + (lane >> 4) * 16;  // 16 rows x 64 B inside a contiguous 4-KB block (tiled F)
          v[u] = *reinterpret_cast<const uint4 *>(base + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
      }
    }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int U, int PAT>
float run(const char *p, int rpw, int reps, uint4 *out, int wgs, int threads) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_pat<U, PAT>), dim3(wgs), dim3(threads), 0, 0, p, rpw, reps, out);
  hipEventRecord(a, 0);
  const int L = 5;
  for (int w = 0; w < L; ++w) hipLaunchKernelGGL((k_pat<U, PAT>), dim3(wgs), dim3(threads), 0, 0, p, rpw, reps, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / L;
}

template <int PAT>
void sweep(const char *p, uint4 *out, const char *name) {
  for (int wgs : {256, 512}) {
    for (int threads : {256, 512}) {
      for (int U : {4, 8, 16}) {
        const int nw = threads / 64;
        const int rpw = 65536 / (wgs * nw);                 // 65536 rows = 512 MiB in total
        float ms = U == 4 ? run<4, PAT>(p, rpw, 1, out, wgs, threads) : U == 8 ? run<8, PAT>(p, rpw, 1, out, wgs, threads)
                                                                              : run<16, PAT>(p, rpw, 1, out, wgs, threads);
        double bytes = (double)wgs * nw * rpw * ROWB;
        double gbps = bytes / (ms * 1e-3) / 1e9;
        printf("%s hbm wgs=%d threads=%d U=%d  %.0f GB/s  %.1f B/clk/CU\n", name, wgs, threads, U, gbps, gbps * 1e9 / 256 / 2.4e9);
        fflush(stdout);
      }
    }
  }
  // cache-resident: every wave re-reads its first 16 rows (128 KB per wave: L2 hits)
  for (int threads : {256, 512}) {
    const int wgs = 256, nw = threads / 64;
    float ms = run<8, PAT>(p, 16, 16, out, wgs, threads);
    double bytes = (double)wgs * nw * 16 * ROWB * 16;
    double gbps = bytes / (ms * 1e-3) / 1e9;
    printf("%s l2  wgs=%d threads=%d U=8  %.0f GB/s  %.1f B/clk/CU\n", name, wgs, threads, gbps, gbps * 1e9 / 256 / 2.4e9);
  }
}

int main() {
  const size_t total = (size_t)65536 * ROWB;
  char *p; uint4 *out;
  hipMalloc(&p, total);
  hipMalloc(&out, (size_t)1 << 24);
  hipMemset(p, 1, total);
  sweep<0>(p, out, "frag16x64 ");
  sweep<1>(p, out, "rows2x512 ");
  sweep<2>(p, out, "contig1KB ");
  sweep<3>(p, out, "frag_tiled");
  return 0;
}
