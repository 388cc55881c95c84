// bprx_probe.hip -- measurement helper, not on the hot path: a plain streaming-read kernel (16 B per lane, 8 loads in
// flight per lane, 512 workgroups x 1024 threads: the shape that reads fastest on MI355X in scripts/probe/ldbw.hip).
// bench.py times it on the launch stream to quote every roofline fraction against what THIS device's HBM delivers to a
// streaming kernel as well as against the 8 TB/s specification.
#include "bprx_internal.h"

namespace {
constexpr int PU = 8;
__global__ __launch_bounds__(1024) void k_probe_stream(const uint4 *__restrict__ p, size_t per_wg, uint4 *__restrict__ sink) {
  const uint4 *base = p + (size_t)blockIdx.x * per_wg;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = threadIdx.x; i + (size_t)(PU - 1) * 1024 < per_wg; i += (size_t)PU * 1024) {
    uint4 v[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) v[u] = base[i + (size_t)u * 1024];
#pragma unroll
    for (int u = 0; u < PU; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  // keeps the loads alive; one 16-byte store per workgroup at most
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && threadIdx.x == 0) sink[blockIdx.x] = acc;
}
}  // namespace

extern "C" int64_t bprx_probe_stream_read(const void *buf, int64_t bytes, void *sink, void *stream) {
  if (!buf || !sink || bytes < (int64_t)512 * PU * 1024 * 16) return BPRX_E_INVALID;
  const size_t per_wg = (size_t)bytes / 16 / 512 / ((size_t)PU * 1024) * ((size_t)PU * 1024);   // whole trips only
  hipLaunchKernelGGL(k_probe_stream, dim3(512), dim3(1024), 0, (hipStream_t)stream, (const uint4 *)buf, per_wg, (uint4 *)sink);
  if (hipGetLastError() != hipSuccess) return BPRX_E_HIP;
  return (int64_t)(per_wg * 16 * 512);                    // bytes actually read
}
