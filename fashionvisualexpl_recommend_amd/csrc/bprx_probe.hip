// bprx_probe.hip -- measurement helper, not on the hot path: a plain streaming-read kernel (16 B per lane, 8 loads in
// flight per lane, 512 workgroups x 1024 threads: the shape that reads fastest on MI355X in scripts/probe/ldbw.hip).
// bench.py times it on the launch stream to quote every roofline fraction against what THIS device's HBM delivers to a
// streaming kernel as well as against the 8 TB/s specification.
#include "bprx_internal.h"

namespace {
constexpr int PU = 8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <bool NT>
__global__ __launch_bounds__(1024) void k_probe_stream(const u32x4_t *__restrict__ p, size_t per_wg, u32x4_t *__restrict__ sink) {
  const u32x4_t *base = p + (size_t)blockIdx.x * per_wg;
  u32x4_t acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i + (size_t)(PU - 1) * 1024 < per_wg; i += (size_t)PU * 1024) {
    u32x4_t v[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) v[u] = NT ? __builtin_nontemporal_load(base + i + (size_t)u * 1024) : base[i + (size_t)u * 1024];
#pragma unroll
    for (int u = 0; u < PU; ++u) acc ^= v[u];
  }
  // keeps the loads alive; one 16-byte store per workgroup at most
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && threadIdx.x == 0) sink[blockIdx.x] = acc;
}

// Random-row probe: one lane group per listed row, 16 B per lane (the access shape of k_triplet_grad / k_item_seg: a
// 256..1024-byte table row picked by an index); mode 0 reads the row, mode 1 reads it and writes it back in place
// (the exclusive-row sgd update).  `idx` holds n row numbers (distinct when mode = 1).
template <int G>
__global__ __launch_bounds__(256) void k_probe_rows(float *__restrict__ table, const int32_t *__restrict__ idx, int64_t n,
                                                    int row_floats, int mode, float *__restrict__ sink) {
  const int64_t job = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (job >= n) return;
  float *row = table + (size_t)idx[job] * row_floats;
  float acc = 0.f;
  for (int c = lane * 4; c < row_floats; c += G * 4) {
    float4 v = *reinterpret_cast<const float4 *>(row + c);
    if (mode) {
      v.x *= 1.0000001f; v.y *= 1.0000001f; v.z *= 1.0000001f; v.w *= 1.0000001f;
      *reinterpret_cast<float4 *>(row + c) = v;
    } else acc += v.x + v.y + v.z + v.w;
  }
  if (!mode && acc == 1.2345e30f) sink[0] = acc;           // keeps the loads alive
}
}  // namespace

extern "C" int64_t bprx_probe_row_gather(void *table, int64_t num_rows, int32_t row_floats, const int32_t *idx, int64_t n,
                                         int32_t mode, void *sink, void *stream) {
  if (!table || !idx || !sink || num_rows <= 0 || n <= 0 || row_floats < 16 || row_floats % 4) return BPRX_E_INVALID;
  const int per = row_floats / 4;                           // lanes that a row's 16-byte pieces fill
  const int G = per >= 64 ? 64 : per >= 32 ? 32 : per >= 16 ? 16 : per >= 8 ? 8 : 4;
  const dim3 grid((unsigned)((n * G + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  switch (G) {
    case 64: hipLaunchKernelGGL(k_probe_rows<64>, grid, dim3(256), 0, s, (float *)table, idx, n, row_floats, mode, (float *)sink); break;
    case 32: hipLaunchKernelGGL(k_probe_rows<32>, grid, dim3(256), 0, s, (float *)table, idx, n, row_floats, mode, (float *)sink); break;
    case 16: hipLaunchKernelGGL(k_probe_rows<16>, grid, dim3(256), 0, s, (float *)table, idx, n, row_floats, mode, (float *)sink); break;
    case 8: hipLaunchKernelGGL(k_probe_rows<8>, grid, dim3(256), 0, s, (float *)table, idx, n, row_floats, mode, (float *)sink); break;
    default: hipLaunchKernelGGL(k_probe_rows<4>, grid, dim3(256), 0, s, (float *)table, idx, n, row_floats, mode, (float *)sink); break;
  }
  if (hipGetLastError() != hipSuccess) return BPRX_E_HIP;
  return n * (int64_t)row_floats * 4 * (mode ? 2 : 1);     // bytes moved
}

static int64_t probe_stream(const void *buf, int64_t bytes, void *sink, void *stream, bool nt) {
  if (!buf || !sink || bytes < (int64_t)512 * PU * 1024 * 16) return BPRX_E_INVALID;
  const size_t per_wg = (size_t)bytes / 16 / 512 / ((size_t)PU * 1024) * ((size_t)PU * 1024);   // whole trips only
  if (nt) hipLaunchKernelGGL(k_probe_stream<true>, dim3(512), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t *)buf, per_wg, (u32x4_t *)sink);
  else hipLaunchKernelGGL(k_probe_stream<false>, dim3(512), dim3(1024), 0, (hipStream_t)stream, (const u32x4_t *)buf, per_wg, (u32x4_t *)sink);
  if (hipGetLastError() != hipSuccess) return BPRX_E_HIP;
  return (int64_t)(per_wg * 16 * 512);                    // bytes actually read
}

extern "C" int64_t bprx_probe_stream_read(const void *buf, int64_t bytes, void *sink, void *stream) {
  return probe_stream(buf, bytes, sink, stream, false);
}
// the same with `nt` (streaming, no-retain) loads: what the bf16 feature passes use (Infinity Cache policy)
extern "C" int64_t bprx_probe_stream_read_nt(const void *buf, int64_t bytes, void *sink, void *stream) {
  return probe_stream(buf, bytes, sink, stream, true);
}
