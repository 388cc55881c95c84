// bprx_route.hip -- fixed-capacity row routing for the all-to-all multi-GPU modes (user-sharded BPRMF moves
// item rows, the partitioned-user form of item-sharded VBPR moves user rows).  Stateless entry points (no handle), all
// pointers are device pointers; every rank sends exactly `cap` slots to every rank, so the collectives take equal splits
// and nothing is read back to the host to size them (round 2 did this routing with ~30 torch passes per step).
//
//   bprx_route_plan         request r (global row id) -> slot[r] = owner * cap + position among the requests to that owner
//                           (-1: the owner's bucket is full -> *overflow = 1); send_idx[slot] = the owner's local row id.
//                           Rows the requesting rank owns ITSELF never enter the send buffers: slot[r] = -2 - local row id;
//                           unpack copies them straight from the rank's tables and pack adds their gradients straight into them.
//                           Positions are taken per workgroup: requests counted per owner in LDS, one cursor atomic per
//                           (workgroup, owner).  The caller pre-fills send_idx with -1 and zeroes the cursors
//                           (bprx_route_reset).
//   bprx_route_gather       owner side: out[q] = [t0[idx[q]] | t1[idx[q]]] for the requested rows (idx < 0: slot unused, skipped)
//   bprx_route_unpack       requester side: dst0[r] | dst1[r] = got[slot[r]]  (slot < 0: zero row)
//   bprx_route_pack         requester side, gradients: send[slot[r]] = [src0[r] | src1[r]], and src rows back to zero
//   bprx_route_scatter_add  owner side: t0[idx[q]] += scale * rows[q][0:w0], t1[idx[q]] += scale * rows[q][w0:w0+w1]
// A routed row is [w0 floats | w1 floats | pad to a multiple of 4 floats]; a part moves 16 B per lane where its width and
// pointers allow, element-wise otherwise (BPRMF's [Gi row | Bi]: 128 + 1 floats in rows of 132).
//
// Row multiplicities (optional `cnt` arrays, one int per row of the owner's shard, all-zero between steps): nine in ten rows of
// a batch are asked for once, and a row that is added to by ONE occurrence needs no float atomics.  The kernels that see the
// row ids first count them (plan: the requester's own rows; gather: the rows the other ranks ask for); the adding kernels
// (pack for own rows, scatter_add for the returned ones) read the count: 1 -> plain 16-byte read-modify-writes and the count
// back to zero; more -> float atomics as before, then one returning add of 0x10000: the occurrence that finds every other one
// finished (high half + 1 == low half) returns the count to zero.  A reader can never mistake a partly finished row for an
// exclusive one (its count is never exactly 1), so no pass is needed to clear the counts.  Own and returned rows are added by
// different (stream-ordered) kernels, hence two count arrays.  (A row asked for more than 65 535 times in one step overflows the
// halves: its count may then never return to zero and the row keeps the atomic path -- slower, never wrong: a count equals 1
// only for a clean row with one occurrence.)
#include "bprx_internal.h"

namespace {

constexpr int RT_MAXW = 64;   // ranks

__global__ __launch_bounds__(256) void k_route_plan(const int32_t *__restrict__ ids, int64_t na, const int32_t *__restrict__ ids_b,
                                                    int64_t n, int ush, int W, int cap, int me,
                                                    int32_t *__restrict__ slot, int32_t *__restrict__ send_idx,
                                                    int32_t *__restrict__ cursor, int32_t *__restrict__ overflow,
                                                    int32_t *__restrict__ own_cnt) {
  __shared__ int hist[RT_MAXW], base[RT_MAXW];
  const int tid = threadIdx.x;
  if (tid < W) hist[tid] = 0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + tid;
  int owner = -1, local = 0, rank = 0;
  if (r < n) {
    const int g = r < na ? ids[r] : ids_b[r - na];         // (two id arrays back to back: positives, then negatives)
    owner = g / ush;
    if (g < 0 || owner >= W) owner = -1;                  // out of range: never routed (reported as overflow)
    else {
      local = g - owner * ush;
      if (owner != me) rank = atomicAdd(&hist[owner], 1);  // (rows this rank owns itself never enter the send buffers)
      else if (own_cnt) atomicAdd(own_cnt + local, 1);     // ... and are counted: see "Row multiplicities"
    }
  }
  __syncthreads();
  if (tid < W) base[tid] = hist[tid] ? atomicAdd(cursor + tid, hist[tid]) : 0;
  __syncthreads();
  if (r < n) {
    int s = -1;
    if (owner == me && owner >= 0) s = -2 - local;         // own row: slot <= -2 encodes the local row id
    else if (owner >= 0) {
      const int pos = base[owner] + rank;
      if (pos < cap) { s = owner * cap + pos; send_idx[s] = local; }
    }
    if (s == -1) *overflow = 1;
    slot[r] = s;
  }
}

// One lane group of 16 per row.  A routed row is [w0 floats | w1 floats | pad] with a stride of PS = (w0 + w1 + 3) & ~3 floats
// (16-byte aligned rows whatever the widths).  v0 / v1: that part moves 16 B per lane (width a multiple of 4 floats, table
// pointer 16-byte aligned), else element-wise.
__host__ __device__ inline int route_ps(int w0, int w1) { return (w0 + w1 + 3) & ~3; }

__device__ __forceinline__ void row_copy(float *__restrict__ dst, const float *__restrict__ src, int w, int lane, bool vec) {
  if (vec) for (int c = lane * 4; c < w; c += 64) *reinterpret_cast<float4 *>(dst + c) = *reinterpret_cast<const float4 *>(src + c);
  else for (int c = lane; c < w; c += 16) dst[c] = src[c];
}
__device__ __forceinline__ void row_zero(float *__restrict__ dst, int w, int lane, bool vec) {
  if (vec) for (int c = lane * 4; c < w; c += 64) *reinterpret_cast<float4 *>(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);
  else for (int c = lane; c < w; c += 16) dst[c] = 0.f;
}

__device__ __forceinline__ void row_axpy(float *__restrict__ t, const float *__restrict__ g, int w, int lane, bool vec, float scale) {
  if (vec && w <= 256) {                                  // every load of the row in flight before the first store
    float4 x[4], y[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane * 4 + it * 64;
      if (c < w) { x[it] = *reinterpret_cast<float4 *>(t + c); y[it] = *reinterpret_cast<const float4 *>(g + c); }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int c = lane * 4 + it * 64;
      if (c < w) {
        x[it].x += scale * y[it].x; x[it].y += scale * y[it].y; x[it].z += scale * y[it].z; x[it].w += scale * y[it].w;
        *reinterpret_cast<float4 *>(t + c) = x[it];
      }
    }
  } else if (vec) for (int c = lane * 4; c < w; c += 64) {
      float4 x = *reinterpret_cast<float4 *>(t + c);
      const float4 y = *reinterpret_cast<const float4 *>(g + c);
      x.x += scale * y.x; x.y += scale * y.y; x.z += scale * y.z; x.w += scale * y.w;
      *reinterpret_cast<float4 *>(t + c) = x;
    }
  else for (int c = lane; c < w; c += 16) t[c] += scale * g[c];
}
// t0[i] | t1[i] += scale * (g0 | g1) by one 16-lane group; cnt (optional): the row's multiplicity in this kernel's work
__device__ __forceinline__ void row_add(float *__restrict__ t0, int w0, float *__restrict__ t1, int w1, int i, const float *g0,
                                        const float *g1, int lane, bool v0, bool v1, float scale, int32_t *__restrict__ cnt) {
  int c1 = 0;
  if (cnt) {
    if (lane == 0) c1 = cnt[i];
    c1 = __shfl(c1, 0, 16);
  }
  if (c1 == 1) {                                          // the only occurrence of this row: nobody else adds to it here
    row_axpy(t0 + (size_t)i * w0, g0, w0, lane, v0, scale);
    if (w1) row_axpy(t1 + (size_t)i * w1, g1, w1, lane, v1, scale);
    if (lane == 0) cnt[i] = 0;
    return;
  }
  for (int c = lane; c < w0; c += 16) atomicAdd(t0 + (size_t)i * w0 + c, scale * g0[c]);      // (lane = element: contiguous dwords)
  for (int c = lane; c < w1; c += 16) atomicAdd(t1 + (size_t)i * w1 + c, scale * g1[c]);
  if (cnt && lane == 0) {
    const int old = atomicAdd(cnt + i, 0x10000);          // one more occurrence finished; the last one clears the count
    if ((old >> 16) + 1 == (old & 0xffff)) cnt[i] = 0;
  }
}

__global__ __launch_bounds__(256) void k_route_gather(const float *__restrict__ t0, int w0, const float *__restrict__ t1, int w1,
                                                      int rows0, const int32_t *__restrict__ idx, int64_t n,
                                                      float *__restrict__ out, int v0, int v1, int32_t *__restrict__ cnt) {
  const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  if (q >= n) return;
  const int i = idx[q];
  if ((unsigned)i >= (unsigned)rows0) return;             // unused slot (or a bad id): the requester never reads it
  if (cnt && lane == 0) atomicAdd(cnt + i, 1);            // its gradient will come back to this row (k_route_scatter_add)
  float *o = out + (size_t)q * route_ps(w0, w1);
  row_copy(o, t0 + (size_t)i * w0, w0, lane, v0);
  if (w1) row_copy(o + w0, t1 + (size_t)i * w1, w1, lane, v1);
}

__global__ __launch_bounds__(256) void k_route_unpack(const float *__restrict__ got, const int32_t *__restrict__ slot, int64_t n,
                                                      float *__restrict__ d0, int w0, float *__restrict__ d1, int w1, int v0, int v1,
                                                      const float *__restrict__ t0, const float *__restrict__ t1, int rows0) {
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  if (r >= n) return;
  const int s = slot[r];
  if (s <= -2 && t0 && (unsigned)(-2 - s) < (unsigned)rows0) {      // a row this rank owns: straight from its tables
    const int i = -2 - s;
    row_copy(d0 + (size_t)r * w0, t0 + (size_t)i * w0, w0, lane, v0);
    if (w1) row_copy(d1 + (size_t)r * w1, t1 + (size_t)i * w1, w1, lane, v1);
    return;
  }
  if (s < 0) {
    row_zero(d0 + (size_t)r * w0, w0, lane, v0);
    if (w1) row_zero(d1 + (size_t)r * w1, w1, lane, v1);
    return;
  }
  const float *g = got + (size_t)s * route_ps(w0, w1);
  row_copy(d0 + (size_t)r * w0, g, w0, lane, v0);
  if (w1) row_copy(d1 + (size_t)r * w1, g + w0, w1, lane, v1);
}

__global__ __launch_bounds__(256) void k_route_pack(float *__restrict__ s0, int w0, float *__restrict__ s1, int w1,
                                                    const int32_t *__restrict__ slot, int64_t n, float *__restrict__ send, int v0, int v1,
                                                    float *__restrict__ t0, float *__restrict__ t1, int rows0, float scale,
                                                    int32_t *__restrict__ own_cnt, int vt0, int vt1,
                                                    int32_t *__restrict__ send_idx, int64_t nslots, int32_t *__restrict__ cursor, int W) {
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int lane = threadIdx.x & 15;
  // the exchange's send list and cursors return to "empty" for the next step's plan (they were consumed by this step's
  // all-to-all long ago): no memsets between the steps
  if (send_idx) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t q = t; q < nslots; q += (int64_t)gridDim.x * 256) send_idx[q] = -1;
    if (t < W) cursor[t] = 0;
  }
  if (r >= n) return;
  const int s = slot[r];
  float *a = s0 + (size_t)r * w0, *b = w1 ? s1 + (size_t)r * w1 : nullptr;
  if (s <= -2 && t0 && (unsigned)(-2 - s) < (unsigned)rows0) {      // own row: added here
    row_add(t0, w0, t1, w1, -2 - s, a, b, lane, vt0, vt1, scale, own_cnt);
  } else if (s >= 0) {
    float *o = send + (size_t)s * route_ps(w0, w1);
    row_copy(o, a, w0, lane, v0);
    if (w1) row_copy(o + w0, b, w1, lane, v1);
  }
  row_zero(a, w0, lane, v0);                              // the staging gradients return to all-zero
  if (w1) row_zero(b, w1, lane, v1);
}

// one 16-lane group per returned row
__global__ __launch_bounds__(256) void k_route_scatter_add(float *__restrict__ t0, int w0, float *__restrict__ t1, int w1, int rows0,
                                                           const int32_t *__restrict__ idx, const float *__restrict__ rows,
                                                           int64_t n, float scale, int32_t *__restrict__ cnt, int v0, int v1) {
  const int ps = route_ps(w0, w1);
  const int lane = threadIdx.x & 15;
  for (int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4; q < n; q += (int64_t)gridDim.x * 16) {
    const int i = idx[q];
    if ((unsigned)i >= (unsigned)rows0) continue;
    const float *g = rows + (size_t)q * ps;
    row_add(t0, w0, t1, w1, i, g, g + w0, lane, v0, v1, scale, cnt);
  }
}

inline int vec_ok(int w, const void *table, const void *packed) {   // (packed rows are 16-byte aligned when their base is)
  return (w % 4 == 0 && (((uintptr_t)table | (uintptr_t)packed) & 15) == 0) ? 1 : 0;
}
inline unsigned rows_grid(int64_t n) { return (unsigned)((n * 16 + 255) / 256); }
inline int launched() { return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP; }

}  // namespace

extern "C" int bprx_route_reset(int32_t *send_idx, int64_t nslots, int32_t *cursor, int32_t nranks, void *stream) {
  if (!send_idx || !cursor || nslots < 0 || nranks <= 0 || nranks > RT_MAXW) return BPRX_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(send_idx, 0xff, (size_t)nslots * sizeof(int32_t), s) != hipSuccess) return BPRX_E_HIP;
  if (hipMemsetAsync(cursor, 0, (size_t)nranks * sizeof(int32_t), s) != hipSuccess) return BPRX_E_HIP;
  return BPRX_OK;
}

extern "C" int bprx_route_plan(const int32_t *ids, int64_t n, const int32_t *ids_b, int64_t n_b, int32_t rows_per_rank,
                               int32_t nranks, int32_t cap, int32_t my_rank, int32_t *slot, int32_t *send_idx, int32_t *cursor,
                               int32_t *overflow, int32_t *own_cnt, void *stream) {
  if ((!ids && n) || (!ids_b && n_b) || !slot || !send_idx || !cursor || !overflow || n < 0 || n_b < 0 || rows_per_rank <= 0 ||
      nranks <= 0 || nranks > RT_MAXW || cap <= 0)
    return BPRX_E_INVALID;
  if (n + n_b == 0) return BPRX_OK;                        // (an empty batch: nothing requested, the exchange still runs)
  hipLaunchKernelGGL(k_route_plan, dim3((unsigned)((n + n_b + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, n, ids_b,
                     n + n_b, rows_per_rank, nranks, cap, my_rank, slot, send_idx, cursor, overflow, own_cnt);
  return launched();
}

extern "C" int bprx_route_gather(const float *t0, int32_t w0, const float *t1, int32_t w1, int32_t num_rows, const int32_t *idx,
                                 int64_t n, float *out, int32_t *cnt, void *stream) {
  if (!t0 || !idx || !out || w0 <= 0 || w1 < 0 || (w1 && !t1) || num_rows <= 0 || n < 0) return BPRX_E_INVALID;
  if (n == 0) return BPRX_OK;
  hipLaunchKernelGGL(k_route_gather, dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream, t0, w0, t1, w1, num_rows, idx, n, out,
                     vec_ok(w0, t0, out), w0 % 4 == 0 ? vec_ok(w1, t1, out) : 0, cnt);
  return launched();
}

extern "C" int bprx_route_unpack(const float *got, const int32_t *slot, int64_t n, float *dst0, int32_t w0, float *dst1, int32_t w1,
                                 const float *own0, const float *own1, int32_t own_rows, void *stream) {
  if (!got || !slot || !dst0 || w0 <= 0 || w1 < 0 || (w1 && !dst1) || n < 0 || (own0 && w1 && !own1)) return BPRX_E_INVALID;
  if (n == 0) return BPRX_OK;
  hipLaunchKernelGGL(k_route_unpack, dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream, got, slot, n, dst0, w0, dst1, w1,
                     vec_ok(w0, dst0, got) & (own0 ? vec_ok(w0, own0, got) : 1),
                     w0 % 4 == 0 ? (vec_ok(w1, dst1, got) & (own1 ? vec_ok(w1, own1, got) : 1)) : 0, own0, own1, own_rows);
  return launched();
}

extern "C" int bprx_route_pack(float *src0, int32_t w0, float *src1, int32_t w1, const int32_t *slot, int64_t n, float *send,
                               float *own0, float *own1, int32_t own_rows, float scale, int32_t *own_cnt, int32_t *send_idx,
                               int64_t nslots, int32_t *cursor, int32_t nranks, void *stream) {
  if (!src0 || !slot || !send || w0 <= 0 || w1 < 0 || (w1 && !src1) || n < 0 || (own0 && w1 && !own1) ||
      (send_idx && (!cursor || nslots < 0 || nranks <= 0 || nranks > RT_MAXW)))
    return BPRX_E_INVALID;
  if (n == 0 && !send_idx) return BPRX_OK;
  unsigned grid = rows_grid(n);
  if (grid < 1) grid = 1;                                  // (an empty batch still returns the send list to "empty")
  hipLaunchKernelGGL(k_route_pack, dim3(grid), dim3(256), 0, (hipStream_t)stream, src0, w0, src1, w1, slot, n, send,
                     vec_ok(w0, src0, send), w0 % 4 == 0 ? vec_ok(w1, src1, send) : 0, own0, own1, own_rows, scale, own_cnt,
                     own0 ? vec_ok(w0, own0, src0) : 0, (own1 && w0 % 4 == 0) ? vec_ok(w1, own1, src1) : 0, send_idx, nslots, cursor,
                     nranks);
  return launched();
}

extern "C" int bprx_route_scatter_add(float *t0, int32_t w0, float *t1, int32_t w1, int32_t num_rows, const int32_t *idx,
                                      const float *rows, int64_t n, float scale, int32_t *cnt, void *stream) {
  if (!t0 || !idx || !rows || w0 <= 0 || w1 < 0 || (w1 && !t1) || num_rows <= 0 || n < 0) return BPRX_E_INVALID;
  if (n == 0) return BPRX_OK;
  int64_t blocks = (n * 16 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_route_scatter_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t0, w0, t1, w1, num_rows, idx,
                     rows, n, scale, cnt, vec_ok(w0, t0, rows), w0 % 4 == 0 ? vec_ok(w1, t1, rows) : 0);
  return launched();
}
