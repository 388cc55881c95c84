// bprx_philox.hip -- device-side throughput sampler (replaces the role of
// DataLoader.all_triple_batches, dataset.py:83-114, when bit-compatibility with the reference's MT19937 stream
// is not required).  Counter-based Philox4x32-10: triplet n of stream `seed` is a pure function of (seed, n):
//   block(n, a) = philox(key = seed, ctr = (n_lo, n_hi, a, 0))
//   positive p  = mulhi64(block(n,0).xy, N)  -> (pos_user[p], items_sorted[p])   uniform over training interactions
//   negative j  = mulhi32(block(n,a).z, I), a = 0,1,.. until j is not a positive of the user (binary search in the
//                 user's ascending item list; at most 1024 attempts)              uniform over non-positives
// so any rank can regenerate any other rank's triplets, and the CPU twin (oracle/bpr_oracle.c orc_sample_philox)
// is bit-exact.  One thread per triplet; 12 B written per triplet.
#include <hip/hip_runtime.h>

#include "bprx_internal.h"

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the owner plane (id >> sh: one byte) and the local plane (id & (2^sh - 1): a byte at sh == 8, else 16 bits) of a triplet's items
__device__ __forceinline__ void put_planes(uint8_t *__restrict__ own8, uint8_t *__restrict__ loc8, long long pi, long long pj,
                                           int32_t ii, int32_t jj, int sh) {
  own8[pi] = (uint8_t)(ii >> sh); own8[pj] = (uint8_t)(jj >> sh);
  if (sh == 8) { loc8[pi] = (uint8_t)ii; loc8[pj] = (uint8_t)jj; }
  else {
    uint16_t *l16 = reinterpret_cast<uint16_t *>(loc8);
    const int m = (1 << sh) - 1;
    l16[pi] = (uint16_t)(ii & m); l16[pj] = (uint16_t)(jj & m);
  }
}

__global__ __launch_bounds__(256) void k_sample_philox(const int64_t *__restrict__ indptr, const int32_t *__restrict__ items,
                                                       const int32_t *__restrict__ pos_user, unsigned long long N, uint32_t I,
                                                       uint32_t k0, uint32_t k1, unsigned long long first, long long B,
                                                       int32_t *__restrict__ u, int32_t *__restrict__ i, int32_t *__restrict__ j,
                                                       uint8_t *__restrict__ own8, uint8_t *__restrict__ loc8, long long pl_i,
                                                       long long pl_j, int sh) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const unsigned long long n = first + (unsigned long long)b;
  uint32_t r[4];
  philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 0u, 0u, k0, k1, r);
  const unsigned long long x = ((unsigned long long)r[1] << 32) | r[0];
  const unsigned long long p = __umul64hi(x, N);
  const int32_t uu = pos_user[p];
  const long long lo0 = indptr[uu], len = indptr[uu + 1] - lo0;
  const int32_t *lst = items + lo0;
  int32_t jj = 0;
  for (uint32_t a = 0; a < 1024u; ++a) {
    if (a) philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), a, 0u, k0, k1, r);
    jj = (int32_t)__umulhi(r[2], I);
    long long lo = 0, hi = len;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (lst[mid] < jj) lo = mid + 1; else hi = mid;
    }
    if (!(lo < len && lst[lo] == jj)) break;
  }
  const int32_t ii = items[p];
  u[b] = uu; i[b] = ii; j[b] = jj;
  if (own8) put_planes(own8, loc8, pl_i + b, pl_j + b, ii, jj, sh);   // byte planes of the item ids for the handle's index pass
}

// Epoch-walk mode (the reference's order, dataset.py:93-107, as a stateless stream): within epoch e the users come in the
// order perm[0..U) and every positive of a user is emitted once, consecutively; stream position n (0 <= n < N) of the
// epoch maps to the user a with epoch_ptr[a] <= n < epoch_ptr[a+1] (binary search) and to that user's positive number
// n - epoch_ptr[a].  The negative is the Philox rejection draw keyed by (seed; n, epoch).  Batches are user-grouped like
// the reference's, every interaction is visited exactly once per epoch.
__global__ __launch_bounds__(256) void k_sample_epoch(const int64_t *__restrict__ indptr, const int32_t *__restrict__ items,
                                                      const int32_t *__restrict__ perm, const int64_t *__restrict__ epoch_ptr,
                                                      const int32_t *__restrict__ pos_slot, int U, uint32_t I, uint32_t k0, uint32_t k1, uint32_t epoch,
                                                      long long first, long long B, int32_t *__restrict__ u,
                                                      int32_t *__restrict__ i, int32_t *__restrict__ j,
                                                      uint8_t *__restrict__ own8, uint8_t *__restrict__ loc8, long long pl_i,
                                                      long long pl_j, int sh) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const long long n = first + b;
  int lo = 0, hi = U;                                   // largest a with epoch_ptr[a] <= n (and a non-empty list)
  if (pos_slot) lo = pos_slot[n];                       // precomputed once per epoch
  else
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (epoch_ptr[mid] <= n) lo = mid; else hi = mid;
    }
  const int32_t uu = perm[lo];
  const long long l0 = indptr[uu], len = indptr[uu + 1] - l0;
  const int32_t *lst = items + l0;
  uint32_t r[4];
  int32_t jj = 0;
  for (uint32_t a = 0; a < 1024u; ++a) {
    philox4x32_10((uint32_t)n, (uint32_t)((unsigned long long)n >> 32), a, epoch, k0, k1, r);
    jj = (int32_t)__umulhi(r[2], I);
    long long l = 0, h = len;
    while (l < h) {
      const long long mid = (l + h) >> 1;
      if (lst[mid] < jj) l = mid + 1; else h = mid;
    }
    if (!(l < len && lst[l] == jj)) break;
  }
  const int32_t ii = lst[n - epoch_ptr[lo]];
  u[b] = uu; i[b] = ii; j[b] = jj;
  if (own8) put_planes(own8, loc8, pl_i + b, pl_j + b, ii, jj, sh);
}

// The user order of an epoch: a keyed permutation of [0, U) evaluated POINTWISE -- slot a of epoch `epoch` holds user
//   perm(a) = cycle-walk of a 4-round Feistel network over 2*half bits (half = ceil(bits(U-1) / 2)), round function
//             F_r(R) = philox(key = seed, ctr = (R, r, 0xFFFFFFFE, epoch))[0] & (2^half - 1); walk until the value is < U
// (a bijection of [0, 2^(2 half)) restricted to [0, U) by cycle-walking: a bijection of [0, U); < 4 walks on average).  No keys,
// no sort: until the end of round 3 the order was the stable argsort of per-user Philox keys -- a device merge sort of 8 launches
// and 60 us per epoch inside a preparation of 23 launches and 158 us (5 us per C2 step); this is one launch of a few us.  The third
// counter word never collides with the negative draws (attempt numbers < 1024).  CPU twin: oracle orc_epoch_perm.
__device__ __forceinline__ uint32_t epoch_perm_at(uint32_t a, uint32_t U, int half, uint32_t k0, uint32_t k1, uint32_t epoch) {
  const uint32_t mask = (1u << half) - 1u;
  uint32_t x = a;
  do {
    uint32_t L = x >> half, R = x & mask;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
      uint32_t o[4];
      philox4x32_10(R, r, 0xFFFFFFFEu, epoch, k0, k1, o);
      const uint32_t t = L ^ (o[0] & mask);
      L = R; R = t;
    }
    x = (L << half) | R;
  } while (x >= U);
  return x;
}

__host__ __device__ inline int epoch_perm_half(uint32_t U) {          // half the bits of the Feistel domain (>= 1)
  int nb = 1;
  while (nb < 32 && (U - 1u) >> nb) ++nb;
  return (nb + 1) / 2;
}

// slot a -> its user and the length of that user's list (the prefix sums of the lengths are the epoch's position offsets)
__global__ __launch_bounds__(256) void k_epoch_prepare(uint32_t k0, uint32_t k1, uint32_t epoch, int U, int half,
                                                       const int64_t *__restrict__ indptr, int32_t *__restrict__ perm,
                                                       int64_t *__restrict__ lens) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= U) return;
  const uint32_t u = epoch_perm_at((uint32_t)a, (uint32_t)U, half, k0, k1, epoch);
  perm[a] = (int32_t)u;
  lens[a] = indptr[u + 1] - indptr[u];
}

// position -> slot: positions [epoch_ptr[a], epoch_ptr[a+1]) belong to slot a (one lane group of 8 per slot)
__global__ __launch_bounds__(256) void k_epoch_slots(const int64_t *__restrict__ epoch_ptr, int U, int32_t *__restrict__ pos_slot,
                                                     int64_t num_pos) {
  const int a = (blockIdx.x * 256 + threadIdx.x) >> 3, lane = threadIdx.x & 7;
  if (a >= U) return;
  const int64_t p0 = epoch_ptr[a];
  int64_t p1 = epoch_ptr[a + 1];
  p1 = p1 < num_pos ? p1 : num_pos;                        // (never beyond the array, whatever the CSR claims)
  for (int64_t p = p0 + lane; p < p1; p += 8) pos_slot[p] = a;
}

}  // namespace

extern "C" int bprx_epoch_prepare(uint64_t seed, uint32_t epoch, int32_t num_users, const int64_t *indptr, int32_t *perm,
                                  int64_t *lens, void *stream) {
  if (!indptr || !perm || !lens || num_users <= 0) return BPRX_E_INVALID;
  hipLaunchKernelGGL(k_epoch_prepare, dim3((unsigned)((num_users + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32), epoch, num_users, epoch_perm_half((uint32_t)num_users), indptr, perm, lens);
  return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP;
}

extern "C" int bprx_epoch_slots(const int64_t *epoch_ptr, int32_t num_users, int32_t *pos_slot, int64_t num_pos, void *stream) {
  if (!epoch_ptr || !pos_slot || num_users <= 0 || num_pos < 0) return BPRX_E_INVALID;
  hipLaunchKernelGGL(k_epoch_slots, dim3((unsigned)(((int64_t)num_users * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     epoch_ptr, num_users, pos_slot, num_pos);
  return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP;
}

// `h` (optional): the handle whose next step will run on these buffers.  When its index pass can use them (segment mode,
// num_items <= 65 536, whole batch a multiple of 16), the sampler also writes the high and the low byte of every item id into
// the handle's byte planes (2 B per occurrence): k_index_seg's owners then scan ONE byte per occurrence -- sixteen values per
// 16-byte load, tested together -- instead of four bytes and a compare per value.  batch_offset / batch_size: this call fills
// triplets [batch_offset, batch_offset + B) of a batch of batch_size (an epoch crossing fills a batch in two calls); the planes
// are valid for the step that is called with exactly these buffers and B = batch_size, and are consumed by it.
static void plane_args(bprx_handle *h, const int32_t *pos, const int32_t *neg, int64_t B, int64_t batch_offset, int64_t batch_size,
                       uint8_t **own8, uint8_t **loc8, long long *pl_i, long long *pl_j, int *sh) {
  *own8 = nullptr; *loc8 = nullptr; *pl_i = 0; *pl_j = 0; *sh = 8;
  if (!h || !h->own8 || batch_size <= 0 || batch_size > h->cfg.max_batch || batch_offset < 0 || batch_offset + B > batch_size) return;
  if (batch_offset == 0) { h->idx8_pos = pos; h->idx8_neg = neg; h->idx8_B = batch_size; h->idx8_n = 0; }
  else if (h->idx8_n < 0 || h->idx8_B != batch_size || pos != h->idx8_pos + batch_offset || neg != h->idx8_neg + batch_offset) {
    h->idx8_n = -1;                                        // not a continuation of the batch begun at offset 0: no planes
    return;
  }
  h->idx8_n += B;
  *own8 = h->own8; *loc8 = h->loc8; *pl_i = batch_offset; *pl_j = batch_size + batch_offset; *sh = h->idx8_shift;
}

extern "C" int bprx_sample_epoch_h(bprx_handle *h, const int64_t *indptr, const int32_t *items_sorted, const int32_t *perm,
                                   const int64_t *epoch_ptr, const int32_t *pos_slot, int32_t num_users, int32_t num_items,
                                   uint64_t seed, uint32_t epoch, int64_t first, int64_t B, int32_t *user, int32_t *pos, int32_t *neg,
                                   int64_t batch_offset, int64_t batch_size, void *stream) {
  if (!indptr || !items_sorted || !perm || !epoch_ptr || !user || !pos || !neg || num_users <= 0 || num_items <= 0 ||
      B < 0 || first < 0)
    return BPRX_E_INVALID;
  if (B == 0) return BPRX_OK;
  uint8_t *own8, *loc8;
  long long pl_i, pl_j;
  int sh;
  plane_args(h, pos, neg, B, batch_offset, batch_size, &own8, &loc8, &pl_i, &pl_j, &sh);
  hipLaunchKernelGGL(k_sample_epoch, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, indptr,
                     items_sorted, perm, epoch_ptr, pos_slot, num_users, (uint32_t)num_items, (uint32_t)seed, (uint32_t)(seed >> 32),
                     epoch, (long long)first, (long long)B, user, pos, neg, own8, loc8, pl_i, pl_j, sh);
  return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP;
}

extern "C" int bprx_sample_epoch(const int64_t *indptr, const int32_t *items_sorted, const int32_t *perm,
                                 const int64_t *epoch_ptr, const int32_t *pos_slot, int32_t num_users, int32_t num_items, uint64_t seed,
                                 uint32_t epoch, int64_t first, int64_t B, int32_t *user, int32_t *pos, int32_t *neg,
                                 void *stream) {
  return bprx_sample_epoch_h(nullptr, indptr, items_sorted, perm, epoch_ptr, pos_slot, num_users, num_items, seed, epoch, first, B,
                             user, pos, neg, 0, 0, stream);
}

extern "C" int bprx_sample_philox_h(bprx_handle *h, const int64_t *indptr, const int32_t *items_sorted, const int32_t *pos_user,
                                    int64_t num_pos, int32_t num_items, uint64_t seed, uint64_t first, int64_t B, int32_t *user,
                                    int32_t *pos, int32_t *neg, int64_t batch_offset, int64_t batch_size, void *stream) {
  if (!indptr || !items_sorted || !pos_user || !user || !pos || !neg || num_pos <= 0 || num_items <= 0 || B < 0)
    return BPRX_E_INVALID;
  if (B == 0) return BPRX_OK;
  uint8_t *own8, *loc8;
  long long pl_i, pl_j;
  int sh;
  plane_args(h, pos, neg, B, batch_offset, batch_size, &own8, &loc8, &pl_i, &pl_j, &sh);
  hipLaunchKernelGGL(k_sample_philox, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, indptr,
                     items_sorted, pos_user, (unsigned long long)num_pos, (uint32_t)num_items, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (unsigned long long)first, (long long)B, user, pos, neg, own8, loc8, pl_i, pl_j, sh);
  return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP;
}

extern "C" int bprx_sample_philox(const int64_t *indptr, const int32_t *items_sorted, const int32_t *pos_user,
                                  int64_t num_pos, int32_t num_items, uint64_t seed, uint64_t first, int64_t B,
                                  int32_t *user, int32_t *pos, int32_t *neg, void *stream) {
  return bprx_sample_philox_h(nullptr, indptr, items_sorted, pos_user, num_pos, num_items, seed, first, B, user, pos, neg, 0, 0,
                              stream);
}
