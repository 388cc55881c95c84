// bprx_eval.hip -- device-side evaluation.
//   k_score_gemm   predict_all rows [u0,u1): out[u][i] = Bi[i] + <Gu[u],Gi[i]> (+ <Tu[u],P_i[0:d]> + P_i[d])
//                  (BPRMF.py:78-85 / VBPR.py:88-97) as an fp32-in / fp32-accumulate MFMA GEMM
//                  (v_mfma_f32_32x32x2_f32: exact fp32 products, k-ordered fp32 sums -- no bf16 in the ranking path)
//   k_eval_users   Evaluator._eval_by_user (Evaluator.py:82-128) from a block of score rows: exact integer rank
//                  counting, so for identical fp32 scores the metrics equal the reference's definitions bit for bit
//                  (ties included: negatives rank before the held-out items, `>=` counts against them).
// The U x I matrix is never resident: the caller walks user blocks (scores of one block: nb x I fp32 in HBM).
#include "bprx_internal.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TM = 128, TN = 128, GKC = 16, GLD = GKC + 1;   // 17-float LDS rows: conflict-free column reads

struct GemmArgs {
  const float *Gu, *Gi, *Bi, *Tu, *P;
  int U, I, k, d, PS;
};

// one K phase: acc += A[rows, 0:K] . B[cols, 0:K]^T   (A row stride lda, B row stride ldb)
__device__ __forceinline__ void gemm_phase(const float *__restrict__ A, int lda, int arow0, int arows,
                                           const float *__restrict__ Bm, int ldb, int brow0, int brows, int K,
                                           float (*As)[GLD], float (*Bs)[GLD], f32x16 (&acc)[2][2], int wr, int wc, int lane) {
  const int t = threadIdx.x;
  for (int k0 = 0; k0 < K; k0 += GKC) {
    __syncthreads();
    // 128 rows x 16 floats per operand: thread t -> row t/2, 8 floats at (t%2)*8
    {
      const int r = t >> 1, c0 = (t & 1) * 8;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int kk = k0 + c0 + x;
        As[r][c0 + x] = (r < arows && kk < K) ? A[(size_t)(arow0 + r) * lda + kk] : 0.f;
        Bs[r][c0 + x] = (r < brows && kk < K) ? Bm[(size_t)(brow0 + r) * ldb + kk] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GKC; kk += 2) {
      const int kq = kk + (lane >> 5), rr = lane & 31;
      const float a0 = As[wr * 64 + rr][kq], a1 = As[wr * 64 + 32 + rr][kq];
      const float b0 = Bs[wc * 64 + rr][kq], b1 = Bs[wc * 64 + 32 + rr][kq];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256) void k_score_gemm(GemmArgs g, int u0, int u1, float *__restrict__ out) {
  __shared__ float As[TM][GLD];
  __shared__ float Bs[TN][GLD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
  const int i0 = blockIdx.x * TN, ub = u0 + blockIdx.y * TM;
  const int arows = min(TM, u1 - ub), brows = min(TN, g.I - i0);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  gemm_phase(g.Gu, g.k, ub, arows, g.Gi, g.k, i0, brows, g.k, As, Bs, acc, wr, wc, lane);
  if (g.d) gemm_phase(g.Tu, g.d, ub, arows, g.P, g.PS, i0, brows, g.d, As, Bs, acc, wr, wc, lane);
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int ic = i0 + wc * 64 + b * 32 + (lane & 31);
    if (ic >= g.I) continue;
    const float bias = g.Bi[ic] + (g.d ? g.P[(size_t)ic * g.PS + g.d] : 0.f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ur = ub + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (ur < u1) out[(size_t)(ur - u0) * g.I + ic] = acc[a][b][r] + bias;   // same order as BPRMF.py:85: Bi + (...)
      }
  }
}

constexpr int EVMAX = 32;   // held-out items per user handled on the device (the reference's split holds out 1)

// One workgroup per user.  out[5] = hr, prec, rec, auc, ndcg (double); out[0] = -1 marks "no held-out items"
// (Evaluator.py:88-89: such users are skipped), out[0] = -2 marks "too many held-out items for the device path".
__global__ __launch_bounds__(256) void k_eval_users(const float *__restrict__ S, int u0, int I,
                                                    const int64_t *__restrict__ tr_ptr, const int32_t *__restrict__ tr_items,
                                                    const int64_t *__restrict__ ev_ptr, const int32_t *__restrict__ ev_items,
                                                    int K, double *__restrict__ out, int32_t *__restrict__ errflag) {
  __shared__ float sp[EVMAX];
  __shared__ int ev[EVMAX];
  __shared__ int cnt_all[EVMAX], cnt_sub[EVMAX];
  __shared__ int n_tr_only;
  const int u = u0 + blockIdx.x, tid = threadIdx.x;
  const float *s = S + (size_t)blockIdx.x * I;
  double *o = out + (size_t)blockIdx.x * 5;
  const int64_t e0 = ev_ptr[u];
  const int nev = (int)(ev_ptr[u + 1] - e0);
  if (nev <= 0 || nev > EVMAX) {
    if (tid < 5) o[tid] = tid == 0 ? (nev <= 0 ? -1.0 : -2.0) : 0.0;
    return;
  }
  if (tid < nev) {
    int it = ev_items[e0 + tid];
    if ((unsigned)it >= (unsigned)I) { *errflag = 5; it = 0; }      // reported by bprx_sync_check, never dereferenced
    ev[tid] = it;
    sp[tid] = s[it];
    cnt_all[tid] = 0;
    cnt_sub[tid] = 0;
  }
  if (tid == 0) n_tr_only = 0;
  __syncthreads();
  // #(all items with score >= sp_t): one sweep of the score row per held-out item (the row is L2-resident)
  for (int t = 0; t < nev; ++t) {
    const float spt = sp[t];
    int loc = 0;
    for (int i = tid; i < I; i += 256) loc += s[i] >= spt ? 1 : 0;
    if (loc) atomicAdd(&cnt_all[t], loc);
  }
  // minus the train items that are not held-out items, minus the held-out items themselves
  const int64_t t0 = tr_ptr[u];
  const int ntr = (int)(tr_ptr[u + 1] - t0);
  for (int q = tid; q < ntr; q += 256) {
    const int it = tr_items[t0 + q];
    bool is_ev = false;
    for (int t = 0; t < nev; ++t) is_ev |= ev[t] == it;
    if (is_ev || (unsigned)it >= (unsigned)I) continue;
    atomicAdd(&n_tr_only, 1);
    const float v = s[it];
    for (int t = 0; t < nev; ++t)
      if (v >= sp[t]) atomicAdd(&cnt_sub[t], 1);
  }
  __syncthreads();
  if (tid == 0) {
    long long position = 0;
    int hits = 0;
    const long long nneg = (long long)I - n_tr_only - nev;
    const long long topn = (long long)K < nneg + nev ? (long long)K : nneg + nev;
    for (int t = 0; t < nev; ++t) {
      int ge_ev = 0, before = 0;                         // held-out items with score >= sp_t; those ranked before t
      for (int q = 0; q < nev; ++q) {
        if (sp[q] >= sp[t]) ++ge_ev;
        if (q != t && (sp[q] > sp[t] || (sp[q] == sp[t] && q < t))) ++before;
      }
      const long long neg_ge = (long long)cnt_all[t] - cnt_sub[t] - ge_ev;   // Evaluator.py:96-98
      position += neg_ge;
      if (neg_ge + before < topn) ++hits;                                    // heapq.nlargest membership, :104-115
    }
    o[0] = hits > 0 ? 1.0 : 0.0;                                             // :117
    o[1] = topn > 0 ? (double)hits / (double)topn : 0.0;                     // :123
    o[2] = (double)hits / (double)nev;                                       // :126
    o[3] = 1.0 - (double)position / ((double)nneg * (double)nev);            // :100
    o[4] = position < K ? log(2.0) / log((double)position + 2.0) : 0.0;      // :120
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same metrics for an ITEM-SHARDED model (every rank holds the score columns of its own items): everything
// k_eval_users counts is additive over item shards once the held-out items' scores are known everywhere.
//   k_eval_pos     sp[user][t] = score of held-out item t where this rank owns it, 0 elsewhere   -> all-reduce(sum): exact
//   k_eval_counts  per user, over the rank's own columns: #(items >= sp_t), #(train-only items >= sp_t), #(train-only items)
//                                                                                                 -> all-reduce(sum)
//   k_eval_finish  the reference's formulas (Evaluator.py:96-126) from the summed counts: identical to k_eval_users on the
//                  concatenated score row, bit for bit.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_eval_pos(const float *__restrict__ S, int u0, int nb, int Iloc, int item_lo, int Itot,
                                                  const int64_t *__restrict__ ev_ptr, const int32_t *__restrict__ ev_items,
                                                  float *__restrict__ sp, int32_t *__restrict__ errflag) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)nb * EVMAX) return;
  const int r = (int)(e / EVMAX), t = (int)(e % EVMAX);
  const int64_t e0 = ev_ptr[u0 + r];
  const int nev = (int)(ev_ptr[u0 + r + 1] - e0);
  float v = 0.f;
  if (t < nev && nev <= EVMAX) {
    int it = ev_items[e0 + t];
    if ((unsigned)it >= (unsigned)Itot) { *errflag = 5; it = 0; }
    const int li = it - item_lo;
    if ((unsigned)li < (unsigned)Iloc) v = S[(size_t)r * Iloc + li];
  }
  sp[e] = v;
}

__global__ __launch_bounds__(256) void k_eval_counts(const float *__restrict__ S, int u0, int Iloc, int item_lo, int Itot,
                                                     const int64_t *__restrict__ tr_ptr, const int32_t *__restrict__ tr_items,
                                                     const int64_t *__restrict__ ev_ptr, const int32_t *__restrict__ ev_items,
                                                     const float *__restrict__ spg, int32_t *__restrict__ counts) {
  __shared__ float sp[EVMAX];
  __shared__ int ev[EVMAX];
  __shared__ int cnt_all[EVMAX], cnt_sub[EVMAX];
  __shared__ int n_tr_only;
  const int u = u0 + blockIdx.x, tid = threadIdx.x;
  const float *s = S + (size_t)blockIdx.x * Iloc;
  int32_t *o = counts + (size_t)blockIdx.x * (2 * EVMAX + 1);
  const int64_t e0 = ev_ptr[u];
  const int nev = (int)(ev_ptr[u + 1] - e0);
  if (nev <= 0 || nev > EVMAX) {
    for (int q = tid; q < 2 * EVMAX + 1; q += 256) o[q] = 0;
    return;
  }
  if (tid < EVMAX) { cnt_all[tid] = 0; cnt_sub[tid] = 0; }
  if (tid < nev) {
    int it = ev_items[e0 + tid];
    if ((unsigned)it >= (unsigned)Itot) it = 0;                    // (reported by k_eval_pos)
    ev[tid] = it;
    sp[tid] = spg[(size_t)blockIdx.x * EVMAX + tid];
  }
  if (tid == 0) n_tr_only = 0;
  __syncthreads();
  for (int t = 0; t < nev; ++t) {
    const float spt = sp[t];
    int loc = 0;
    for (int i = tid; i < Iloc; i += 256) loc += s[i] >= spt ? 1 : 0;
    if (loc) atomicAdd(&cnt_all[t], loc);
  }
  const int64_t t0 = tr_ptr[u];
  const int ntr = (int)(tr_ptr[u + 1] - t0);
  for (int q = tid; q < ntr; q += 256) {
    const int it = tr_items[t0 + q];
    const int li = it - item_lo;
    if ((unsigned)li >= (unsigned)Iloc) continue;                  // another rank's column (or out of range: nobody's)
    bool is_ev = false;
    for (int t = 0; t < nev; ++t) is_ev |= ev[t] == it;
    if (is_ev) continue;
    atomicAdd(&n_tr_only, 1);
    const float v = s[li];
    for (int t = 0; t < nev; ++t)
      if (v >= sp[t]) atomicAdd(&cnt_sub[t], 1);
  }
  __syncthreads();
  if (tid < EVMAX) { o[tid] = tid < nev ? cnt_all[tid] : 0; o[EVMAX + tid] = tid < nev ? cnt_sub[tid] : 0; }
  if (tid == 0) o[2 * EVMAX] = n_tr_only;
}

__global__ __launch_bounds__(256) void k_eval_finish(int u0, int nb, int Itot, const int64_t *__restrict__ ev_ptr,
                                                     const float *__restrict__ spg, const int32_t *__restrict__ counts, int K,
                                                     double *__restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nb) return;
  double *o = out + (size_t)r * 5;
  const int nev = (int)(ev_ptr[u0 + r + 1] - ev_ptr[u0 + r]);
  if (nev <= 0 || nev > EVMAX) {
    o[0] = nev <= 0 ? -1.0 : -2.0; o[1] = o[2] = o[3] = o[4] = 0.0;
    return;
  }
  const float *sp = spg + (size_t)r * EVMAX;
  const int32_t *c = counts + (size_t)r * (2 * EVMAX + 1);
  long long position = 0;
  int hits = 0;
  const long long nneg = (long long)Itot - c[2 * EVMAX] - nev;
  const long long topn = (long long)K < nneg + nev ? (long long)K : nneg + nev;
  for (int t = 0; t < nev; ++t) {
    int ge_ev = 0, before = 0;
    for (int q = 0; q < nev; ++q) {
      if (sp[q] >= sp[t]) ++ge_ev;
      if (q != t && (sp[q] > sp[t] || (sp[q] == sp[t] && q < t))) ++before;
    }
    const long long neg_ge = (long long)c[t] - c[EVMAX + t] - ge_ev;             // Evaluator.py:96-98
    position += neg_ge;
    if (neg_ge + before < topn) ++hits;                                          // heapq.nlargest membership, :104-115
  }
  o[0] = hits > 0 ? 1.0 : 0.0;
  o[1] = topn > 0 ? (double)hits / (double)topn : 0.0;
  o[2] = (double)hits / (double)nev;
  o[3] = 1.0 - (double)position / ((double)nneg * (double)nev);
  o[4] = position < K ? log(2.0) / log((double)position + 2.0) : 0.0;
}

// ------------------------------------------------------------------------------------------------------------
// Evaluator.store_recommendation on the device (Evaluator.py:225-239): per user, the train items are masked with -inf
// IN the score row (as the reference does: results[u][training_list[u]] = -np.inf) and the K largest scores are
// selected.  One workgroup per user: radix select of the K-th largest key over the row (four 8-bit histogram passes on
// the order-preserving integer image of the fp32 scores; the row is L2-resident), one collection pass, and a bitonic
// sort of the K candidates by (score descending, item ascending).
// The reference orders equal scores by numpy's unstable argsort (implementation- and CPU-dependent); rows whose output
// depends on such a tie -- equal scores among the K selected, or at the selection boundary, or fewer than K unmasked
// items -- are FLAGGED (flag[row] = 1) so that the caller can redo exactly those rows with numpy on the (masked) row.
// ------------------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 1024;

__device__ __forceinline__ uint32_t f2key(float x) {       // larger float <-> larger key; -inf is the smallest finite-order key
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void k_topk(float *__restrict__ S, int u0, int I, const int64_t *__restrict__ tr_ptr,
                                              const int32_t *__restrict__ tr_items, int K, int32_t *__restrict__ idx_out,
                                              float *__restrict__ val_out, int32_t *__restrict__ flag_out) {
  __shared__ int hist[256];
  __shared__ uint32_t s_prefix, s_mask;
  __shared__ int s_remaining, s_ngt, s_neq, s_bad;
  __shared__ uint32_t ckey[TOPK_MAX];
  __shared__ int cidx[TOPK_MAX];
  const int u = u0 + blockIdx.x, tid = threadIdx.x;
  float *s = S + (size_t)blockIdx.x * I;
  const int64_t t0 = tr_ptr[u];
  const int ntr = (int)(tr_ptr[u + 1] - t0);
  for (int q = tid; q < ntr; q += 256) {
    const int it = tr_items[t0 + q];
    if ((unsigned)it < (unsigned)I) s[it] = -INFINITY;
  }
  const int Kc = K < I ? K : I;                            // argsort()[-k:] returns min(k, I) entries
  if (tid == 0) { s_prefix = 0; s_mask = 0; s_remaining = Kc; s_ngt = 0; s_neq = 0; s_bad = K > I ? 1 : 0; }
  __syncthreads();
  for (int pass = 3; pass >= 0; --pass) {
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix, mask = s_mask;
    for (int i = tid; i < I; i += 256) {
      const uint32_t key = f2key(s[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rem = s_remaining, b = 255;
      for (; b > 0; --b) {                                  // from the largest byte value down
        if (hist[b] >= rem) break;
        rem -= hist[b];
      }
      s_remaining = rem;                                    // still wanted among the elements whose byte == b
      s_prefix = prefix | ((uint32_t)b << (8 * pass));
      s_mask = mask | (255u << (8 * pass));
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;                            // key of the K-th largest score; `want_eq` of the elements equal to
  const int want_eq = s_remaining;                          // it belong to the list, all Kc - want_eq larger ones do
  for (int i = tid; i < I; i += 256) {
    const uint32_t key = f2key(s[i]);
    if (key > kth) {
      const int p = atomicAdd(&s_ngt, 1);
      ckey[p] = key; cidx[p] = i;
    } else if (key == kth) {
      const int e = atomicAdd(&s_neq, 1);                   // more than want_eq of them: a tie straddles the boundary (flagged)
      if (e < want_eq) { ckey[Kc - want_eq + e] = key; cidx[Kc - want_eq + e] = i; }
    }
  }
  __syncthreads();
  if (tid == 0 && (s_neq != want_eq || kth == f2key(-INFINITY))) s_bad = 1;   // boundary tie / masked items reach the list
  __syncthreads();
  // bitonic sort of the Kc candidates by (key descending, item ascending), padded to a power of two with minimal keys
  int n2 = 1;
  while (n2 < Kc) n2 <<= 1;
  for (int q = Kc + tid; q < n2; q += 256) { ckey[q] = 0u; cidx[q] = 0x7fffffff; }
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int q = tid; q < n2; q += 256) {
        const int partner = q ^ stride;
        if (partner > q) {
          const bool desc = (q & size) == 0;                // first half of each `size` block sorted "better first"
          const uint32_t ka = ckey[q], kb = ckey[partner];
          const int ia = cidx[q], ib = cidx[partner];
          const bool a_better = ka > kb || (ka == kb && ia < ib);
          if (a_better != desc) { ckey[q] = kb; ckey[partner] = ka; cidx[q] = ib; cidx[partner] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int q = tid; q < Kc; q += 256) {
    if (q + 1 < Kc && ckey[q] == ckey[q + 1]) s_bad = 1;    // equal scores inside the list: numpy's order is unspecified
    idx_out[(size_t)blockIdx.x * K + q] = cidx[q];
    val_out[(size_t)blockIdx.x * K + q] = s[cidx[q]];
  }
  for (int q = Kc + tid; q < K; q += 256) { idx_out[(size_t)blockIdx.x * K + q] = -1; val_out[(size_t)blockIdx.x * K + q] = 0.f; }
  __syncthreads();
  if (tid == 0) flag_out[blockIdx.x] = s_bad;
}

}  // namespace

extern "C" int bprx_topk(bprx_handle *h, int32_t u0, int32_t u1, float *scores, const int64_t *train_ptr,
                         const int32_t *train_items, int32_t K, int32_t *idx, float *val, int32_t *flag, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (u0 < 0 || u1 > h->cfg.num_users || u0 > u1 || !scores || !train_ptr || !train_items || !idx || !val || !flag)
    BPRX_FAIL(h, BPRX_E_INVALID, "topk: bad argument");
  if (K <= 0 || K > TOPK_MAX) BPRX_FAIL(h, BPRX_E_INVALID, "topk: K=%d outside [1, %d]", K, TOPK_MAX);
  if (u0 == u1) return BPRX_OK;
  hipLaunchKernelGGL(k_topk, dim3(u1 - u0), dim3(256), 0, (hipStream_t)stream, scores, u0, h->cfg.num_items, train_ptr,
                     train_items, K, idx, val, flag);
  BPRX_LAUNCH_CHECK(h, "k_topk");
  return BPRX_OK;
}

// used by bprx_score_block when the factor widths allow the MFMA path (K step of 2)
int bprx_launch_score_gemm(bprx_handle *h, int32_t u0, int32_t u1, float *out, hipStream_t s) {
  GemmArgs g;
  g.Gu = h->t.Gu; g.Gi = h->t.Gi; g.Bi = h->t.Bi; g.Tu = h->t.Tu; g.P = h->P;
  g.U = h->cfg.num_users; g.I = h->cfg.num_items; g.k = h->cfg.embed_k; g.d = h->cfg.embed_d; g.PS = h->PS;
  dim3 grid((g.I + TN - 1) / TN, (u1 - u0 + TM - 1) / TM);
  hipLaunchKernelGGL(k_score_gemm, grid, dim3(256), 0, s, g, u0, u1, out);
  BPRX_LAUNCH_CHECK(h, "k_score_gemm");
  return BPRX_OK;
}

extern "C" int bprx_eval_users(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, const int64_t *train_ptr,
                               const int32_t *train_items, const int64_t *eval_ptr, const int32_t *eval_items, int32_t K,
                               double *out, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (u0 < 0 || u1 > h->cfg.num_users || u0 > u1 || !scores || !train_ptr || !train_items || !eval_ptr || !eval_items ||
      !out || K <= 0)
    BPRX_FAIL(h, BPRX_E_INVALID, "eval_users: bad argument");
  if (u0 == u1) return BPRX_OK;
  hipLaunchKernelGGL(k_eval_users, dim3(u1 - u0), dim3(256), 0, (hipStream_t)stream, scores, u0, h->cfg.num_items, train_ptr,
                     train_items, eval_ptr, eval_items, K, out, h->errflag);
  BPRX_LAUNCH_CHECK(h, "k_eval_users");
  return BPRX_OK;
}

// ---- item-sharded evaluation (see k_eval_pos / k_eval_counts / k_eval_finish) ----
extern "C" int bprx_eval_pos(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, int32_t item_lo, int32_t items_total,
                             const int64_t *eval_ptr, const int32_t *eval_items, float *sp, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (u0 < 0 || u0 > u1 || !scores || !eval_ptr || !eval_items || !sp || item_lo < 0 || items_total <= 0)
    BPRX_FAIL(h, BPRX_E_INVALID, "eval_pos: bad argument");
  if (u0 == u1) return BPRX_OK;
  const int nb = u1 - u0;
  hipLaunchKernelGGL(k_eval_pos, dim3((unsigned)(((int64_t)nb * EVMAX + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, u0,
                     nb, h->cfg.num_items, item_lo, items_total, eval_ptr, eval_items, sp, h->errflag);
  BPRX_LAUNCH_CHECK(h, "k_eval_pos");
  return BPRX_OK;
}

extern "C" int bprx_eval_counts(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, int32_t item_lo, int32_t items_total,
                                const int64_t *train_ptr, const int32_t *train_items, const int64_t *eval_ptr,
                                const int32_t *eval_items, const float *sp, int32_t *counts, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (u0 < 0 || u0 > u1 || !scores || !train_ptr || !train_items || !eval_ptr || !eval_items || !sp || !counts || item_lo < 0)
    BPRX_FAIL(h, BPRX_E_INVALID, "eval_counts: bad argument");
  if (u0 == u1) return BPRX_OK;
  hipLaunchKernelGGL(k_eval_counts, dim3(u1 - u0), dim3(256), 0, (hipStream_t)stream, scores, u0, h->cfg.num_items, item_lo,
                     items_total, train_ptr, train_items, eval_ptr, eval_items, sp, counts);
  BPRX_LAUNCH_CHECK(h, "k_eval_counts");
  return BPRX_OK;
}

extern "C" int bprx_eval_finish(bprx_handle *h, int32_t u0, int32_t u1, int32_t items_total, const int64_t *eval_ptr,
                                const float *sp, const int32_t *counts, int32_t K, double *out, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (u0 < 0 || u0 > u1 || !eval_ptr || !sp || !counts || !out || K <= 0 || items_total <= 0)
    BPRX_FAIL(h, BPRX_E_INVALID, "eval_finish: bad argument");
  if (u0 == u1) return BPRX_OK;
  const int nb = u1 - u0;
  hipLaunchKernelGGL(k_eval_finish, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u0, nb, items_total,
                     eval_ptr, sp, counts, K, out);
  BPRX_LAUNCH_CHECK(h, "k_eval_finish");
  return BPRX_OK;
}
