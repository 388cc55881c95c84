// bprx_sparse.hip -- gfx950 kernels for the factor-table half of the BPR step:
//   k_score         Model.call                         BPRMF.py:55-76 / VBPR.py:59-86
//   k_triplet_grad  forward + analytic gradients       BPRMF.py:87-122 / VBPR.py:99-141 (GradientTape restated)
//   k_apply_sgd / k_adam_sparse / k_dense_update       optimizer.apply_gradients  BPRMF.py:123 / VBPR.py:142
//   k_score_block   predict_all                        BPRMF.py:78-85 / VBPR.py:88-97
//
// Layout: a group of G lanes (G = 8..64, a power of two, G*4 >= row length where possible) owns one
// (user, item[, item]) tuple; each lane moves 16 B of a factor row per load, so a row is one fully
// coalesced segment; dot products are reduced with wavefront shuffles inside the group (64-wide waves,
// groups never straddle a wave).  Everything is HBM-/L2-bound gather-scatter: no LDS, no MFMA.
//
// Batch-synchronous semantics: k_triplet_grad reads only pre-update values and ADDS per-occurrence
// gradients into zero-initialised dense staging tables (fp32 global atomics, one 16-B-per-lane row
// segment per wave instruction); the optimizer kernels then apply each touched row exactly once and
// re-zero the staging rows.
#include "bprx_internal.h"

namespace {

struct SparseArgs {
  const float *Gu, *Gi, *Bi, *Tu;
  float *dGu, *dGi, *dBi, *dTu;
  uint32_t *flagU, *flagI;
  const float *P;   // [*, PS] projections (VBPR) or nullptr
  float *W;         // [I, PS]
  float *lossb;
  int32_t *errflag;
  int U, I, k, d, PS;
  float reg;
  int item_atomics;   // 1: item-side gradients by global atomics (staging tables); 0: occurrence segments + k_item_seg
  // exclusive-row fast path (sgd): multiplicity of every row in the batch; rows used by exactly one triplet are
  // updated in place by that triplet's group (6 row transfers per triplet, no staging, no atomics, no apply pass)
  int32_t *cntU, *cntI;
  float *wGu, *wGi, *wBi, *wTu;   // writable aliases of the tables
  int fast;                       // any fast side on (k_apply_sgd then resets the counters)
  int fastU, fastI;               // per side: off for rows whose gradients are exported (staging rows)
  float lr;
  // occurrence segments (item_atomics == 0)
  const int32_t *seg_lead, *seg_nlead;   // chunk leaders (occurrence numbers) and their count
  int seg_guard;             // this step runs on index state computed ahead of it (bprx_hint_next_batch): offsets are checked
  int seg_cap;               // entries allocated (2 * max_batch): offsets are clamped to it, so that index state that does not
                             // belong to the batch (a bprx_hint_next_batch whose buffers were changed afterwards) cannot
                             // address outside the allocation
  const int32_t *seg_rank;   // [2B] rank of occurrence (role*B + b) within its item
  const int32_t *seg_ptr;    // [I]  first entry of the item's segment
  int2 *seg_ent;             // [2B] {user | role << 31, g_b}
  int32_t *hot_done;         // [I]  finished chunks of a hot item (k_item_seg's last-finisher hand-off), all-zero between steps
  // shared-row list (sgd fast path): the occurrence that marks a shared row first appends it (kind << 30 | row); the apply
  // pass then walks this list instead of every occurrence of the batch
  int32_t *slist, *slist_n;
  int use_list;
  int wg_combine;            // user-row gradients of a workgroup-wide user meet in LDS (BPRX_WG_COMBINE=0: wave-level only)
  int reg_items;             // exclusive item rows stored from the forward pass's registers (G >= 32, BPRMF)
};

constexpr int SEG_CAP = 64;   // entries of one item walked by ONE lane group; hotter items are cut into chunks of SEG_CAP
                              // (measured at Zipf(1.0) / (1.5), k_item_seg: 16 -> 137 / 155 us, 32 -> 74 / 90, 64 -> 55 / 58, 128 -> 58 / 60:
                              //  the chunks' atomics on the item's few staging lines cost more than a longer serial walk)
                              // entries, one lane group each (k_item_seg)

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
  return v;
}

__device__ __forceinline__ int clamp_idx(int v, int n, int32_t *errflag, int code) {
  if ((unsigned)v >= (unsigned)n) {
    *errflag = code;
    return v < 0 ? 0 : n - 1;
  }
  return v;
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

__device__ __forceinline__ uint16_t f2bf_s(float x) {   // round-to-nearest-even; inputs are finite
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// x_ui = Bi[i] + <Gu[u],Gi[i]> (+ <Tu[u],P[0:d]> + P[d])
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_score(SparseArgs a, const int32_t *__restrict__ user,
                                               const int32_t *__restrict__ item, int64_t B, int p_by_pair,
                                               float *__restrict__ x) {
  const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (b >= B) return;
  const int u = clamp_idx(user[b], a.U, a.errflag, 1), i = clamp_idx(item[b], a.I, a.errflag, 2);
  const float *gu = a.Gu + (size_t)u * a.k, *gi = a.Gi + (size_t)i * a.k;
  float s = 0.f;
  if (VEC) {
    for (int c = lane * 4; c < a.k; c += G * 4) {
      float4 p = ld4(gu + c), q = ld4(gi + c);
      s += p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
    }
  } else {
    for (int c = lane; c < a.k; c += G) s += gu[c] * gi[c];
  }
  s = group_sum<G>(s);
  float xv = a.Bi[i] + s;
  if (a.d > 0) {
    const float *tu = a.Tu + (size_t)u * a.d;
    const float *P = a.P + (size_t)(p_by_pair ? b : i) * a.PS;
    float t = 0.f;
    if (VEC) {
      for (int c = lane * 4; c < a.d; c += G * 4) {
        float4 p = ld4(tu + c), q = ld4(P + c);
        t += p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
      }
    } else {
      for (int c = lane; c < a.d; c += G) t += tu[c] * P[c];
    }
    t = group_sum<G>(t);
    xv = xv + t + P[a.d];
  }
  if (lane == 0) x[b] = xv;
}

__device__ __forceinline__ void atomic_add4(float *p, float4 v) {
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

__device__ __forceinline__ int clamp_quiet(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }   // as clamp_idx()

// Multiplicity of every user / item row in the batch (an item counts in both roles).  One thread per triplet, three
// int atomics (non-returning; returning on the item side when `rank` is wanted: the value returned is the rank of the
// occurrence among its item's occurrences).  The counters are reset by k_apply_sgd (users) / k_item_seg (items).
__global__ __launch_bounds__(256) void k_row_count(const int32_t *__restrict__ user, const int32_t *__restrict__ pos,
                                                   const int32_t *__restrict__ neg, int64_t B, int U, int I,
                                                   int32_t *__restrict__ cntU, int32_t *__restrict__ cntI, int doU, int doI,
                                                   int32_t *__restrict__ rank, int32_t *__restrict__ seg_cursor,
                                                   uint4 *__restrict__ zero16, size_t nzero16,
                                                   int32_t *__restrict__ ilist, int32_t *__restrict__ ilist_n, int ilist_cap,
                                                   int32_t *__restrict__ slist, int32_t *__restrict__ slist_n, int slist_cap) {
  // housekeeping that would otherwise be two hipMemsetAsync launches (5-6 us each on the trace): the segment cursor, and
  // the bf16 W image of the previous step (consumed by its backward projection), re-zeroed for k_item_seg
  if (seg_cursor && blockIdx.x == 0 && threadIdx.x == 0) { seg_cursor[0] = 0; seg_cursor[1] = 0; }
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nzero16; e += (size_t)gridDim.x * 256)
    zero16[e] = make_uint4(0, 0, 0, 0);
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = b < B;
  const int u = valid ? clamp_quiet(user[b], U) : 0, i = valid ? clamp_quiet(pos[b], I) : 0, j = valid ? clamp_quiet(neg[b], I) : 0;
  if (!rank && !ilist && !slist) {
    if (valid && doU) atomicAdd(cntU + u, 1);
    if (valid && doI) { atomicAdd(cntI + i, 1); atomicAdd(cntI + j, 1); }
    return;
  }
  if (!rank) {
    // The count atomics RETURN here (three independent ones per thread, in flight together) and the value tells what else
    // this occurrence has to do:
    //   ilist (list mode, sparse VBPR batches): it found its item's count at 0 -> it appends the item to the list of the
    //         batch's distinct items;
    //   slist (sgd exclusive-row fast path): it found a row's count at 1 -> the row is SHARED and exactly this occurrence
    //         lists it (kind << 30 | row) for the apply pass, which then walks that list instead of every occurrence.
    // The appends of a workgroup are prefix-summed and take ONE atomic per list cursor.
    __shared__ int lw[2][4], lbase[2];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    bool ai = false, aj = false, su = false, si = false, sj = false;
    // Users: neighbouring lanes with the same user (the reference's user-grouped order: ~20 triplets per user) add ONCE,
    // through the first lane of their run -- the memory-side atomics are what this kernel waits for (196 K of them at
    // B = 65 536), and a run of n finds the values old, old+1, ..., old+n-1 exactly as n single adds would have.
    const unsigned long long vm = __ballot(valid);                  // (valid lanes are a prefix of the wave: b < B)
    const int up = __shfl_up(u, 1, 64);
    const bool uhead = valid && (lane == 0 || up != u);
    const unsigned long long hm = __ballot(uhead);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);          // lanes 0 .. lane
    const int hl = (hm & le) ? 63 - __clzll((long long)(hm & le)) : lane;                 // first lane of my run
    const unsigned long long stops = (hm | ~vm) & ~(hl == 63 ? ~0ull : ((2ull << hl) - 1ull));
    const int run_end = stops ? __ffsll((long long)stops) - 1 : 64;                       // first lane after my run
    int ou = -1, oi = -1, oj = -1, hu = -1;
    // (the item atomics first, the run heads' user atomic behind them: the wait at the end of that branch covers all three)
    if (valid && doI) { oi = atomicAdd(cntI + i, 1); oj = atomicAdd(cntI + j, 1); }      // i == j: the second add returns one more
    if (valid && doU && lane == hl) hu = atomicAdd(cntU + u, run_end - hl);
    hu = __shfl(hu, hl, 64);
    if (valid && doU) ou = hu + (lane - hl);
    if (valid) {
      ai = ilist && oi == 0; aj = ilist && oj == 0;
      su = slist && ou == 1; si = slist && oi == 1; sj = slist && oj == 1;
    }
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long mi = __ballot(ai), mj = __ballot(aj), nu = __ballot(su), ni = __ballot(si), nj = __ballot(sj);
    const int pre0 = __popcll(mi & below) + __popcll(mj & below);
    const int pre1 = __popcll(nu & below) + __popcll(ni & below) + __popcll(nj & below);
    if (lane == 0) { lw[0][w] = __popcll(mi) + __popcll(mj); lw[1][w] = __popcll(nu) + __popcll(ni) + __popcll(nj); }
    __syncthreads();
    if (threadIdx.x < 2) {
      const int q = threadIdx.x, tot = lw[q][0] + lw[q][1] + lw[q][2] + lw[q][3];
      lbase[q] = tot ? atomicAdd(q == 0 ? ilist_n : slist_n, tot) : 0;
    }
    __syncthreads();
    int at0 = lbase[0] + pre0, at1 = lbase[1] + pre1;
    for (int q = 0; q < w; ++q) { at0 += lw[0][q]; at1 += lw[1][q]; }
    if (ai) { if (at0 < ilist_cap) ilist[at0] = i; ++at0; }
    if (aj) { if (at0 < ilist_cap) ilist[at0] = j; }
    if (su) { if (at1 < slist_cap) slist[at1] = u; ++at1; }
    if (si) { if (at1 < slist_cap) slist[at1] = i | (1 << 30); ++at1; }
    if (sj) { if (at1 < slist_cap) slist[at1] = j | (1 << 30); }
    return;
  }
  if (valid && doU) atomicAdd(cntU + u, 1);
  // Ranks: the workgroup's 512 occurrences are first counted per item in an LDS hash table (LDS atomics), then ONE global
  // returning atomic per distinct item and workgroup fetches the base rank.  A hot item (Zipf popularity: thousands of
  // occurrences per batch) then costs one same-address global atomic per workgroup instead of one per occurrence
  // (~15 ns each, serialised: measured 95 us for this kernel at Zipf(1.0) without the table).
  constexpr int HS = 1024;
  __shared__ int hkey[HS], hcnt[HS], hbase[HS];
  for (int t = threadIdx.x; t < HS; t += 256) { hkey[t] = -1; hcnt[t] = 0; }
  __syncthreads();
  int slot_i = 0, slot_j = 0, loc_i = 0, loc_j = 0;
  if (valid) {
    int sl = (int)(((unsigned)i * 2654435761u) >> 22);                       // 10-bit multiplicative hash
    for (;;) {
      const int prev = atomicCAS(&hkey[sl], -1, i);
      if (prev == -1 || prev == i) break;
      sl = (sl + 1) & (HS - 1);
    }
    slot_i = sl; loc_i = atomicAdd(&hcnt[sl], 1);
    sl = (int)(((unsigned)j * 2654435761u) >> 22);
    for (;;) {
      const int prev = atomicCAS(&hkey[sl], -1, j);
      if (prev == -1 || prev == j) break;
      sl = (sl + 1) & (HS - 1);
    }
    slot_j = sl; loc_j = atomicAdd(&hcnt[sl], 1);
  }
  __syncthreads();
  {
    // one returning atomic per distinct item of the workgroup: the (up to) four of a thread are all requested before the
    // first result is stored (one after the other they were four memory round trips)
    int hk[HS / 256], hb[HS / 256];
#pragma unroll
    for (int q = 0; q < HS / 256; ++q) hk[q] = hkey[threadIdx.x + q * 256];
#pragma unroll
    for (int q = 0; q < HS / 256; ++q) hb[q] = hk[q] >= 0 ? atomicAdd(cntI + hk[q], hcnt[threadIdx.x + q * 256]) : 0;
#pragma unroll
    for (int q = 0; q < HS / 256; ++q) hbase[threadIdx.x + q * 256] = hb[q];
  }
  __syncthreads();
  if (valid) {
    rank[b] = hbase[slot_i] + loc_i;
    rank[B + b] = hbase[slot_j] + loc_j;
  }
}

// Segment allocation: the rank-0 occurrence of every item of the batch reserves cnt[item] entries.  One thread per
// occurrence; the reservations of a 1024-thread workgroup are prefix-summed (shuffles + LDS) and taken with ONE atomic
// on the cursor (same-address returning atomics cost ~15 ns each: one per wave measured 32 us for 2K waves).
__global__ __launch_bounds__(1024) void k_seg_alloc(const int32_t *__restrict__ pos, const int32_t *__restrict__ neg,
                                                    int64_t B, int I, const int32_t *__restrict__ rank,
                                                    const int32_t *__restrict__ cntI, int32_t *__restrict__ seg_ptr,
                                                    int32_t *__restrict__ cursor, int32_t *__restrict__ lead) {
  __shared__ int wsum[16], lsum[16];
  __shared__ int wbase, lbase;
  const int64_t job = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int item = 0, c = 0;
  bool leads = false;                            // this occurrence leads a chunk of its item's segment (rank 0, CAP, 2 CAP, ...)
  if (job < 2 * B) {
    item = clamp_quiet(job < B ? pos[job] : neg[job - B], I);
    const int rk = rank[job];
    if (rk == 0) c = cntI[item];
    leads = rk % SEG_CAP == 0;
  }
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const unsigned long long lb = __ballot(leads);
  if (lane == 63) wsum[w] = incl;
  if (lane == 0) lsum[w] = __popcll(lb);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0, l = 0;
    for (int q = 0; q < 16; ++q) { const int v = wsum[q]; wsum[q] = t; t += v; const int lv = lsum[q]; lsum[q] = l; l += lv; }
    wbase = t ? atomicAdd(cursor, t) : 0;
    lbase = l ? atomicAdd(cursor + 1, l) : 0;
  }
  __syncthreads();
  if (c) seg_ptr[item] = wbase + wsum[w] + incl - c;
  // the chunk leaders, compacted: k_item_seg runs one lane group per listed occurrence instead of one per occurrence with
  // two thirds of the groups leaving at once (C2: 46 K leaders of 131 K occurrences)
  if (leads) lead[lbase + lsum[w] + __popcll(lb & ((1ull << lane) - 1ull))] = (int32_t)job;
}

// One group per triplet: forward scores, g = dloss/d(x+ - x-), per-occurrence gradients -> staging tables
// (or, for rows no other triplet of the batch uses, the finished sgd update straight into the table).
// SEG: the launch is in segment mode (a.item_atomics == 0); a compile-time flag so that the register-sourced backward of
// the user side (below) costs the atomic-mode instantiation nothing.
template <int G, bool VEC, bool SEG>
__device__ __forceinline__ void triplet_grad_body(const SparseArgs &a, const int32_t *__restrict__ user,
                                                  const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B) {
  const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  // `full`: every lane group of this workgroup has a triplet (all but the last workgroup): only then may the workgroup
  // meet at barriers (user-row combination below); in a partial workgroup the surplus groups leave here
  const bool full = ((int64_t)(blockIdx.x + 1) * 256) / G <= B;
  if (b >= B) return;
  // (the three index loads are issued before the first is looked at: clamp_idx's error store would otherwise order them
  //  one behind the other -- three memory round trips instead of one at the head of every wave)
  const int u_raw = user[b], i_raw = pos[b], j_raw = neg[b];
  const int u = clamp_idx(u_raw, a.U, a.errflag, 1);
  const int i = clamp_idx(i_raw, a.I, a.errflag, 2), j = clamp_idx(j_raw, a.I, a.errflag, 3);
  const int k = a.k, d = a.d;
  // ... and everything that depends on the indices alone is requested together with the rows, not after them
  const float bi = a.Bi[i], bj = a.Bi[j];
  int mulU = 0, mulI = 0, mulJ = 0;                       // (ONE uniform branch: a load under a branch of its own is waited for on the spot)
  if (a.fastU | a.fastI) { mulU = a.cntU[u]; mulI = a.cntI[i]; mulJ = a.cntI[j]; }
  int rkI = 0, rkJ = 0, spI = 0, spJ = 0;
  if (!a.item_atomics) {
    rkI = a.seg_rank[b]; rkJ = a.seg_rank[B + b]; spI = a.seg_ptr[i]; spJ = a.seg_ptr[j];
    spI += rkI; spJ += rkJ; rkI = 0; rkJ = 0;             // entry slots
    if (a.seg_guard) {
      const unsigned top = (unsigned)a.seg_cap - 1u;
      spI = (int)((unsigned)spI < top ? (unsigned)spI : top); spJ = (int)((unsigned)spJ < top ? (unsigned)spJ : top);
    }
  }
  const float *gu = a.Gu + (size_t)u * k, *gi = a.Gi + (size_t)i * k, *gj = a.Gi + (size_t)j * k;
  const float *tu = d ? a.Tu + (size_t)u * d : nullptr;
  const float *Pi = d ? a.P + (size_t)i * a.PS : nullptr, *Pj = d ? a.P + (size_t)j * a.PS : nullptr;

  // ---- forward: the un-differenced per-item scores of the reference (BPRMF.py:101-102) ----
  float si = 0.f, sj = 0.f, nrm = 0.f;   // <gu,gi>, <gu,gj>, |gu|^2+|gi|^2+|gj|^2 (+|tu|^2)
  // rows of the (usually only) forward pass, kept for the LDS-combined backward of the user side
  float4 fp = make_float4(0.f, 0.f, 0.f, 0.f), fq = fp, fr = fp, tp = fp, tq = fp, tr = fp;
  if (VEC) {
    for (int c = lane * 4; c < k; c += G * 4) {
      float4 p = ld4(gu + c), q = ld4(gi + c), r = ld4(gj + c);
      if (G >= 32) { fp = p; fq = q; fr = r; }
      si += p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
      sj += p.x * r.x + p.y * r.y + p.z * r.z + p.w * r.w;
      nrm += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w + q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w +
             r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    }
  } else {
    for (int c = lane; c < k; c += G) {
      float p = gu[c], q = gi[c], r = gj[c];
      si += p * q; sj += p * r; nrm += p * p + q * q + r * r;
    }
  }
  float ti = 0.f, tj = 0.f;
  if (d) {
    if (VEC) {
      for (int c = lane * 4; c < d; c += G * 4) {
        float4 p = ld4(tu + c), q = ld4(Pi + c), r = ld4(Pj + c);
        if (G >= 32) { tp = p; tq = q; tr = r; }
        ti += p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w;
        tj += p.x * r.x + p.y * r.y + p.z * r.z + p.w * r.w;
        nrm += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
      }
    } else {
      for (int c = lane; c < d; c += G) {
        float p = tu[c];
        ti += p * Pi[c]; tj += p * Pj[c]; nrm += p * p;
      }
    }
  }
  si = group_sum<G>(si); sj = group_sum<G>(sj); nrm = group_sum<G>(nrm);
  float xp = bi + si, xn = bj + sj;
  if (d) {
    ti = group_sum<G>(ti); tj = group_sum<G>(tj);
    xp = xp + ti + Pi[d];
    xn = xn + tj + Pj[d];
  }
  const float diff = xp - xn;
  const bool inr = (diff >= -80.0f) && (diff <= 1e8f);                 // tf.clip_by_value gradient mask
  const float cl = fminf(fmaxf(diff, -80.0f), 1e8f);
  const float z = -cl;                                                 // softplus(z), stable form
  const float sp = z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z));
  const float g = inr ? -1.0f / (1.0f + expf(diff)) : 0.f;            // -sigmoid(-diff)
  const float reg = a.reg, r2 = 2.f * reg, lr = a.lr;
  // exclusive rows: nobody else reads or writes them in this batch, so the in-place update is batch-synchronous
  const bool exU = a.fastU && mulU == 1;
  const bool exI = a.fastI && mulI == 1, exJ = a.fastI && mulJ == 1;               // i == j gives count 2: shared
  // item side: global atomics, or (segments) one 8-byte entry per occurrence (hot items are chunked later)
  const bool iaI = a.item_atomics, iaJ = a.item_atomics;
  if (lane == 0) {
    a.lossb[b] = sp + reg * (nrm + bi * bi + bj * bj * 0.1f);          // BPRMF.py:108-112 / VBPR.py:121-126
    if (a.use_list) {
      // shared rows were listed for the apply pass by k_row_count (the occurrence that found the count at one); exclusive
      // rows (finished by this group alone) reset their multiplicity here -- nobody else looks at it
      if (exU) a.cntU[u] = 0;
      if (iaI) {
        if (exI) { a.wBi[i] = bi - lr * (g + r2 * bi); a.cntI[i] = 0; }
        else atomicAdd(a.dBi + i, g + r2 * bi);
      } else a.seg_ent[spI + rkI] = make_int2(u, __float_as_int(g));
      if (iaJ) {
        if (exJ) { a.wBi[j] = bj - lr * (-g + (r2 * 0.1f) * bj); a.cntI[j] = 0; }
        else atomicAdd(a.dBi + j, -g + (r2 * 0.1f) * bj);
      } else a.seg_ent[spJ + rkJ] = make_int2((int)((unsigned)u | 0x80000000u), __float_as_int(g));
    } else {
    if (!exU) a.flagU[u] = 1u;
    if (iaI) {
      if (exI) a.wBi[i] = bi - lr * (g + r2 * bi);
      else { atomicAdd(a.dBi + i, g + r2 * bi); a.flagI[i] = 1u; }
    } else a.seg_ent[spI + rkI] = make_int2(u, __float_as_int(g));
    if (iaJ) {
      if (exJ) a.wBi[j] = bj - lr * (-g + (r2 * 0.1f) * bj);
      else { atomicAdd(a.dBi + j, -g + (r2 * 0.1f) * bj); a.flagI[j] = 1u; }
    } else a.seg_ent[spJ + rkJ] = make_int2((int)((unsigned)u | 0x80000000u), __float_as_int(g));
    }
  }
  // ---- backward: per-occurrence gradients from the same pre-update rows (L1/L2 hits) ----
  // Lane l of the group owns elements l, l+G, ...: every atomic wave-instruction then adds G CONTIGUOUS dwords per
  // row (full 64-B memory-side atomic requests).  The float4 layout of the forward pass would scatter each
  // instruction over every 4th dword and quadruple the request count (measured: 4x slower).
  float *au = a.dGu + (size_t)u * k, *ai = a.dGi + (size_t)i * k, *aj = a.dGi + (size_t)j * k;
  float *pu = a.wGu + (size_t)u * k, *pi = a.wGi + (size_t)i * k, *pj = a.wGi + (size_t)j * k;
  // User-grouped batches (the reference's order, the epoch-walk sampler): when every group of this wave works on the
  // SAME user, their user-row gradients are summed across the groups with wavefront shuffles and added once -- 64/G
  // times fewer atomic bytes on dGu/dTu.  Partners (lane ^ G, lane ^ 2G, ...) always hold the same element index c.
  bool comb = false;
  if (G < 64) {
    const int u0 = __shfl(u, 0, 64);
    comb = !exU && (__ballot(1) == ~0ull) && __all(u == u0);
  }
  const bool lead = (threadIdx.x & 63) < G;
  // ... and when all four waves of the workgroup work on that same user (runs of ~20 triplets per user in the epoch
  // order), the four wave sums meet in LDS and are added once: 256/G times fewer atomic bytes than one add per triplet
  // (at k = d = 256 a lane group IS a wave: 4x; without this the kernel is atomic-bound there).
  constexpr int WGROW = 1024;                             // floats per wave row in LDS (k + d <= WGROW)
  __shared__ int s_u[4];
  __shared__ __attribute__((aligned(16))) float s_du[4][WGROW];
  bool wgc = false;
  const int wv = threadIdx.x >> 6;
  if (full && k + d <= WGROW && a.wg_combine) {
    const int u0 = __shfl(u, 0, 64);
    const bool wave_ok = G == 64 ? !exU : comb;
    if ((threadIdx.x & 63) == 0) s_u[wv] = wave_ok ? u0 : -1 - wv;
    __syncthreads();
    wgc = s_u[0] >= 0 && s_u[0] == s_u[1] && s_u[1] == s_u[2] && s_u[2] == s_u[3];
  }
  // Segment mode with a workgroup-wide user (the common case in epoch order): nothing of the item side is left to do
  // here and the user-row gradient goes to LDS, so the backward pass needs NO second read of the rows in the
  // lane = element layout -- it is formed from the forward pass's registers (16 B per lane) and summed across the groups.
  // (wide rows only, G >= 32: at G = 16 the 27 four-byte re-reads are L1 hits and cheaper than the extra live registers:
  //  measured 38 -> 40 us on C2, 62 -> 55 us at k = d = 128, 106 -> 84 us at k = d = 256)
  // ... and, on the atomic path (sparse batches), also the finished rows of EXCLUSIVE items: one 16-B store per lane from
  // the registers instead of four 4-B re-reads and four 4-B stores (C3 shard: ~88 % of the item rows).
  const bool regs_ok = G >= 32 && VEC && k <= 4 * G && d <= 4 * G;
  const bool from_regs = regs_ok && wgc && (SEG ? !iaI : d == 0);   // (atomic path with d > 0: the W rows are written below)
  // (stored AFTER the element loop, which still re-reads the pre-update item rows for the user-side gradient)
  const bool regI = !SEG && regs_ok && a.reg_items && d == 0 && iaI && exI, regJ = !SEG && regs_ok && a.reg_items && d == 0 && iaJ && exJ;
  // ... and the gradients of SHARED item rows on that path (atomic staging): formed from the same registers and turned into
  // the lane = element layout of the atomics (full 128-B requests) through a per-group LDS row, instead of sending the whole
  // wave through the element loop below because one of its four item rows is shared (C3 shard: 12 % of the rows, 40 % of the
  // waves).  Only where the loop would otherwise be skipped (user side from registers).
  const bool shI = !SEG && regs_ok && wgc && a.reg_items && d == 0 && iaI && !exI;
  const bool shJ = !SEG && regs_ok && wgc && a.reg_items && d == 0 && iaJ && !exJ;
  const bool doneI = !iaI || regI || shI, doneJ = !iaJ || regJ || shJ;   // item rows with nothing to do in the element loop below
  if (from_regs) {
    const int c = lane * 4;
    float4 du = make_float4(g * (fq.x - fr.x) + r2 * fp.x, g * (fq.y - fr.y) + r2 * fp.y, g * (fq.z - fr.z) + r2 * fp.z,
                            g * (fq.w - fr.w) + r2 * fp.w);
    float4 dt = make_float4(g * (tq.x - tr.x) + r2 * tp.x, g * (tq.y - tr.y) + r2 * tp.y, g * (tq.z - tr.z) + r2 * tp.z,
                            g * (tq.w - tr.w) + r2 * tp.w);
#pragma unroll
    for (int o = G; o < 64; o <<= 1) {
      du.x += __shfl_xor(du.x, o, 64); du.y += __shfl_xor(du.y, o, 64); du.z += __shfl_xor(du.z, o, 64); du.w += __shfl_xor(du.w, o, 64);
      dt.x += __shfl_xor(dt.x, o, 64); dt.y += __shfl_xor(dt.y, o, 64); dt.z += __shfl_xor(dt.z, o, 64); dt.w += __shfl_xor(dt.w, o, 64);
    }
    if (lead && c < k) *reinterpret_cast<float4 *>(&s_du[wv][c]) = du;
    if (lead && c < d) *reinterpret_cast<float4 *>(&s_du[wv][k + c]) = dt;
  }
  // (wave-uniform trip count: the shuffles below need every lane; groups with nothing left skip their row parts)
  const bool skip_loop = from_regs && __all(doneI && doneJ);
  for (int c = lane; c < (skip_loop ? 0 : k); c += G) {
    const float p = gu[c], q = gi[c], r = gj[c];
    if (!from_regs) {
      float du = g * (q - r) + r2 * p;
      if (exU) pu[c] = p - lr * du;
      else if (wgc) {
#pragma unroll
        for (int o = G; o < 64; o <<= 1) du += __shfl_xor(du, o, 64);
        if (lead) s_du[wv][c] = du;
      } else if (comb) {
#pragma unroll
        for (int o = G; o < 64; o <<= 1) du += __shfl_xor(du, o, 64);
        if (lead) atomicAdd(au + c, du);
      } else atomicAdd(au + c, du);
    }
    if (!doneI) {
      const float di = g * p + r2 * q;
      if (exI) pi[c] = q - lr * di; else atomicAdd(ai + c, di);
    }
    if (!doneJ) {
      const float dj = -g * p + r2 * r;
      if (exJ) pj[c] = r - lr * dj; else atomicAdd(aj + c, dj);
    }
  }
  if (!SEG && regs_ok && __any(shI || shJ)) {             // wave-uniform entry; the LDS row belongs to this lane group alone
    __shared__ __attribute__((aligned(16))) float s_tr[256 / G][4 * G];
    float *row = s_tr[threadIdx.x / G];
    const int c4 = lane * 4;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const bool sh = side ? shJ : shI;
      const float4 q = side ? fr : fq;
      const float sg = side ? -g : g;
      // (LDS instructions of a wave complete in order: the reads below see what the wave's lanes stored here)
      if (sh && c4 < k)
        *reinterpret_cast<float4 *>(row + c4) = make_float4(sg * fp.x + r2 * q.x, sg * fp.y + r2 * q.y, sg * fp.z + r2 * q.z,
                                                            sg * fp.w + r2 * q.w);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float *dst = side ? aj : ai;
      if (sh) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int e = x * G + lane;
          if (e < k) atomicAdd(dst + e, reinterpret_cast<volatile float *>(row)[e]);
        }
      }
      __builtin_amdgcn_wave_barrier();                     // the row is rewritten for the other side
    }
  }
  if (regI || regJ) {
    const int c = lane * 4;
    if (regI && c < k)
      *reinterpret_cast<float4 *>(pi + c) = make_float4(fq.x - lr * (g * fp.x + r2 * fq.x), fq.y - lr * (g * fp.y + r2 * fq.y),
                                                        fq.z - lr * (g * fp.z + r2 * fq.z), fq.w - lr * (g * fp.w + r2 * fq.w));
    if (regJ && c < k)
      *reinterpret_cast<float4 *>(pj + c) = make_float4(fr.x - lr * (-g * fp.x + r2 * fr.x), fr.y - lr * (-g * fp.y + r2 * fr.y),
                                                        fr.z - lr * (-g * fp.z + r2 * fr.z), fr.w - lr * (-g * fp.w + r2 * fr.w));
  }
  if (d && !from_regs) {
    float *at = a.dTu + (size_t)u * d, *pt = a.wTu + (size_t)u * d;
    float *wi = a.W + (size_t)i * a.PS, *wj = a.W + (size_t)j * a.PS;
    for (int c = lane; c <= d; c += G) {                 // c == d: the Bp column of [theta_u | 1]
      const bool last = c == d;
      const float p = last ? 1.f : tu[c];
      if (!last) {
        float dt = g * (Pi[c] - Pj[c]) + r2 * p;
        if (exU) pt[c] = p - lr * dt;
        else if (wgc) {
#pragma unroll
          for (int o = G; o < 64; o <<= 1) dt += __shfl_xor(dt, o, 64);
          if (lead) s_du[wv][k + c] = dt;
        } else if (comb) {
#pragma unroll
          for (int o = G; o < 64; o <<= 1) dt += __shfl_xor(dt, o, 64);
          if (lead) atomicAdd(at + c, dt);
        } else atomicAdd(at + c, dt);
      }
      // W is all-zero before the step: a sole contributor stores
      if (iaI) { if (exI) wi[c] = g * p; else atomicAdd(wi + c, g * p); }
      if (iaJ) { if (exJ) wj[c] = -g * p; else atomicAdd(wj + c, -g * p); }
    }
  }
  if (wgc) {                                              // workgroup-uniform
    __syncthreads();
    const int uw = s_u[0];
    for (int e = threadIdx.x; e < k + d; e += 256) {
      const float sum = (s_du[0][e] + s_du[1][e]) + (s_du[2][e] + s_du[3][e]);
      atomicAdd(e < k ? a.dGu + (size_t)uw * k + e : a.dTu + (size_t)uw * d + (e - k), sum);
    }
  }
}

template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_triplet_grad(SparseArgs a, const int32_t *__restrict__ user,
                                                      const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B) {
  triplet_grad_body<G, VEC, false>(a, user, pos, neg, B);
}
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_triplet_grad_seg(SparseArgs a, const int32_t *__restrict__ user,
                                                          const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B) {
  triplet_grad_body<G, VEC, true>(a, user, pos, neg, B);
}

// sgd: one group per occurrence; the first to claim a touched row applies  p -= lr*dG  and re-zeroes dG.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_apply_sgd(float *Gu, float *Gi, float *Bi, float *Tu, SparseArgs a,
                                                   const int32_t *__restrict__ user, const int32_t *__restrict__ pos,
                                                   const int32_t *__restrict__ neg, int64_t B, float lr, int first_kind,
                                                   int end_kind) {
  const int64_t job = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G + (int64_t)first_kind * B;
  const int lane = threadIdx.x % G;
  if (job >= (int64_t)end_kind * B) return;
  const int kind = (int)(job / B);
  const int64_t b = job - (int64_t)kind * B;
  int row;
  uint32_t *flag;
  int32_t *cntp;
  // Users: only the first occurrence of a RUN of equal users tries the claim -- whatever the batch order, every touched user
  // has a run head, and in the reference's user-grouped order (~20 triplets per user) that is 1 claim atomic in 20
  // (memory-side atomics: 65 536 of them were most of this kernel on C2)
  bool head = true;
  if (kind == 0) {
    const int ur = user[b], up = b > 0 ? user[b - 1] : -1;
    head = b == 0 || up != ur;
    row = clamp_idx(ur, a.U, a.errflag, 1); flag = a.flagU + row; cntp = a.cntU + row;
  } else { row = clamp_idx(kind == 1 ? pos[b] : neg[b], a.I, a.errflag, 2); flag = a.flagI + row; cntp = a.cntI + row; }
  if (kind == 0 ? a.fastU : a.fastI) {                 // rows with multiplicity 1 were finished by k_triplet_grad
    const int c1 = *cntp;
    if (lane == 0 && c1) *cntp = 0;                    // reset for the next step (every job of the row may do it)
    if (c1 == 1) return;
  }
  if (!head) return;
  unsigned claimed = 0;
  if (lane == 0) claimed = atomicExch(flag, 0u);
  claimed = __shfl(claimed, 0, G);
  if (!claimed) return;
  float *p0 = kind == 0 ? Gu : Gi, *g0 = kind == 0 ? a.dGu : a.dGi;
  const int k = a.k;
  float *p = p0 + (size_t)row * k, *gr = g0 + (size_t)row * k;
  if (VEC) {
    for (int c = lane * 4; c < k; c += G * 4) {
      float4 v = ld4(p + c), gg = ld4(gr + c);
      v.x -= lr * gg.x; v.y -= lr * gg.y; v.z -= lr * gg.z; v.w -= lr * gg.w;
      *reinterpret_cast<float4 *>(p + c) = v;
      *reinterpret_cast<float4 *>(gr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int c = lane; c < k; c += G) { p[c] -= lr * gr[c]; gr[c] = 0.f; }
  }
  if (kind == 0 && a.d) {
    const int d = a.d;
    float *t = Tu + (size_t)row * d, *gt = a.dTu + (size_t)row * d;
    if (VEC) {
      for (int c = lane * 4; c < d; c += G * 4) {
        float4 v = ld4(t + c), gg = ld4(gt + c);
        v.x -= lr * gg.x; v.y -= lr * gg.y; v.z -= lr * gg.z; v.w -= lr * gg.w;
        *reinterpret_cast<float4 *>(t + c) = v;
        *reinterpret_cast<float4 *>(gt + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      for (int c = lane; c < d; c += G) { t[c] -= lr * gt[c]; gt[c] = 0.f; }
    }
  }
  if (kind != 0 && lane == 0) { Bi[row] -= lr * a.dBi[row]; a.dBi[row] = 0.f; }
}

// sgd over the SHARED-row list (a.use_list): k_row_count listed every row that more than one triplet of the batch uses;
// exclusive rows were finished (and their multiplicities reset) by their own triplet.  A fixed grid strides over the list,
// whose length is only known on the device; block 0 clears the cursor of the NEXT step (two cursors alternate).
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_apply_sgd_list(float *Gu, float *Gi, float *Bi, float *Tu, SparseArgs a,
                                                        const int32_t *__restrict__ list, const int32_t *__restrict__ n_ptr,
                                                        int32_t *__restrict__ n_next, int cap, float lr) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_next = 0;
  int n = *n_ptr;
  n = n < cap ? n : cap;
  const int lane = threadIdx.x % G;
  const int ngroups = (int)(gridDim.x * 256 / G);
  for (int e = (int)((blockIdx.x * 256 + threadIdx.x) / G); e < n; e += ngroups) {
    const int ent = list[e], kind = ent >> 30, row = ent & 0x3fffffff;
    const int k = a.k;
    float *p = (kind == 0 ? Gu : Gi) + (size_t)row * k, *gr = (kind == 0 ? a.dGu : a.dGi) + (size_t)row * k;
    if (VEC) {
      for (int c = lane * 4; c < k; c += G * 4) {
        float4 v = ld4(p + c), gg = ld4(gr + c);
        v.x -= lr * gg.x; v.y -= lr * gg.y; v.z -= lr * gg.z; v.w -= lr * gg.w;
        *reinterpret_cast<float4 *>(p + c) = v;
        *reinterpret_cast<float4 *>(gr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      for (int c = lane; c < k; c += G) { p[c] -= lr * gr[c]; gr[c] = 0.f; }
    }
    if (kind == 0 && a.d) {
      const int d = a.d;
      float *t = Tu + (size_t)row * d, *gt = a.dTu + (size_t)row * d;
      if (VEC) {
        for (int c = lane * 4; c < d; c += G * 4) {
          float4 v = ld4(t + c), gg = ld4(gt + c);
          v.x -= lr * gg.x; v.y -= lr * gg.y; v.z -= lr * gg.z; v.w -= lr * gg.w;
          *reinterpret_cast<float4 *>(t + c) = v;
          *reinterpret_cast<float4 *>(gt + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        for (int c = lane; c < d; c += G) { t[c] -= lr * gt[c]; gt[c] = 0.f; }
      }
    }
    if (lane == 0) {
      if (kind == 0) a.cntU[row] = 0;
      else { Bi[row] -= lr * a.dBi[row]; a.dBi[row] = 0.f; a.cntI[row] = 0; }
    }
  }
}

// adam_tf23, sparse-variable rule (TF-2.3 Keras Adam is NOT lazy: every row of the table decays and moves every step):
//   m = m*b1 + g*(1-b1); v = v*b2 + g*g*(1-b2); var -= lr_t*m/(sqrt(v)+eps)     (g == 0 on untouched rows)
// One element, one step.  The whole-table sweep and the lazy catch-up replay share this function, so that a replayed
// step performs bit for bit the arithmetic the sweep would have performed.
__device__ __forceinline__ void adam_elem(float &p, float &m, float &v, float g, float b1, float b2, float lr_t, float eps) {
#pragma clang fp contract(off)   // no fused multiply-adds: the same roundings wherever this is inlined (scalar sweep, float4 replay)
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  const float mt = m * b1 + g * omb1;
  const float vt = v * b2 + (g * g) * omb2;
  m = mt; v = vt;
  p = p - lr_t * mt / (sqrtf(vt) + eps);
}

__global__ __launch_bounds__(256) void k_adam_sparse(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                                     float *__restrict__ g, size_t n, float b1, float b2, float lr_t, float eps) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    float pp = p[e], mm = m[e], vv = v[e];
    adam_elem(pp, mm, vv, g[e], b1, b2, lr_t, eps);
    p[e] = pp; m[e] = mm; v[e] = vv;
    g[e] = 0.f;
  }
}

// The four sparse-variable sweeps of a step and the clearing of the two claim-mark arrays in ONE launch (six launches of a
// few microseconds each before: a third of a batch-256 step on the reference CLI's default shapes).  Same element function.
struct AdamSweepSeg { float *p, *m, *v, *g; size_t n; };
struct AdamSweepAll { AdamSweepSeg seg[4]; uint32_t *flag[2]; size_t nflag[2]; };
__global__ __launch_bounds__(256) void k_adam_sparse_all(AdamSweepAll a, float b1, float b2, float lr_t, float eps) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const AdamSweepSeg sg = a.seg[q];
    for (size_t e = first; e < sg.n; e += stride) {
      float pp = sg.p[e], mm = sg.m[e], vv = sg.v[e];
      adam_elem(pp, mm, vv, sg.g[e], b1, b2, lr_t, eps);
      sg.p[e] = pp; sg.m[e] = mm; sg.v[e] = vv;
      sg.g[e] = 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
    for (size_t e = first; e < a.nflag[q]; e += stride) a.flag[q][e] = 0u;
}

// ------------------------------------------------------------------------------------------------------------
// LAZY-EXACT adam_tf23.  The sweep above moves 6*(U+I)*(k+d)*4 bytes per step whatever the batch.  A row that
// receives no gradient in a step still changes (m, v decay; var moves by lr_s*m/(sqrt(v)+eps)), but by a recurrence that
// depends only on the row's own (p, m, v) and on the step's lr_s -- so it can be REPLAYED later, exactly: last[row] = the
// step up to which the row is current; before a row is read (forward pass of a batch that uses it, bprx_score_*,
// bprx_sync_adam) the skipped steps last+1 .. t are replayed in registers with g = 0 through adam_elem(), the very
// function the sweep uses.  lr_s of the last ADAM_HIST steps is kept in a device ring (written by the step's catch-up
// kernel); the host forces a full catch-up sweep before the ring would wrap.  Rows with m = v = 0 (never touched) are
// fixed points of the recurrence and are skipped.
// ------------------------------------------------------------------------------------------------------------
constexpr int ADAM_HIST = 8192;

struct AdamLazy {
  float b1, b2, eps;
  const float *lr_hist;          // lr_s at lr_hist[s & (ADAM_HIST - 1)]
};

// lr_s for s = from+1 .. to without a global load per step: lane l of the G-lane group fetches lr_{from+1+l} (+ a window of
// G steps at a time), the step loop reads it with a shuffle.  (A load of lr_hist[s] inside the per-element recurrence put
// an L2 round trip into every step of the chain: 122 us for the C2 catch-up instead of ~20.)
template <int G>
struct LrWindow {
  const AdamLazy &a;
  int from, lane, base;
  float mine;
  __device__ __forceinline__ LrWindow(const AdamLazy &a_, int from_, int lane_) : a(a_), from(from_), lane(lane_), base(from_ + 1) {
    mine = a.lr_hist[(base + lane) & (ADAM_HIST - 1)];
  }
  __device__ __forceinline__ float at(int s) {            // s ascending, called by all G lanes together
    if (s - base >= G) { base += G; mine = a.lr_hist[(base + lane) & (ADAM_HIST - 1)]; }
    return __shfl(mine, s - base, G);
  }
};
template <>
struct LrWindow<1> {
  const AdamLazy &a;
  __device__ __forceinline__ LrWindow(const AdamLazy &a_, int, int) : a(a_) {}
  __device__ __forceinline__ float at(int s) { return a.lr_hist[s & (ADAM_HIST - 1)]; }
};

// replay steps from+1 .. to on up to two rows of a table pair (user: Gu + Tu; item: Gi) of n0 / n1 floats, steps OUTER
// (one lr per step for all elements).  VEC (n % 4 == 0): lane l of the G-lane group owns float4 l of each row (n <= 4G);
// else elements l, l+G, ... are handled row by row with the scalar form.
template <int G, bool VEC>
__device__ __forceinline__ void adam_replay_rows(float *p0, float *m0, float *v0, int n0, float *p1, float *m1, float *v1, int n1,
                                                 int lane, int from, int to, const AdamLazy &a) {
  if (VEC && n0 <= 4 * G && n1 <= 4 * G) {
    const int c = lane * 4;
    const bool h0 = c < n0, h1 = c < n1;
    float4 pp0 = make_float4(0.f, 0.f, 0.f, 0.f), mm0 = pp0, vv0 = pp0, pp1 = pp0, mm1 = pp0, vv1 = pp0;
    if (h0) { pp0 = ld4(p0 + c); mm0 = ld4(m0 + c); vv0 = ld4(v0 + c); }
    if (h1) { pp1 = ld4(p1 + c); mm1 = ld4(m1 + c); vv1 = ld4(v1 + c); }
    // rows whose m and v are all zero (never touched) are fixed points of the recurrence: p - lr*0/(0 + eps) == p
    const bool nz = mm0.x != 0.f || mm0.y != 0.f || mm0.z != 0.f || mm0.w != 0.f || vv0.x != 0.f || vv0.y != 0.f || vv0.z != 0.f ||
                    vv0.w != 0.f || mm1.x != 0.f || mm1.y != 0.f || mm1.z != 0.f || mm1.w != 0.f || vv1.x != 0.f || vv1.y != 0.f ||
                    vv1.z != 0.f || vv1.w != 0.f;
    const unsigned long long bal = __ballot(nz);
    const unsigned long long gm = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << ((threadIdx.x & 63) / G * G));
    if ((bal & gm) == 0ull) return;                          // group-uniform
    LrWindow<G> lw(a, from, lane);
    for (int s = from + 1; s <= to; ++s) {
      const float lr = lw.at(s);
      adam_elem(pp0.x, mm0.x, vv0.x, 0.f, a.b1, a.b2, lr, a.eps); adam_elem(pp0.y, mm0.y, vv0.y, 0.f, a.b1, a.b2, lr, a.eps);
      adam_elem(pp0.z, mm0.z, vv0.z, 0.f, a.b1, a.b2, lr, a.eps); adam_elem(pp0.w, mm0.w, vv0.w, 0.f, a.b1, a.b2, lr, a.eps);
      if (n1) {
        adam_elem(pp1.x, mm1.x, vv1.x, 0.f, a.b1, a.b2, lr, a.eps); adam_elem(pp1.y, mm1.y, vv1.y, 0.f, a.b1, a.b2, lr, a.eps);
        adam_elem(pp1.z, mm1.z, vv1.z, 0.f, a.b1, a.b2, lr, a.eps); adam_elem(pp1.w, mm1.w, vv1.w, 0.f, a.b1, a.b2, lr, a.eps);
      }
    }
    if (h0) { *reinterpret_cast<float4 *>(p0 + c) = pp0; *reinterpret_cast<float4 *>(m0 + c) = mm0; *reinterpret_cast<float4 *>(v0 + c) = vv0; }
    if (h1) { *reinterpret_cast<float4 *>(p1 + c) = pp1; *reinterpret_cast<float4 *>(m1 + c) = mm1; *reinterpret_cast<float4 *>(v1 + c) = vv1; }
    return;
  }
  for (int which = 0; which < 2; ++which) {
    float *p = which ? p1 : p0, *m = which ? m1 : m0, *v = which ? v1 : v0;
    const int n = which ? n1 : n0;
    for (int c0 = 0; c0 < n; c0 += G) {                   // all G lanes walk the steps together (shuffles), masked past the end
      const int c = c0 + lane;
      float pp = 0.f, mm = 0.f, vv = 0.f;
      if (c < n) { pp = p[c]; mm = m[c]; vv = v[c]; }
      if (G > 1) {
        const unsigned long long bal = __ballot(mm != 0.f || vv != 0.f);
        const unsigned long long gm = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << ((threadIdx.x & 63) / G * G));
        if ((bal & gm) == 0ull) continue;                    // group-uniform
      } else if (mm == 0.f && vv == 0.f) continue;
      LrWindow<G> lw(a, from, lane);
      for (int s = from + 1; s <= to; ++s) adam_elem(pp, mm, vv, 0.f, a.b1, a.b2, lw.at(s), a.eps);
      if (c < n) { p[c] = pp; m[c] = mm; v[c] = vv; }
    }
  }
}

struct AdamTables {
  float *Gu, *mGu, *vGu, *Tu, *mTu, *vTu, *Gi, *mGi, *vGi, *Bi, *mBi, *vBi;
  int32_t *lastU, *lastI;
  int U, I, k, d;
};

// all rows of one user (Gu, Tu) or one item (Gi, Bi) from step `from` to step `to`
template <int G, bool VEC>
__device__ __forceinline__ void adam_replay_kind(const AdamTables &T, bool usr, int row, int lane, int from, int to, const AdamLazy &a) {
  if (usr) {
    const size_t ok = (size_t)row * T.k, od = (size_t)row * T.d;
    adam_replay_rows<G, VEC>(T.Gu + ok, T.mGu + ok, T.vGu + ok, T.k, T.d ? T.Tu + od : nullptr, T.d ? T.mTu + od : nullptr,
                             T.d ? T.vTu + od : nullptr, T.d, lane, from, to, a);
  } else {
    const size_t ok = (size_t)row * T.k;
    adam_replay_rows<G, VEC>(T.Gi + ok, T.mGi + ok, T.vGi + ok, T.k, nullptr, nullptr, nullptr, 0, lane, from, to, a);
    if (lane == 0) adam_replay_rows<1, false>(T.Bi + row, T.mBi + row, T.vBi + row, 1, nullptr, nullptr, nullptr, 0, 0, from, to, a);
  }
}

// Before the forward pass of step t: every row the batch uses is brought to step t-1.  One lane group per occurrence
// (kind 0 user, 1 positive item, 2 negative item); the group that raises last[row] to t-1 first owns the replay, the
// others find it done (the rows are read by the NEXT kernel).  Block 0 also records lr_t of this step in the ring.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_adam_catchup(AdamTables T, AdamLazy a, const int32_t *__restrict__ user,
                                                      const int32_t *__restrict__ pos, const int32_t *__restrict__ neg,
                                                      int64_t B, int t, float lr_t, float *__restrict__ lr_hist_w) {
  if (blockIdx.x == 0 && threadIdx.x == 0) lr_hist_w[t & (ADAM_HIST - 1)] = lr_t;
  const int64_t job = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  const bool valid = job < 3 * B;
  const int kind = valid ? (int)(job / B) : 0;
  const int64_t b = valid ? job - (int64_t)kind * B : 0;
  const int row = kind == 0 ? clamp_quiet(user[b], T.U) : clamp_quiet(kind == 1 ? pos[b] : neg[b], T.I);
  int32_t *last = kind == 0 ? T.lastU + row : T.lastI + row;
  int old = t;
  if (valid && lane == 0) {
    old = *last;                                           // most occurrences find their row current (touched last step, or a
    if (old < t - 1) old = atomicMax(last, t - 1);         // sibling occurrence came first): no atomic then
  }
  // The few rows that need a replay get the WHOLE wave (lane = element), one after the other: in the reference's visiting
  // order a user's ~20 triplets are neighbours, so at most one of a wave's 64/G lane groups has work and a replay inside
  // the group would run the long recurrence with G of 64 lanes (measured on C2: 112 us for the catch-up, ALU-bound).
  const int wl = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 64 / G; ++q) {
    const int old_q = __shfl(old, q * G, 64);
    if (old_q >= t - 1) continue;                          // wave-uniform
    const int row_q = __shfl(row, q * G, 64), kind_q = __shfl(kind, q * G, 64);
    adam_replay_kind<64, false>(T, kind_q == 0, row_q, wl, old_q, t - 1, a);
  }
}

// After the gradients of step t are staged: one lane group per occurrence claims its touched row (flag), replays what is
// still missing up to t-1 (nothing after k_adam_catchup; rows touched only by OTHER ranks' batches in the replicated
// multi-GPU step were not caught up), applies step t with the staged gradient, re-zeroes the staging row, last = t.
template <int G, bool VEC>
__device__ __forceinline__ void adam_apply_row(float *p, float *m, float *v, float *g, int n, int lane, int t, float lr_t,
                                               const AdamLazy &a) {
  if (VEC) {
    for (int c = lane * 4; c < n; c += G * 4) {
      float4 pp = ld4(p + c), mm = ld4(m + c), vv = ld4(v + c);
      const float4 gg = ld4(g + c);
      adam_elem(pp.x, mm.x, vv.x, gg.x, a.b1, a.b2, lr_t, a.eps); adam_elem(pp.y, mm.y, vv.y, gg.y, a.b1, a.b2, lr_t, a.eps);
      adam_elem(pp.z, mm.z, vv.z, gg.z, a.b1, a.b2, lr_t, a.eps); adam_elem(pp.w, mm.w, vv.w, gg.w, a.b1, a.b2, lr_t, a.eps);
      *reinterpret_cast<float4 *>(p + c) = pp; *reinterpret_cast<float4 *>(m + c) = mm; *reinterpret_cast<float4 *>(v + c) = vv;
      *reinterpret_cast<float4 *>(g + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int c = lane; c < n; c += G) {
      float pp = p[c], mm = m[c], vv = v[c];
      adam_elem(pp, mm, vv, g[c], a.b1, a.b2, lr_t, a.eps);
      p[c] = pp; m[c] = mm; v[c] = vv;
      g[c] = 0.f;
    }
  }
}

template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_adam_apply_lazy(AdamTables T, AdamLazy a, float *dGu, float *dTu, float *dGi, float *dBi,
                                                         uint32_t *flagU, uint32_t *flagI, const int32_t *__restrict__ user,
                                                         const int32_t *__restrict__ pos, const int32_t *__restrict__ neg,
                                                         int64_t B, int t, float lr_t, int first_kind, int end_kind) {
  const int64_t job = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G + (int64_t)first_kind * B;
  const int lane = threadIdx.x % G;
  if (job >= (int64_t)end_kind * B) return;
  const int kind = (int)(job / B);
  const int64_t b = job - (int64_t)kind * B;
  const int raw = kind == 0 ? user[b] : (kind == 1 ? pos[b] : neg[b]);
  if (kind == 0 && b > 0 && user[b - 1] == raw) return;    // users: run heads only (see k_apply_sgd)
  const int row = clamp_quiet(raw, kind == 0 ? T.U : T.I);
  uint32_t *flag = kind == 0 ? flagU + row : flagI + row;
  unsigned claimed = 0;
  if (lane == 0) claimed = atomicExch(flag, 0u);
  claimed = __shfl(claimed, 0, G);
  if (!claimed) return;
  int32_t *last = kind == 0 ? T.lastU + row : T.lastI + row;
  const int from = *last;
  if (from < t - 1) adam_replay_kind<G, VEC>(T, kind == 0, row, lane, from, t - 1, a);   // group-uniform; rare (see above)
  if (kind == 0) {
    adam_apply_row<G, VEC>(T.Gu + (size_t)row * T.k, T.mGu + (size_t)row * T.k, T.vGu + (size_t)row * T.k, dGu + (size_t)row * T.k, T.k, lane, t, lr_t, a);
    if (T.d) adam_apply_row<G, VEC>(T.Tu + (size_t)row * T.d, T.mTu + (size_t)row * T.d, T.vTu + (size_t)row * T.d, dTu + (size_t)row * T.d, T.d, lane, t, lr_t, a);
  } else {
    adam_apply_row<G, VEC>(T.Gi + (size_t)row * T.k, T.mGi + (size_t)row * T.k, T.vGi + (size_t)row * T.k, dGi + (size_t)row * T.k, T.k, lane, t, lr_t, a);
    if (lane == 0) adam_apply_row<1, false>(T.Bi + row, T.mBi + row, T.vBi + row, dBi + row, 1, 0, t, lr_t, a);
  }
  if (lane == 0) *last = t;                                // (every lane has read `from` before: same wave, in order)
}

// Full catch-up (bprx_sync_adam; before predict_all / a snapshot; before the lr ring wraps): every row to step t.
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_adam_sync(AdamTables T, AdamLazy a, int t) {
  const int lane = threadIdx.x % G;
  const int64_t ngroups = (int64_t)gridDim.x * 256 / G;
  for (int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G; r < (int64_t)T.U + T.I; r += ngroups) {
    const bool usr = r < T.U;
    const int row = usr ? (int)r : (int)(r - T.U);
    int32_t *last = usr ? T.lastU + row : T.lastI + row;
    const int old = *last;
    if (old >= t) continue;
    adam_replay_kind<G, VEC>(T, usr, row, lane, old, t, a);
    if (lane == 0) *last = t;
  }
}

__global__ void k_fill_i32(int32_t *p, size_t n, int32_t v) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = v;
}

__global__ void k_clear_flags(uint32_t *f, size_t n) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) f[e] = 0u;
}

// Dense shared parameters E [D,d] and Bp [D]: grad = (sum of the SK split-K slabs of the backward projection, or the
// all-reduced dEp) + 2*reg*param, then sgd or the dense ApplyAdam rule
//   m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps)                      (VBPR.py:142).
// A block owns tiles of DU_KB k-rows x PS columns of the padded [D][PS] slab layout (contiguous: coalesced slab reads).
// ||E||^2+||Bp||^2 (pre-update, VBPR.py:127) leaves as one double per block in sqpart[] (summed in fixed order by
// k_loss_reduce: no atomics, reproducible).
// The last kernel of a VBPR step also does the step's housekeeping, so that no separate launch has to:
//   * bf16 features: the NEXT step's [E|Bp]^T images (chunk-major Et and fragment-major EtF, see k_cast_Et) are written
//     from the updated values through an LDS transpose -- the step needs no k_cast_Et launch;
//   * fp8 features: max|E,Bp| of the updated values goes to absmax_out (one atomicMax per block), so the next step's
//     k_cast_Et8 needs no k_absmax launch in front of it;
//   * list mode: the fp32 W rows of the listed items return to zero, their multiplicities are reset when nobody else does
//     it, and the OTHER list cursor (the one the next list-mode step appends through) is cleared.
constexpr int DU_KB = 8;
__global__ __launch_bounds__(256) void k_dense_update(float *__restrict__ E, float *__restrict__ Bp, float *mE, float *vE,
                                                      float *mBp, float *vBp, const float *__restrict__ dEp,
                                                      const float *__restrict__ part, int SK, int D, int d, int PS, int adam,
                                                      float lr_t, float reg, float b1, float b2, float eps,
                                                      double *__restrict__ sqpart, float gscale, uint16_t *__restrict__ Et,
                                                      uint16_t *__restrict__ EtF, const int32_t *__restrict__ ilist,
                                                      const int32_t *__restrict__ ilist_n, int32_t *__restrict__ ilist_n_next,
                                                      int bound, float *__restrict__ W, int32_t *__restrict__ cnt_reset,
                                                      uint32_t *__restrict__ absmax_out, uint4 *__restrict__ zero16,
                                                      size_t nzero16) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[DU_KB][288];   // PS <= 272
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  // the bf16 W image, consumed by this step's backward projection, back to zero for the next step's k_item_seg (when that
  // step's index pass -- which does it otherwise -- has already run: bprx_hint_next_batch)
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nzero16; e += (size_t)gridDim.x * 256)
    zero16[e] = make_uint4(0, 0, 0, 0);
  if (ilist_n_next && blockIdx.x == 0 && threadIdx.x == 0) *ilist_n_next = 0;
  if (ilist) {
    int n = *ilist_n;
    n = n < bound ? n : bound;
    const int per = PS / 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)n * per; e += (int64_t)gridDim.x * 256) {
      const int p = (int)(e / per), c4 = (int)(e % per);
      const int item = ilist[p];
      reinterpret_cast<float4 *>(W + (size_t)item * PS)[c4] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cnt_reset && c4 == 0) cnt_reset[item] = 0;
    }
  }
  double sq = 0.0;
  uint32_t amax = 0;                                     // bit pattern of max|new value| (monotonic for non-negative floats)
  const size_t total = (size_t)D * PS;
  const int ntile = (D + DU_KB - 1) / DU_KB;
  for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
    const int k0 = tl * DU_KB;
    // one thread per 4 consecutive columns: the slab reads are 16-B loads, eight slabs in flight per thread (4-B loads left
    // a block with 8 KB in flight: five round trips per tile; C2 10.7 us for 23 MB)
    const int PQ = PS >> 2;                              // PS % 16 == 0
    for (int q = threadIdx.x; q < DU_KB * PQ; q += 256) {
      const int kr = q / PQ, n4 = (q - kr * PQ) * 4, kk = k0 + kr;
      float nvv[4] = {0.f, 0.f, 0.f, 0.f};
      if (kk < D && n4 <= d) {
        const size_t e = (size_t)kk * PS + n4;
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
        if (part) {                                      // fused split-K reduction (single-GPU step), fixed slab order
          int sidx = 0;
          for (; sidx + 8 <= SK; sidx += 8) {            // 8 independent loads in flight, then a fixed-order sum
            float4 t[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) t[x] = ld4(part + (size_t)(sidx + x) * total + e);
#pragma unroll
            for (int x = 0; x < 8; ++x) { gs[0] += t[x].x; gs[1] += t[x].y; gs[2] += t[x].z; gs[3] += t[x].w; }
          }
          for (; sidx < SK; ++sidx) {
            const float4 t = ld4(part + (size_t)sidx * total + e);
            gs[0] += t.x; gs[1] += t.y; gs[2] += t.z; gs[3] += t.w;
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) gs[c] *= gscale;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int n = n4 + c;
            if (n <= d) gs[c] = n < d ? dEp[(size_t)kk * d + n] : dEp[(size_t)D * d + kk];
          }
        }
        // (all reads of the four elements before the first write: a store between two loads orders them -- four
        //  dependent round trips per thread otherwise)
        float *pp[4], *pm[4], *pv_[4];
        float pv[4], mo[4] = {0.f, 0.f, 0.f, 0.f}, vo[4] = {0.f, 0.f, 0.f, 0.f};
        bool on[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int n = n4 + c;
          on[c] = n <= d;
          const int nn = on[c] ? n : d;                    // (a valid address for the masked lanes)
          pp[c] = nn < d ? E + (size_t)kk * d + nn : Bp + kk;
          pm[c] = adam ? (nn < d ? mE + (size_t)kk * d + nn : mBp + kk) : nullptr;
          pv_[c] = adam ? (nn < d ? vE + (size_t)kk * d + nn : vBp + kk) : nullptr;
          pv[c] = *pp[c];
        }
        if (adam) {
#pragma unroll
          for (int c = 0; c < 4; ++c) { mo[c] = *pm[c]; vo[c] = *pv_[c]; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!on[c]) continue;
          sq += (double)pv[c] * (double)pv[c];
          const float gg = gs[c] + 2.f * reg * pv[c];
          float nv;
          if (adam) {
            const float mt = mo[c] + (gg - mo[c]) * omb1;
            const float vt = vo[c] + (gg * gg - vo[c]) * omb2;
            *pm[c] = mt; *pv_[c] = vt;
            nv = pv[c] - lr_t * mt / (sqrtf(vt) + eps);
          } else {
            nv = pv[c] - lr_t * gg;
          }
          *pp[c] = nv;
          nvv[c] = nv;
          const uint32_t av = __float_as_uint(nv) & 0x7fffffffu;
          amax = av > amax ? av : amax;
        }
      }
      if (Et) {
#pragma unroll
        for (int c = 0; c < 4; ++c) tile[kr][n4 + c] = f2bf_s(nvv[c]);
      }
    }
    if (Et) {                                            // D % 128 == 0 with bf16 features: whole tiles only
      __syncthreads();
      for (int n = threadIdx.x; n < PS; n += 256) {
        uint4 v;
        v.x = (uint32_t)tile[0][n] | ((uint32_t)tile[1][n] << 16);
        v.y = (uint32_t)tile[2][n] | ((uint32_t)tile[3][n] << 16);
        v.z = (uint32_t)tile[4][n] | ((uint32_t)tile[5][n] << 16);
        v.w = (uint32_t)tile[6][n] | ((uint32_t)tile[7][n] << 16);
        const int e = k0 & 127;                          // k0 % 8 == 0: both images take the 8 values as one 16-B piece
        *reinterpret_cast<uint4 *>(Et + ((size_t)(k0 >> 7) * PS + n) * 128 + e) = v;
        *reinterpret_cast<uint4 *>(EtF + (((((size_t)(k0 >> 7) * 4 + (e >> 5)) * (PS >> 4) + (n >> 4)) * 64) +
                                          ((e >> 3) & 3) * 16 + (n & 15)) * 8) = v;
      }
      __syncthreads();
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) sqpart[blockIdx.x] = red[0];
  if (absmax_out) {
    __shared__ uint32_t wm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(amax, o, 64); amax = v > amax ? v : amax; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t b = wm[0];
      for (int q = 1; q < 4; ++q) b = wm[q] > b ? wm[q] : b;
      if (b) atomicMax(absmax_out, b);
    }
  }
}

// loss = sum_b lossb[b] + reg*(||E||^2+||Bp||^2); fixed summation order (one block), double accumulation.
__global__ __launch_bounds__(1024) void k_loss_reduce(const float *__restrict__ lossb, int64_t B,
                                                      const double *__restrict__ sqpart, int nsq, float reg,
                                                      float *__restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < B; b += 1024) s += (double)lossb[b];
  for (int q = threadIdx.x; q < nsq; q += 1024) s += (double)reg * sqpart[q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)red[0];
}

// predict_all rows [u0,u1): out[u-u0][i] = Bi[i] + <Gu[u],Gi[i]> (+ <Tu[u],P_i[0:d]> + P_i[d]).
// One thread per (u, i); a 16x16 tile of users x items per block keeps both row sets L1-resident.
__global__ __launch_bounds__(256) void k_score_block(SparseArgs a, int u0, int u1, float *__restrict__ out) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ub = u0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i >= a.I || ub >= u1) return;
  const float *gu = a.Gu + (size_t)ub * a.k, *gi = a.Gi + (size_t)i * a.k;
  float s = 0.f;
  for (int c = 0; c < a.k; ++c) s += gu[c] * gi[c];
  float xv = a.Bi[i] + s;
  if (a.d) {
    const float *tu = a.Tu + (size_t)ub * a.d, *P = a.P + (size_t)i * a.PS;
    float t = 0.f;
    for (int c = 0; c < a.d; ++c) t += tu[c] * P[c];
    xv = xv + t + P[a.d];
  }
  out[(size_t)(ub - u0) * a.I + i] = xv;
}



// ------------------------------------------------------------------------------------------------------------
// Item-side gradients without global float atomics ("occurrence segments").
//   k_row_count    the returning count atomic gives every occurrence its rank within its item
//   k_seg_alloc    the rank-0 occurrence reserves a contiguous segment of cnt[item] entries (wave-aggregated bump)
//   k_triplet_grad writes, per occurrence, the 8-byte entry {user | role << 31, g_b} at seg_ptr[item] + rank
//   k_item_seg     one group per occurrence; the rank-0 group of an item walks the item's segment, gathers the user
//                  rows and accumulates in registers
//                     acc_g += +-g*gamma_u     acc_t += +-g*theta_u     gsum += +-g
//                  then finishes the item in one pass: the L2-regularised gradient is applied to Gi/Bi in place (sgd) or
//                  stored to the staging tables (adam), and the W row of the backward projection is written once
//                  (bf16 for the MFMA path: no conversion pass; k_row_count re-zeroed the image, so untouched rows are zero).
// The user side reads pre-update item rows in k_triplet_grad, which has completed before k_item_seg starts.
// Global float atomics moved 1032 of the 1544 B per triplet at ~1 TB/s (the chip-wide atomic rate); here the same bytes
// are plain 16-B-per-lane row gathers.  A segment longer than SEG_CAP entries (hot item) is cut into chunks of SEG_CAP,
// one lane group each (led by the occurrences of rank 0, SEG_CAP, 2 SEG_CAP, ...): bounded serial walk per group; the
// chunks' partial sums meet in the item's staging rows and the chunk that finishes last completes the item.
// ------------------------------------------------------------------------------------------------------------
// ADAM: 0 = sgd (items finished in place), 1 = adam_tf23 with whole-table sweeps (the gradient goes to the staging tables),
//       2 = lazy-exact adam_tf23: the item's Adam step is taken right here from the registers that hold its row and its
//           gradient (the row is current: k_adam_catchup ran) -- no staging round trip, no apply pass for the items.
struct AdamFuse { float *mGi, *vGi, *mBi, *vBi; int32_t *lastI; float b1, b2, eps; int t; };

template <int G, int ADAM>
__global__ __launch_bounds__(256) void k_item_seg(SparseArgs a, float *__restrict__ Gi, float *__restrict__ Bi,
                                                  float *__restrict__ Wf, uint16_t *__restrict__ Wb,
                                                  const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B,
                                                  float lr, AdamFuse af) {
  const int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  // one lane group per chunk leader (k_seg_alloc's list): rank 0 owns an ordinary item, ranks 0, CAP, 2 CAP, ... share a
  // hot one.  (BPRX_SEG_LEAD=0: one group per occurrence, the non-leaders leave)
  int64_t job = e0;
  if (a.seg_nlead) {
    if (e0 >= a.seg_nlead[0]) return;
    job = a.seg_lead[e0];
  } else if (job >= 2 * B) return;
  const int rk = a.seg_rank[job];
  const int item_raw = job < B ? pos[job] : neg[job - B];            // (requested together with the rank, not after it)
  if (rk % SEG_CAP != 0) return;
  const int item = clamp_quiet(item_raw, a.I);
  const int n = a.cntI[item];
  const int e_first = a.seg_ptr[item] + rk;
  int ns = n - rk < SEG_CAP ? n - rk : SEG_CAP;
  if (a.seg_guard && (e_first < 0 || e_first + ns > a.seg_cap)) ns = 0;   // (index state of another batch: see seg_cap)
  const int2 *ent = a.seg_ent + e_first;
  const int k = a.k, d = a.d;
  const int c4 = lane * 4;
  const bool hk = c4 < k, hd = c4 < d;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), at = ag;
  float gsum = 0.f;
  int nj = 0;
  int e = 0;
  // Row loads are unconditional (lanes past the row end re-read column 0 and are masked in the sums): with the loads
  // inside per-lane `if (hk)` / `if (hd)` regions the compiler waits for the gamma rows before it issues the theta
  // rows -- three dependent round trips per iteration instead of two.
  const int ck = hk ? c4 : 0, cd = hd ? c4 : 0;
  const float mk = hk ? 1.f : 0.f, md = hd ? 1.f : 0.f;
  const float *TuS = d ? a.Tu : a.Gu;                                 // d == 0: any valid address, md == 0
  const int ds = d ? d : k;
  // the item's own row and bias depend on `item` alone: requested now, beside the first entries, not after the loop
  const size_t og = (size_t)item * k + c4, ow = (size_t)item * a.PS;
  float4 q = ld4(Gi + (size_t)item * k + ck);
  const float pb = Bi[item];
  for (; e + 2 <= ns; e += 2) {                                       // two entries in flight
    const int2 r0 = ent[e], r1 = ent[e + 1];
    const int u0 = r0.x & 0x7fffffff, u1 = r1.x & 0x7fffffff;
    const float4 p0 = ld4(a.Gu + (size_t)u0 * k + ck), p1 = ld4(a.Gu + (size_t)u1 * k + ck);
    const float4 t0 = ld4(TuS + (size_t)u0 * ds + cd), t1 = ld4(TuS + (size_t)u1 * ds + cd);
    const float s0 = r0.x < 0 ? -__int_as_float(r0.y) : __int_as_float(r0.y);
    const float s1 = r1.x < 0 ? -__int_as_float(r1.y) : __int_as_float(r1.y);
    nj += (r0.x < 0) + (r1.x < 0);
    gsum += s0 + s1;
    const float g0 = s0 * mk, g1 = s1 * mk, h0 = s0 * md, h1 = s1 * md;
    ag.x += g0 * p0.x + g1 * p1.x; ag.y += g0 * p0.y + g1 * p1.y; ag.z += g0 * p0.z + g1 * p1.z; ag.w += g0 * p0.w + g1 * p1.w;
    at.x += h0 * t0.x + h1 * t1.x; at.y += h0 * t0.y + h1 * t1.y; at.z += h0 * t0.z + h1 * t1.z; at.w += h0 * t0.w + h1 * t1.w;
  }
  if (e < ns) {
    const int2 r0 = ent[e];
    const int u0 = r0.x & 0x7fffffff;
    const float4 p0 = ld4(a.Gu + (size_t)u0 * k + ck);
    const float4 t0 = ld4(TuS + (size_t)u0 * ds + cd);
    const float s0 = r0.x < 0 ? -__int_as_float(r0.y) : __int_as_float(r0.y);
    nj += (r0.x < 0);
    gsum += s0;
    const float g0 = s0 * mk, h0 = s0 * md;
    ag.x += g0 * p0.x; ag.y += g0 * p0.y; ag.z += g0 * p0.z; ag.w += g0 * p0.w;
    at.x += h0 * t0.x; at.y += h0 * t0.y; at.z += h0 * t0.z; at.w += h0 * t0.w;
  }
  const float r2 = 2.f * a.reg;
  const float fn = (float)ns, fj = (float)nj, fi = (float)(ns - nj);
  float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (hk) gr = make_float4(ag.x + r2 * fn * q.x, ag.y + r2 * fn * q.y, ag.z + r2 * fn * q.z, ag.w + r2 * fn * q.w);
  float gb = gsum + r2 * fi * pb + (r2 * 0.1f) * fj * pb;
  float wl = gsum;                                                    // column d of W: the Bp column of [theta_u | 1]
  if (n > SEG_CAP) {
    // ---- hot item: this group holds one chunk.  Partial sums meet in the staging rows (dGi, dBi, fp32 W); the group that
    // finishes last (counter hand-off, fences on both sides) reads the totals back and completes the item below.
    // The chunk's partial rows leave in the lane = element layout (through a per-group LDS row): an atomic wave-instruction
    // then covers contiguous dwords -- 4 memory-side requests per row of 64 floats instead of 16 with the float4 layout
    // (each 64-B line hit by four instructions); all chunks of a hot item queue on the same few lines.
    __shared__ __attribute__((aligned(16))) float s_hot[256 / G][4 * G];
    float *hrow = s_hot[threadIdx.x / G];
    float *const gdst = a.dGi + (size_t)item * k;
    if (hk) *reinterpret_cast<float4 *>(hrow + c4) = gr;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int x = 0; x < 4; ++x) { const int e = x * G + lane; if (e < k) atomicAdd(gdst + e, reinterpret_cast<volatile float *>(hrow)[e]); }
    __builtin_amdgcn_wave_barrier();
    if (d) {
      if (hd) *reinterpret_cast<float4 *>(hrow + c4) = at;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int x = 0; x < 4; ++x) { const int e = x * G + lane; if (e < d) atomicAdd(a.W + ow + e, reinterpret_cast<volatile float *>(hrow)[e]); }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { atomicAdd(a.dBi + item, gb); if (d) atomicAdd(a.W + ow + d, wl); }
    // Everything handed over here was ADDED by device-scope atomics, which execute at the memory side, and is read back by
    // atomics too: the hand-off needs the adds to have been performed before the counter moves (their acknowledgements:
    // vmcnt(0)), not a cache write-back / invalidate (two __threadfence() of ~3.5 us each per chunk before)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int done = 0;
    if (lane == 0) done = atomicAdd(a.hot_done + item, 1);
    done = __shfl(done, 0, G);
    if (done + 1 != (n + SEG_CAP - 1) / SEG_CAP) return;             // not the last chunk of this item
    asm volatile("" ::: "memory");
    // totals, read where the atomics live (memory side), in the same layout and turned back through the LDS row
#pragma unroll
    for (int x = 0; x < 4; ++x) { const int e = x * G + lane; if (e < k) reinterpret_cast<volatile float *>(hrow)[e] = atomicAdd(gdst + e, 0.f); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (hk) { const volatile float *hv = hrow + c4; gr = make_float4(hv[0], hv[1], hv[2], hv[3]); }
    __builtin_amdgcn_wave_barrier();
    if (d) {
#pragma unroll
      for (int x = 0; x < 4; ++x) { const int e = x * G + lane; if (e < d) reinterpret_cast<volatile float *>(hrow)[e] = atomicAdd(a.W + ow + e, 0.f); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (hd) { const volatile float *hv = hrow + c4; at = make_float4(hv[0], hv[1], hv[2], hv[3]); }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { gb = atomicAdd(a.dBi + item, 0.f); if (d) wl = atomicAdd(a.W + ow + d, 0.f); }
    if (lane == 0) a.hot_done[item] = 0;
    if (ADAM == 0) {                                                  // staging back to zero (adam sweeps: it IS the gradient)
      if (hk) *reinterpret_cast<float4 *>(a.dGi + og) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) a.dBi[item] = 0.f;
    }
    if (d && Wb) {                                                    // the fp32 W row was only a staging row here
      if (hd) *reinterpret_cast<float4 *>(a.W + ow + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) a.W[ow + d] = 0.f;
    }
    // fall through: finish the item from the totals
    if (hk && !ADAM) *reinterpret_cast<float4 *>(Gi + og) = make_float4(q.x - lr * gr.x, q.y - lr * gr.y, q.z - lr * gr.z, q.w - lr * gr.w);
    if (lane == 0 && !ADAM) Bi[item] = pb - lr * gb;
    if (lane == 0 && ADAM == 1) a.flagI[item] = 1u;       // hot item, adam sweeps: the totals stay in the staging rows
    if (ADAM == 2) {                                      // hot item, lazy adam: staging back to zero, step taken below
      if (hk) *reinterpret_cast<float4 *>(a.dGi + og) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane == 0) a.dBi[item] = 0.f;
    }
  } else {
    if (hk) {
      if (ADAM == 1) *reinterpret_cast<float4 *>(a.dGi + og) = gr;
      else if (ADAM == 0) *reinterpret_cast<float4 *>(Gi + og) = make_float4(q.x - lr * gr.x, q.y - lr * gr.y, q.z - lr * gr.z, q.w - lr * gr.w);
    }
    if (lane == 0) {
      if (ADAM == 1) { a.dBi[item] = gb; a.flagI[item] = 1u; } else if (ADAM == 0) Bi[item] = pb - lr * gb;
    }
  }
  if (ADAM == 2) {                                        // one Adam step (lr = lr_t) on the item's rows, here and now
    if (hk) {
      float4 mm = ld4(af.mGi + og), vv = ld4(af.vGi + og);
      adam_elem(q.x, mm.x, vv.x, gr.x, af.b1, af.b2, lr, af.eps); adam_elem(q.y, mm.y, vv.y, gr.y, af.b1, af.b2, lr, af.eps);
      adam_elem(q.z, mm.z, vv.z, gr.z, af.b1, af.b2, lr, af.eps); adam_elem(q.w, mm.w, vv.w, gr.w, af.b1, af.b2, lr, af.eps);
      *reinterpret_cast<float4 *>(Gi + og) = q; *reinterpret_cast<float4 *>(af.mGi + og) = mm; *reinterpret_cast<float4 *>(af.vGi + og) = vv;
    }
    if (lane == 0) {
      float pbv = pb, mb = af.mBi[item], vb = af.vBi[item];
      adam_elem(pbv, mb, vb, gb, af.b1, af.b2, lr, af.eps);
      Bi[item] = pbv; af.mBi[item] = mb; af.vBi[item] = vb;
      af.lastI[item] = af.t;
    }
  }
  if (d) {
    if (Wb) {
      if (hd) {
        uint2 pk;
        pk.x = (uint32_t)f2bf_s(at.x) | ((uint32_t)f2bf_s(at.y) << 16);
        pk.y = (uint32_t)f2bf_s(at.z) | ((uint32_t)f2bf_s(at.w) << 16);
        *reinterpret_cast<uint2 *>(Wb + ow + c4) = pk;
      }
      if (lane == 0) Wb[ow + d] = f2bf_s(wl);
    } else if (n <= SEG_CAP) {                                        // fp32 features: W itself is the output (hot items:
      if (hd) *reinterpret_cast<float4 *>(Wf + ow + c4) = at;        // already accumulated in place)
      if (lane == 0) Wf[ow + d] = wl;
    }
  }
  if (lane == 0) a.cntI[item] = 0;                                    // reset for the next step
}

SparseArgs make_args(bprx_handle *h, const float *P) {
  SparseArgs a;
  a.Gu = h->t.Gu; a.Gi = h->t.Gi; a.Bi = h->t.Bi; a.Tu = h->t.Tu;
  a.dGu = h->dGu; a.dGi = h->dGi; a.dBi = h->dBi; a.dTu = h->dTu;
  a.flagU = h->flagU; a.flagI = h->flagI;
  a.P = P; a.W = h->W; a.lossb = h->lossb; a.errflag = h->errflag;
  a.U = h->cfg.num_users; a.I = h->cfg.num_items; a.k = h->cfg.embed_k; a.d = h->cfg.embed_d; a.PS = h->PS;
  a.reg = h->cfg.reg;
  a.cntU = h->cntU; a.cntI = h->cntI;
  a.wGu = h->t.Gu; a.wGi = h->t.Gi; a.wBi = h->t.Bi; a.wTu = h->t.Tu;
  a.fast = h->fast_rows;
  a.fastU = h->fast_rows && !(h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD);
  a.fastI = h->fast_rows && !(h->cfg.flags & BPRX_FLAG_EXPORT_ITEM_GRAD);
  a.lr = h->cfg.lr;
  a.item_atomics = h->item_mode ? 0 : 1;
  if (h->item_mode) {
    // every item row is finished by k_item_seg, which gathers PRE-update user rows after k_triplet_grad: the user
    // side may therefore not be updated in place either (staging + k_apply_sgd)
    a.fastI = 0; a.fastU = 0; a.fast = 0;
  }
  a.seg_rank = h->seg_rank; a.seg_ptr = h->seg_ptr; a.seg_ent = (int2 *)h->seg_ent; a.hot_done = h->hot_done;
  static const int seg_lead_env = getenv("BPRX_SEG_LEAD") ? atoi(getenv("BPRX_SEG_LEAD")) : 1;
  a.seg_lead = h->seg_lead; a.seg_nlead = (h->seg_cursor && seg_lead_env) ? h->seg_cursor + 1 : nullptr;
  a.seg_cap = (int)(2 * h->cfg.max_batch);
  a.seg_guard = h->idx_hinted ? 1 : 0;
  // shared-row list: both sides on the exclusive-row fast path (sgd, atomic staging, no exported gradients)
  a.use_list = (h->slist && a.fastU && a.fastI) ? 1 : 0;
  static const int reg_items_env = getenv("BPRX_REG_ITEMS") ? atoi(getenv("BPRX_REG_ITEMS")) : 1;
  a.reg_items = reg_items_env;
  static const int wg_combine_env = getenv("BPRX_WG_COMBINE") ? atoi(getenv("BPRX_WG_COMBINE")) : 1;
  a.wg_combine = wg_combine_env;
  a.slist = h->slist;
  a.slist_n = h->slist ? h->slist_n + h->slist_slot : nullptr;
  return a;
}

AdamTables make_adam_tables(bprx_handle *h) {
  AdamTables T;
  T.Gu = h->t.Gu; T.mGu = h->t.m_Gu; T.vGu = h->t.v_Gu; T.Tu = h->t.Tu; T.mTu = h->t.m_Tu; T.vTu = h->t.v_Tu;
  T.Gi = h->t.Gi; T.mGi = h->t.m_Gi; T.vGi = h->t.v_Gi; T.Bi = h->t.Bi; T.mBi = h->t.m_Bi; T.vBi = h->t.v_Bi;
  T.lastU = h->lastU; T.lastI = h->lastI;
  T.U = h->cfg.num_users; T.I = h->cfg.num_items; T.k = h->cfg.embed_k; T.d = h->cfg.embed_d;
  return T;
}

// group width: smallest power of two G in [8,64] with G*4 >= max(k,d)
int pick_group(int k, int d, bool vec) {
  int need = k > d ? k : d;
  int per = vec ? 4 : 1;
  int G = 8;
  while (G < 64 && G * per < need) G <<= 1;
  return G;
}

#define DISPATCH_G(G, VEC, KERNEL, grid, s, ...)                                               \
  do {                                                                                         \
    if (VEC) {                                                                                 \
      switch (G) {                                                                             \
        case 8: hipLaunchKernelGGL((KERNEL<8, true>), grid, dim3(256), 0, s, __VA_ARGS__); break;   \
        case 16: hipLaunchKernelGGL((KERNEL<16, true>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
        case 32: hipLaunchKernelGGL((KERNEL<32, true>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<64, true>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      }                                                                                        \
    } else {                                                                                   \
      switch (G) {                                                                             \
        case 8: hipLaunchKernelGGL((KERNEL<8, false>), grid, dim3(256), 0, s, __VA_ARGS__); break;   \
        case 16: hipLaunchKernelGGL((KERNEL<16, false>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
        case 32: hipLaunchKernelGGL((KERNEL<32, false>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL((KERNEL<64, false>), grid, dim3(256), 0, s, __VA_ARGS__); break; \
      }                                                                                        \
    }                                                                                          \
  } while (0)

inline bool vec_ok(const bprx_handle *h) {
  return h->cfg.embed_k % 4 == 0 && h->cfg.embed_d % 4 == 0;   // PS is always a multiple of 16
}

inline dim3 grid_for(int64_t groups, int G) {
  int64_t threads = groups * G;
  return dim3((unsigned)((threads + 255) / 256));
}

}  // namespace

// table[idx[r], :] += scale * rows[r, :]   (owner-side application of routed gradient rows; lane = element, so each
// atomic wave-instruction adds contiguous dwords)
__global__ __launch_bounds__(256) void k_scatter_add(float *__restrict__ table, int num_rows, int ncols,
                                                     const int32_t *__restrict__ idx, const float *__restrict__ rows,
                                                     int64_t n, float scale) {
  const int64_t total = n * ncols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / ncols;
    const int c = (int)(e - r * ncols);
    const int row = idx[r];
    if ((unsigned)row < (unsigned)num_rows) atomicAdd(table + (size_t)row * ncols + c, scale * rows[e]);
  }
}

extern "C" int bprx_scatter_add(float *table, int32_t num_rows, int32_t num_cols, const int32_t *idx, const float *rows,
                                int64_t n, float scale, void *stream) {
  if (!table || !idx || !rows || num_rows <= 0 || num_cols <= 0 || n < 0) return BPRX_E_INVALID;
  if (n == 0) return BPRX_OK;
  int64_t blocks = (n * num_cols + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, num_rows, num_cols, idx,
                     rows, n, scale);
  return hipGetLastError() == hipSuccess ? BPRX_OK : BPRX_E_HIP;
}

// ---- replicated-user multi-GPU step: message packing / application (include/bprx.h) ----
namespace {
// msg (4-byte words): [count,0,0,0 | ids[cap4] | cap*k dGu rows | cap*d dTu rows | D*d + D dense gradient], cap4 = cap
// rounded up to 4 and the whole message to a multiple of 4 words: rows are 16-byte aligned when k and d are multiples of 4
__host__ __device__ inline size_t msg_hdr(int64_t cap) { return 4 + (size_t)((cap + 3) & ~(int64_t)3); }

template <bool VEC>
__device__ __forceinline__ void row_move(float *__restrict__ dst, float *__restrict__ src, int n, int lane, bool keep) {
  constexpr int G = 16;
  if (VEC) {
    for (int c = lane * 4; c < n; c += G * 4) {
      if (keep) *reinterpret_cast<float4 *>(dst + c) = ld4(src + c);
      *reinterpret_cast<float4 *>(src + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    for (int c = lane; c < n; c += G) { if (keep) dst[c] = src[c]; src[c] = 0.f; }
  }
}

// One thread per triplet claims the triplet's user (first occurrence of a touched user owns its row); the workgroup's
// claims are compacted through LDS and take their message slots with ONE cursor atomic (a returning atomic per claimed
// user on a single address paced the first version: 43 us for 3 277 users).  Then 16 lanes move each claimed row.
// The last workgroup to finish publishes the count and re-arms the cursor.
template <bool VEC>
__global__ __launch_bounds__(256) void k_pack_user_msg(const int32_t *__restrict__ user, int64_t B, int U, int k, int d,
                                                       int cap, uint32_t *__restrict__ flagU, float *__restrict__ dGu,
                                                       float *__restrict__ dTu, float *__restrict__ msg,
                                                       int32_t *__restrict__ cursor, int32_t *__restrict__ errflag) {
  __shared__ int s_u[256];
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane64 = tid & 63, w = tid >> 6;
  const int64_t b = (int64_t)blockIdx.x * 256 + tid;
  int u = 0;
  bool claimed = false;
  if (b < B) {
    u = clamp_quiet(user[b], U);
    claimed = atomicExch(flagU + u, 0u) != 0u;
  }
  const unsigned long long bal = __ballot(claimed);
  if (lane64 == 0) s_wave[w] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { off += i < w ? s_wave[i] : 0; tot += s_wave[i]; }
  if (claimed) s_u[off + __popcll(bal & ((1ull << lane64) - 1ull))] = u;
  if (tid == 0) s_base = tot ? atomicAdd(cursor, tot) : 0;
  __syncthreads();
  const int base = s_base;
  const size_t hdr = msg_hdr(cap);
  const int lane = tid & 15;
  for (int e = tid >> 4; e < tot; e += 16) {
    const int uu = s_u[e], slot = base + e;
    const bool keep = slot < cap;
    if (keep) { if (lane == 0) reinterpret_cast<int32_t *>(msg)[4 + slot] = uu; }
    else if (lane == 0) *errflag = 4;                       // more distinct users than the message holds: reported
    const int sl = keep ? slot : 0;
    row_move<VEC>(msg + hdr + (size_t)sl * k, dGu + (size_t)uu * k, k, lane, keep);
    if (d) row_move<VEC>(msg + hdr + (size_t)cap * k + (size_t)sl * d, dTu + (size_t)uu * d, d, lane, keep);
  }
  if (tid == 0) {
    const int prev = atomicAdd(cursor + 1, 1);
    if (prev == (int)gridDim.x - 1) {
      reinterpret_cast<int32_t *>(msg)[0] = atomicExch(cursor, 0);
      cursor[1] = 0;
    }
  }
}

// Application of the gathered messages, in rank order per user whatever the number of ranks, in two launches:
// k_msg_link chains the occurrences of a user across the ranks' messages (an exchange on the user's word `head[u]`; ids are
// distinct within one message, so a chain has at most nranks links); in the second launch the occurrence that finds its
// own code in head[u] owns the user: it walks the chain, and applies the rows by ascending rank (plain read-modify-write:
// every replica performs the same additions in the same order and the replicas stay bit-identical).  head[] is the
// touched-user mark array, all-zero after k_pack_user_msg, and is returned to zero by the owners.
__device__ __forceinline__ bool msg_job(const float *__restrict__ msgs, int nranks, size_t stride, int cap, int U, int64_t job,
                                        int &r, int &slot, int &u) {
  if (job >= (int64_t)nranks * cap) return false;
  r = (int)(job / cap); slot = (int)(job - (int64_t)r * cap);
  const int32_t *m = reinterpret_cast<const int32_t *>(msgs + (size_t)r * stride);
  int cnt = m[0];
  cnt = cnt < cap ? cnt : cap;
  if (slot >= cnt) return false;
  u = m[4 + slot];
  return (unsigned)u < (unsigned)U;
}

__global__ __launch_bounds__(256) void k_msg_link(const float *__restrict__ msgs, int nranks, size_t stride, int cap, int U,
                                                  uint32_t *__restrict__ head, int32_t *__restrict__ next) {
  const int64_t job = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int r, slot, u;
  if (!msg_job(msgs, nranks, stride, cap, U, job, r, slot, u)) return;
  next[job] = (int32_t)atomicExch(head + u, (uint32_t)(job + 1));     // code = job + 1; 0 ends a chain
}

// the owner's walk: the chain's (rank -> slot) pairs by ascending rank; calls f(rank, slot) for each
template <typename F>
__device__ __forceinline__ void msg_chain_in_rank_order(const int32_t *__restrict__ next, int64_t my_job, int cap, int nranks, F f) {
  int lo = -1;                                               // ranks <= lo are done
  for (int n = 0; n < nranks; ++n) {
    int best_r = nranks, best_slot = 0;
    for (int64_t c = my_job + 1; c != 0; c = next[c - 1]) {
      const int rr = (int)((c - 1) / cap);
      if (rr > lo && rr < best_r) { best_r = rr; best_slot = (int)((c - 1) - (int64_t)rr * cap); }
    }
    if (best_r == nranks) break;
    f(best_r, best_slot);
    lo = best_r;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_apply_user_msgs(const float *__restrict__ msgs, int nranks, size_t stride, int cap,
                                                         int U, int k, int d, float *__restrict__ Gu, float *__restrict__ Tu,
                                                         float scale, uint32_t *__restrict__ head,
                                                         const int32_t *__restrict__ next) {
  constexpr int G = 16;
  const int64_t job = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  int r, slot, u;
  if (!msg_job(msgs, nranks, stride, cap, U, job, r, slot, u)) return;
  if (head[u] != (uint32_t)(job + 1)) return;                // another occurrence owns this user
  const size_t hdr = msg_hdr(cap);
  float *pg = Gu + (size_t)u * k, *pt = d ? Tu + (size_t)u * d : nullptr;
  msg_chain_in_rank_order(next, job, cap, nranks, [&](int rr, int ss) {
    const float *m = msgs + (size_t)rr * stride + hdr;
    const float *ig = m + (size_t)ss * k, *it = m + (size_t)cap * k + (size_t)ss * d;
    if (VEC) {
      for (int c = lane * 4; c < k; c += G * 4) {
        float4 p = ld4(pg + c); const float4 g = ld4(ig + c);
        p.x += scale * g.x; p.y += scale * g.y; p.z += scale * g.z; p.w += scale * g.w;
        *reinterpret_cast<float4 *>(pg + c) = p;
      }
      for (int c = lane * 4; c < d; c += G * 4) {
        float4 p = ld4(pt + c); const float4 g = ld4(it + c);
        p.x += scale * g.x; p.y += scale * g.y; p.z += scale * g.z; p.w += scale * g.w;
        *reinterpret_cast<float4 *>(pt + c) = p;
      }
    } else {
      for (int c = lane; c < k; c += G) pg[c] += scale * ig[c];
      for (int c = lane; c < d; c += G) pt[c] += scale * it[c];
    }
  });
  if (lane == 0) head[u] = 0u;
}

template <int G, bool VEC>
__device__ __forceinline__ void row_accum(float *__restrict__ dst, const float *__restrict__ src, int n, int lane) {
  if (VEC) {
    for (int c = lane * 4; c < n; c += G * 4) {
      float4 p = ld4(dst + c); const float4 g = ld4(src + c);
      p.x += g.x; p.y += g.y; p.z += g.z; p.w += g.w;
      *reinterpret_cast<float4 *>(dst + c) = p;
    }
  } else {
    for (int c = lane; c < n; c += G) dst[c] += src[c];
  }
}

// adam_tf23 in the replicated-user step: a user's gradient is the SUM over the ranks' rows (a user may sit in several
// ranks' batches): the owner of the user's chain (k_msg_link) adds the rows into the zeroed staging row by ascending
// rank -- the same additions in the same order on every replica -- and then the user takes ONE lazy-exact Adam step
// (replay of what the row missed, then step t; adam_apply_row re-zeroes the staging row).
template <int G, bool VEC>
__global__ __launch_bounds__(256) void k_adam_apply_msg_users(AdamTables T, AdamLazy a, const float *__restrict__ msgs, int nranks,
                                                              size_t stride, int cap, float *dGu, float *dTu, uint32_t *head,
                                                              const int32_t *__restrict__ next, int t, float lr_t) {
  const int64_t job = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  int r, slot, row;
  if (!msg_job(msgs, nranks, stride, cap, T.U, job, r, slot, row)) return;
  if (head[row] != (uint32_t)(job + 1)) return;
  const size_t hdr = msg_hdr(cap);
  float *gg = dGu + (size_t)row * T.k, *gt = T.d ? dTu + (size_t)row * T.d : nullptr;
  msg_chain_in_rank_order(next, job, cap, nranks, [&](int rr, int ss) {
    const float *m = msgs + (size_t)rr * stride + hdr;
    const float *ig = m + (size_t)ss * T.k, *it = m + (size_t)cap * T.k + (size_t)ss * T.d;
    row_accum<G, VEC>(gg, ig, T.k, lane);                  // lane -> element mapping of adam_apply_row: a lane reads back
    if (T.d) row_accum<G, VEC>(gt, it, T.d, lane);          // only what it wrote itself
  });
  const int from = T.lastU[row];
  if (from < t - 1) adam_replay_kind<G, VEC>(T, true, row, lane, from, t - 1, a);
  adam_apply_row<G, VEC>(T.Gu + (size_t)row * T.k, T.mGu + (size_t)row * T.k, T.vGu + (size_t)row * T.k, gg, T.k, lane, t, lr_t, a);
  if (T.d) adam_apply_row<G, VEC>(T.Tu + (size_t)row * T.d, T.mTu + (size_t)row * T.d, T.vTu + (size_t)row * T.d, gt, T.d, lane, t, lr_t, a);
  if (lane == 0) { T.lastU[row] = t; head[row] = 0u; }
}

// dEp = sum over ranks (fixed order) of the dense parts of their messages
__global__ __launch_bounds__(256) void k_sum_dense_msgs(const float *__restrict__ msgs, int nranks, size_t stride, size_t off,
                                                        size_t n, float *__restrict__ dEp) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    float s = 0.f;
    for (int r = 0; r < nranks; ++r) s += msgs[(size_t)r * stride + off + e];
    dEp[e] = s;
  }
}

}  // namespace

static inline size_t msg_rows_end(const bprx_handle *h, int64_t cap) {       // offset of the dense part
  return msg_hdr(cap) + (size_t)cap * (size_t)(h->cfg.embed_k + h->cfg.embed_d);
}

extern "C" int64_t bprx_user_msg_floats(const bprx_handle *h, int64_t cap) {
  if (!h || cap <= 0) return -1;
  const int64_t d = h->cfg.embed_d, D = h->cfg.feat_dim;
  const int64_t n = (int64_t)msg_rows_end(h, cap) + ((h->cfg.flags & BPRX_FLAG_DENSE_ALLREDUCE) ? 0 : D * d + D);
  return (n + 3) & ~(int64_t)3;
}

extern "C" int bprx_pack_user_msg(bprx_handle *h, const int32_t *user, int64_t B, int64_t cap, float *msg, void *stream) {
  if (!h || !user || !msg || B <= 0 || cap <= 0) return BPRX_E_INVALID;
  if (!(h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD)) BPRX_FAIL(h, BPRX_E_STATE, "pack_user_msg needs BPRX_FLAG_EXPORT_USER_GRAD");
  if (!h->pending_B) BPRX_FAIL(h, BPRX_E_STATE, "pack_user_msg outside a step (after bprx_step_begin[_sparse])");
  hipStream_t s = (hipStream_t)stream;
  const int k = h->cfg.embed_k, d = h->cfg.embed_d;
  const dim3 grid((unsigned)((B + 255) / 256));
  if (vec_ok(h) && ((uintptr_t)msg & 15) == 0)
    hipLaunchKernelGGL(k_pack_user_msg<true>, grid, dim3(256), 0, s, user, B, h->cfg.num_users, k, d, (int)cap, h->flagU, h->dGu,
                       h->dTu, msg, h->msg_cursor, h->errflag);
  else
    hipLaunchKernelGGL(k_pack_user_msg<false>, grid, dim3(256), 0, s, user, B, h->cfg.num_users, k, d, (int)cap, h->flagU, h->dGu,
                       h->dTu, msg, h->msg_cursor, h->errflag);
  BPRX_LAUNCH_CHECK(h, "k_pack_user_msg");
  const size_t nd = (h->cfg.flags & BPRX_FLAG_DENSE_ALLREDUCE) ? 0 : (size_t)h->cfg.feat_dim * (d + 1);
  if (nd) {
    if (h->pending_stage < 2) BPRX_FAIL(h, BPRX_E_STATE, "the message carries dE|dBp: pack it after bprx_step_begin[_dense]");
    BPRX_HIP(h, hipMemcpyAsync(msg + msg_rows_end(h, cap), h->dEp, nd * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return BPRX_OK;
}

extern "C" int bprx_apply_user_msgs(bprx_handle *h, const float *msgs, int32_t nranks, int64_t cap, float scale, void *stream) {
  if (!h || !msgs || nranks <= 0 || cap <= 0) return BPRX_E_INVALID;
  if (!h->bound) BPRX_FAIL(h, BPRX_E_STATE, "tables not bound");
  if (!(h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD)) BPRX_FAIL(h, BPRX_E_STATE, "apply_user_msgs needs BPRX_FLAG_EXPORT_USER_GRAD");
  if ((int64_t)nranks * cap >= ((int64_t)1 << 31) - 1) BPRX_FAIL(h, BPRX_E_INVALID, "nranks * cap too large");
  hipStream_t s = (hipStream_t)stream;
  const int k = h->cfg.embed_k, d = h->cfg.embed_d;
  const size_t stride = (size_t)bprx_user_msg_floats(h, cap);
  const size_t jobs = (size_t)nranks * (size_t)cap;
  if (h->msg_next_n < jobs) {                                                  // first call (or a larger world / capacity)
    BPRX_HIP(h, hipStreamSynchronize(s));
    if (h->msg_next) (void)hipFree(h->msg_next);
    h->msg_next = nullptr; h->msg_next_n = 0;
    if (hipMalloc((void **)&h->msg_next, jobs * sizeof(int32_t)) != hipSuccess) {
      (void)hipGetLastError();
      BPRX_FAIL(h, BPRX_E_NOMEM, "apply_user_msgs: chain links (%zu entries)", jobs);
    }
    h->msg_next_n = jobs;
  }
  const bool vec = vec_ok(h) && ((uintptr_t)msgs & 15) == 0;
  hipLaunchKernelGGL(k_msg_link, dim3((unsigned)((jobs + 255) / 256)), dim3(256), 0, s, msgs, (int)nranks, stride, (int)cap,
                     h->cfg.num_users, h->flagU, h->msg_next);
  if (h->cfg.optimizer == BPRX_OPT_ADAM_TF23) {
    // per touched user: the ranks' rows summed in rank order, then one lazy-exact Adam step (`scale` is sgd's -lr)
    const float tt = (float)h->adam_t;
    const float lr_t = h->cfg.lr * sqrtf(1.0f - powf(h->cfg.beta2, tt)) / (1.0f - powf(h->cfg.beta1, tt));
    const int G = pick_group(k, d, vec);
    const AdamTables T = make_adam_tables(h);
    const AdamLazy al = {h->cfg.beta1, h->cfg.beta2, h->cfg.epsilon, h->lr_hist};
    DISPATCH_G(G, vec, k_adam_apply_msg_users, grid_for((int64_t)jobs, G), s, T, al, msgs, (int)nranks, stride, (int)cap,
               h->dGu, h->dTu, h->flagU, h->msg_next, (int)h->adam_t, lr_t);
  } else if (vec) {
    hipLaunchKernelGGL(k_apply_user_msgs<true>, grid_for((int64_t)jobs, 16), dim3(256), 0, s, msgs, (int)nranks, stride, (int)cap,
                       h->cfg.num_users, k, d, h->t.Gu, h->t.Tu, scale, h->flagU, h->msg_next);
  } else {
    hipLaunchKernelGGL(k_apply_user_msgs<false>, grid_for((int64_t)jobs, 16), dim3(256), 0, s, msgs, (int)nranks, stride, (int)cap,
                       h->cfg.num_users, k, d, h->t.Gu, h->t.Tu, scale, h->flagU, h->msg_next);
  }
  const size_t nd = (h->cfg.flags & BPRX_FLAG_DENSE_ALLREDUCE) ? 0 : (size_t)h->cfg.feat_dim * (d + 1);
  if (nd) {
    unsigned blocks = (unsigned)((nd + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_sum_dense_msgs, dim3(blocks), dim3(256), 0, s, msgs, nranks, stride, msg_rows_end(h, cap), nd, h->dEp);
  }
  BPRX_LAUNCH_CHECK(h, "k_apply_user_msgs");
  return BPRX_OK;
}

// bprx_dense_grad() = sum over ranks (rank order: bit-identical replicas) of `parts` = nranks dense gradients back to back
// (an all-gather of bprx_dense_grad()), for handles created with BPRX_FLAG_DENSE_ALLREDUCE that prefer the ordered sum to
// an RCCL all-reduce
extern "C" int bprx_sum_dense_parts(bprx_handle *h, const float *parts, int32_t nranks, void *stream) {
  if (!h || !parts || nranks <= 0) return BPRX_E_INVALID;
  if (h->cfg.model != BPRX_MODEL_VBPR) BPRX_FAIL(h, BPRX_E_STATE, "sum_dense_parts: VBPR only");
  const size_t nd = (size_t)h->cfg.feat_dim * (h->cfg.embed_d + 1);
  unsigned blocks = (unsigned)((nd + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sum_dense_msgs, dim3(blocks), dim3(256), 0, (hipStream_t)stream, parts, nranks, nd, (size_t)0, nd, h->dEp);
  BPRX_LAUNCH_CHECK(h, "k_sum_dense_msgs");
  return BPRX_OK;
}

int bprx_launch_score(bprx_handle *h, const int32_t *u, const int32_t *i, int64_t B, const float *Prow, int p_by_pair,
                      float *x, hipStream_t s) {
  SparseArgs a = make_args(h, h->cfg.embed_d ? (p_by_pair ? Prow : h->P) : nullptr);
  const bool vec = vec_ok(h);
  const int G = pick_group(a.k, a.d, vec);
  DISPATCH_G(G, vec, k_score, grid_for(B, G), s, a, u, i, B, p_by_pair, x);
  BPRX_LAUNCH_CHECK(h, "k_score");
  return BPRX_OK;
}

// row multiplicities / ranks / segment offsets: needs only the index arrays (not P), so it may run beside the forward
// projection (bprx_step_begin)
int bprx_launch_index_pass(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, hipStream_t s) {
  SparseArgs a = make_args(h, h->P);
  if (h->fast_rows || h->item_mode || h->list_mode) {
    BprxProfScope pc(h, BPRX_PHASE_ROW_COUNT, s);
    const bool zw = h->item_mode && a.d && h->cfg.feat_dtype != BPRX_F_FP32 && !h->pf_launching;   // bf16 W image: rows of untouched items
    // (a prefetched pass runs while the previous step still writes / reads that image: its dense update zeroes it instead)
    const int64_t cap = 2 * B < (int64_t)a.I ? 2 * B : (int64_t)a.I;
    hipLaunchKernelGGL(k_row_count, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, u, i, j, B, a.U, a.I, h->cntU, h->cntI,
                       a.fastU, (a.fastI || h->list_mode) ? 1 : 0, h->item_mode ? h->seg_rank : (int32_t *)nullptr,
                       h->item_mode ? h->seg_cursor : (int32_t *)nullptr, (uint4 *)(zw ? h->Wb : nullptr),
                       zw ? (size_t)a.I * a.PS * sizeof(uint16_t) / 16 : (size_t)0,
                       h->list_mode ? h->ilist : (int32_t *)nullptr, h->list_cur, (int)cap,
                       a.use_list ? a.slist : (int32_t *)nullptr, a.slist_n, (int)(3 * h->cfg.max_batch));
  }
  if (h->item_mode) {
    BprxProfScope pc(h, BPRX_PHASE_SEG_ALLOC, s);
    hipLaunchKernelGGL(k_seg_alloc, dim3((unsigned)((2 * B + 1023) / 1024)), dim3(1024), 0, s, i, j, B, a.I, h->seg_rank, h->cntI,
                       h->seg_ptr, h->seg_cursor, h->seg_lead);
  }
  BPRX_LAUNCH_CHECK(h, "k_row_count/k_seg_alloc");
  return BPRX_OK;
}

int bprx_launch_triplet_grad(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, hipStream_t s) {
  SparseArgs a = make_args(h, h->P);
  const bool vec = vec_ok(h);
  const int G = pick_group(a.k, a.d, vec);
  BprxProfScope ps(h, BPRX_PHASE_TRIPLET, s);
  // W (fp32) must be all-zero here.  bf16 features: k_cast_W (backward variants >= 8) re-zeroes it while converting and
  // k_item_seg re-zeroes the rows it folds in, so only the remaining combinations need the memset.
  const bool bf = h->cfg.feat_dtype != BPRX_F_FP32;     // bf16 W image (bf16 and fp8 features)
  // dense form with fp32 features (or the fp32-W backward variants): W is consumed in place and cleared here, at the
  // next step; list mode returns its rows to zero itself (k_cast_W_rows) and only needs the memset after such a step
  const bool leaves_dirty = a.d && !h->list_mode && (!bf || (!h->item_mode && h->bwd_variant < 8));
  if (leaves_dirty || (a.d && h->W_dirty))
    BPRX_HIP(h, hipMemsetAsync(h->W, 0, (size_t)a.I * a.PS * sizeof(float), s));
  h->W_dirty = leaves_dirty;
  if (h->item_mode) DISPATCH_G(G, vec, k_triplet_grad_seg, grid_for(B, G), s, a, u, i, j, B);
  else DISPATCH_G(G, vec, k_triplet_grad, grid_for(B, G), s, a, u, i, j, B);
  BPRX_LAUNCH_CHECK(h, "k_triplet_grad");
  return BPRX_OK;
}

int bprx_launch_item_seg(bprx_handle *h, const int32_t *i, const int32_t *j, int64_t B, float lr_t, hipStream_t s) {
  if (!h->item_mode) return BPRX_OK;
  SparseArgs a = make_args(h, nullptr);
  const int G = pick_group(a.k, a.d, true);
  const int adam = h->cfg.optimizer == BPRX_OPT_ADAM_TF23 ? (h->adam_lazy ? 2 : 1) : 0;
  const bool bf = h->cfg.feat_dtype != BPRX_F_FP32;     // bf16 W image (bf16 and fp8 features)
  float *Wf = a.d && !bf ? h->W : nullptr;
  uint16_t *Wb = a.d && bf ? (uint16_t *)h->Wb : nullptr;
  const AdamFuse af = {h->t.m_Gi, h->t.v_Gi, h->t.m_Bi, h->t.v_Bi, h->lastI, h->cfg.beta1, h->cfg.beta2, h->cfg.epsilon, (int)h->adam_t};
  BprxProfScope ps(h, BPRX_PHASE_ITEM_SEG, s);
  // (the bf16 image was re-zeroed by k_row_count: rows of untouched items stay zero)
  const dim3 grid = grid_for(2 * B, G);
#define LAUNCH_SEG(GG)                                                                                                   \
  do {                                                                                                                   \
    if (adam == 2) hipLaunchKernelGGL((k_item_seg<GG, 2>), grid, dim3(256), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, i, j, B, lr_t, af); \
    else if (adam) hipLaunchKernelGGL((k_item_seg<GG, 1>), grid, dim3(256), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, i, j, B, lr_t, af); \
    else hipLaunchKernelGGL((k_item_seg<GG, 0>), grid, dim3(256), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, i, j, B, lr_t, af);          \
  } while (0)
  switch (G) {
    case 8: LAUNCH_SEG(8); break;
    case 16: LAUNCH_SEG(16); break;
    case 32: LAUNCH_SEG(32); break;
    default: LAUNCH_SEG(64); break;
  }
#undef LAUNCH_SEG
  BPRX_LAUNCH_CHECK(h, "k_item_seg");
  return BPRX_OK;
}

int bprx_launch_apply(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, float lr_t, hipStream_t s) {
  SparseArgs a = make_args(h, nullptr);
  const size_t U = a.U, I = a.I, k = a.k, d = a.d;
  BprxProfScope ps(h, BPRX_PHASE_APPLY, s);
  if (h->cfg.optimizer == BPRX_OPT_SGD) {
    const bool vec = vec_ok(h);
    const int G = pick_group(a.k, a.d, vec);
    // first_kind = 1 skips the user rows (their gradients are exported to the caller: BPRX_FLAG_EXPORT_USER_GRAD)
    // item rows are finished in place by k_item_seg when that mode is on: kinds [fk, ek)
    if (a.use_list) {
      int64_t blocks = (3 * B * G + 255) / 256;
      if (blocks > 1024) blocks = 1024;
      DISPATCH_G(G, vec, k_apply_sgd_list, dim3((unsigned)blocks), s, h->t.Gu, h->t.Gi, h->t.Bi, h->t.Tu, a, h->slist,
                 h->slist_n + h->slist_slot, h->slist_n + (h->slist_slot ^ 1), (int)(3 * h->cfg.max_batch), lr_t);
      h->slist_slot ^= 1;
      BPRX_LAUNCH_CHECK(h, "k_apply_sgd_list");
      return BPRX_OK;
    }
    const int fk = (h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD) ? 1 : 0;
    const int ek = (h->item_mode || (h->cfg.flags & BPRX_FLAG_EXPORT_ITEM_GRAD)) ? 1 : 3;
    if (ek > fk)
      DISPATCH_G(G, vec, k_apply_sgd, grid_for((int64_t)(ek - fk) * B, G), s, h->t.Gu, h->t.Gi, h->t.Bi, h->t.Tu, a, u, i, j, B,
                 lr_t, fk, ek);
    BPRX_LAUNCH_CHECK(h, "k_apply_sgd");
    return BPRX_OK;
  }
  const float b1 = h->cfg.beta1, b2 = h->cfg.beta2, eps = h->cfg.epsilon;
  if (h->adam_lazy) {                                    // touched rows only (claim per occurrence); everything else is replayed later
    const bool vec = vec_ok(h);
    const int G = pick_group(a.k, a.d, vec);
    const int fk = (h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD) ? 1 : 0;      // replicated multi-GPU: users via bprx_apply_user_msgs
    const int ek = h->item_mode ? 1 : 3;                                     // segments: k_item_seg took the items' steps
    const AdamTables T = make_adam_tables(h);
    const AdamLazy al = {b1, b2, eps, h->lr_hist};
    if (ek > fk)
      DISPATCH_G(G, vec, k_adam_apply_lazy, grid_for((int64_t)(ek - fk) * B, G), s, T, al, h->dGu, h->dTu, h->dGi, h->dBi, h->flagU,
                 h->flagI, u, i, j, B, (int)h->adam_t, lr_t, fk, ek);
    BPRX_LAUNCH_CHECK(h, "k_adam_apply_lazy");
    return BPRX_OK;
  }
  // params order of BPRMF.py:117-121 / VBPR.py:132-139; the sgd claim marks are unused by adam: cleared so that a later
  // optimizer switch starts clean
  AdamSweepAll sw;
  sw.seg[0] = {h->t.Bi, h->t.m_Bi, h->t.v_Bi, h->dBi, I};
  sw.seg[1] = {h->t.Gu, h->t.m_Gu, h->t.v_Gu, h->dGu, U * k};
  sw.seg[2] = {h->t.Gi, h->t.m_Gi, h->t.v_Gi, h->dGi, I * k};
  sw.seg[3] = {h->t.Tu, h->t.m_Tu, h->t.v_Tu, h->dTu, d ? U * d : (size_t)0};
  sw.flag[0] = h->flagU; sw.nflag[0] = U; sw.flag[1] = h->flagI; sw.nflag[1] = I;
  const size_t most = U * (k > d ? k : d) > I * k ? U * (k > d ? k : d) : I * k;
  unsigned blocks = (unsigned)((most + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_adam_sparse_all, dim3(blocks), dim3(256), 0, s, sw, b1, b2, lr_t, eps);
  BPRX_LAUNCH_CHECK(h, "k_adam_sparse");
  return BPRX_OK;
}

// ---- lazy-exact adam_tf23: catch-up before the forward pass, full catch-up on demand ----
int bprx_launch_adam_catchup(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, float lr_t,
                             hipStream_t s) {
  const bool vec = vec_ok(h);
  const int G = pick_group(h->cfg.embed_k, h->cfg.embed_d, vec);
  const AdamTables T = make_adam_tables(h);
  const AdamLazy al = {h->cfg.beta1, h->cfg.beta2, h->cfg.epsilon, h->lr_hist};
  BprxProfScope ps(h, BPRX_PHASE_ADAM_CATCHUP, s);
  DISPATCH_G(G, vec, k_adam_catchup, grid_for(3 * B, G), s, T, al, u, i, j, B, (int)h->adam_t, lr_t, h->lr_hist);
  BPRX_LAUNCH_CHECK(h, "k_adam_catchup");
  return BPRX_OK;
}

// every row to step `t` (<= adam_t); no-op when nothing is pending
int bprx_launch_adam_sync(bprx_handle *h, int64_t t, hipStream_t s) {
  if (!h->adam_lazy || h->adam_synced >= t) return BPRX_OK;
  const bool vec = vec_ok(h);
  const int G = pick_group(h->cfg.embed_k, h->cfg.embed_d, vec);
  const AdamTables T = make_adam_tables(h);
  const AdamLazy al = {h->cfg.beta1, h->cfg.beta2, h->cfg.epsilon, h->lr_hist};
  const int64_t rows = (int64_t)T.U + T.I;
  int64_t blocks = (rows * G + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  DISPATCH_G(G, vec, k_adam_sync, dim3((unsigned)blocks), s, T, al, (int)t);
  BPRX_LAUNCH_CHECK(h, "k_adam_sync");
  h->adam_synced = t;
  return BPRX_OK;
}

// every row counts as current at step t (bind, resume, outside writes): nothing to replay
int bprx_launch_adam_reset(bprx_handle *h, int64_t t, hipStream_t s) {
  if (!h->adam_lazy) return BPRX_OK;
  hipLaunchKernelGGL(k_fill_i32, dim3(512), dim3(256), 0, s, h->lastU, (size_t)h->cfg.num_users, (int32_t)t);
  hipLaunchKernelGGL(k_fill_i32, dim3(512), dim3(256), 0, s, h->lastI, (size_t)h->cfg.num_items, (int32_t)t);
  BPRX_LAUNCH_CHECK(h, "k_fill_i32");
  if (s == nullptr) BPRX_HIP(h, hipStreamSynchronize(nullptr));   // control-path callers (bind, resume, tables_dirty) pass no
  h->adam_synced = t;                                             // stream: complete before work on any other stream
  return BPRX_OK;
}

int bprx_adam_hist(void) { return ADAM_HIST; }

int bprx_launch_dense_update(bprx_handle *h, float lr_t, hipStream_t s) {
  const int D = h->cfg.feat_dim;
  unsigned blocks = (unsigned)((D + DU_KB - 1) / DU_KB);
  if (blocks > BPRX_DENSE_BLOCKS) blocks = BPRX_DENSE_BLOCKS;
  h->dense_blocks = (int)blocks;
  BprxProfScope ps(h, BPRX_PHASE_DENSE, s);
  // fused_reduce: the split-K slabs are summed here (bprx_step); otherwise dEp holds the (all-reduced) gradient
  const float *part = (h->fused_reduce && h->cfg.feat_dtype != BPRX_F_FP32) ? h->part : nullptr;
  // fp8 features: the slabs hold (F*feat_scale)^T W; an all-reduced dEp was already rescaled by k_reduce_parts
  const float gscale = (part && h->cfg.feat_dtype == BPRX_F_FP8) ? 1.0f / h->cfg.feat_scale : 1.0f;
  // bf16 features: this kernel writes the next step's [E|Bp]^T images (fp8 images need the global max first: k_cast_Et8)
  const bool images = h->cfg.feat_dtype == BPRX_F_BF16 && !getenv("BPRX_NO_ET_FUSE");
  const bool lm = h->list_mode != 0;
  const int64_t bound = lm ? h->list_bound : 0;
  hipLaunchKernelGGL(k_dense_update, dim3(blocks), dim3(256), 0, s, h->t.E, h->t.Bp, h->t.m_E, h->t.v_E, h->t.m_Bp,
                     h->t.v_Bp, h->dEp, part, h->SK_step, D, h->cfg.embed_d, h->PS,
                     h->cfg.optimizer == BPRX_OPT_ADAM_TF23 ? 1 : 0, lr_t, h->cfg.reg, h->cfg.beta1, h->cfg.beta2,
                     h->cfg.epsilon, h->loss_acc, gscale, images ? (uint16_t *)h->Et : (uint16_t *)nullptr, (uint16_t *)h->EtF,
                     lm ? (const int32_t *)h->ilist : (const int32_t *)nullptr, (const int32_t *)h->list_cur,
                     lm ? h->ilist_n + (h->list_slot ^ 1) : (int32_t *)nullptr, (int)bound, h->W,
                     (lm && h->list_reset_cnt) ? h->cntI : (int32_t *)nullptr,
                     // fp8: the slot the next k_cast_Et8 reads (cleared by the last one)
                     h->cfg.feat_dtype == BPRX_F_FP8 ? (uint32_t *)h->qs + 2 + h->qs_slot : (uint32_t *)nullptr,
                     (uint4 *)(h->pf_zero_w ? h->Wb : nullptr),
                     h->pf_zero_w ? (size_t)h->cfg.num_items * h->PS * sizeof(uint16_t) / 16 : (size_t)0);
  h->pf_zero_w = false;
  BPRX_LAUNCH_CHECK(h, "k_dense_update");
  h->absmax_valid = h->cfg.feat_dtype == BPRX_F_FP8;
  if (lm) { h->list_slot ^= 1; h->list_mode = 0; }      // the step's list is consumed
  h->et_valid = images;                                 // E / Bp moved: the images were refreshed here, or are stale
  h->p_valid = false;                                   //               the item projections are stale
  return BPRX_OK;
}

int bprx_launch_loss_reduce(bprx_handle *h, int64_t B, float *loss_out, hipStream_t s) {
  BprxProfScope ps(h, BPRX_PHASE_LOSS, s);
  hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(1024), 0, s, h->lossb, B, h->loss_acc,
                     h->cfg.model == BPRX_MODEL_VBPR ? h->dense_blocks : 0, h->cfg.reg, loss_out);
  BPRX_LAUNCH_CHECK(h, "k_loss_reduce");
  return BPRX_OK;
}

int bprx_launch_score_block(bprx_handle *h, int32_t u0, int32_t u1, float *out, hipStream_t s) {
  SparseArgs a = make_args(h, h->P);
  dim3 grid((a.I + 63) / 64, (u1 - u0 + 3) / 4);
  hipLaunchKernelGGL(k_score_block, grid, dim3(256), 0, s, a, u0, u1, out);
  BPRX_LAUNCH_CHECK(h, "k_score_block");
  return BPRX_OK;
}
