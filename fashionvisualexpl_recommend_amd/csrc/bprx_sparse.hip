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
  // exclusive-row fast path (sgd): multiplicity of every row in the batch; rows used by exactly one triplet are
  // updated in place by that triplet's group (6 row transfers per triplet, no staging, no atomics, no apply pass)
  int32_t *cntU, *cntI;
  float *wGu, *wGi, *wBi, *wTu;   // writable aliases of the tables
  int fast;                       // any fast side on (k_apply_sgd then resets the counters)
  int fastU, fastI;               // per side: off for rows whose gradients are exported (staging rows)
  float lr;
  // occurrence segments (segment mode; built by k_index_seg)
  const int4 *seg_lead;      // k_item_seg's work list: {item, first entry, entries of the chunk, entries of the item}
  const int32_t *seg_nlead;  // its length (device)
  int seg_lead_cap, seg_lead_over;   // slots in all; slots of the owners' regions (the overflow list follows)
  int seg_cap;               // entries allocated (2 * max_batch): entry offsets are clamped to it
  const int32_t *seg_rank;   // [2B] rank of occurrence (role*B + b) within its item
  const int32_t *seg_ptr;    // [I]  first entry of the item's segment
  int2 *seg_ent;             // [2B] {user key | role << 31, g_b}; the key indexes the rows below
  int seg_ent_cap;           // entries the buffer holds (stores are bounded: ranks from byte planes that do not belong to the
                             // index arrays -- a caller that changed them after sampling -- must not write outside it)
  const float *uG, *uT;      // PRE-update user rows as k_item_seg gathers them: the tables (key = user id), or the batch's
  int usG, usT;              //   uold rows (key = user slot; k_triplet_seg mode 0); row strides in floats
  int32_t *hot_done;         // [I]  finished chunks of a hot item (k_item_seg's last-finisher hand-off), all-zero between steps
  // finishing lane groups of k_item_seg (segment-mode sgd): the batch's users (k_index_seg) and their number
  const int32_t *ulist, *ulist_n;
  int nfin;                  // workgroups at the front of k_item_seg's grid that finish users (0: none)
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

// Multiplicity of every user / item row in the batch (an item counts in both roles), for the atomic-staging paths (sparse
// batches: exclusive-row fast path, touched-item list).  One thread per triplet, three int atomics.  The counters are reset
// by k_apply_sgd[_list] / k_triplet_grad / k_dense_update.  (Segment mode has its own index pass: k_index_seg.)
__global__ __launch_bounds__(256) void k_row_count(const int32_t *__restrict__ user, const int32_t *__restrict__ pos,
                                                   const int32_t *__restrict__ neg, int64_t B, int U, int I,
                                                   int32_t *__restrict__ cntU, int32_t *__restrict__ cntI, int doU, int doI,
                                                   int32_t *__restrict__ ilist, int32_t *__restrict__ ilist_n, int ilist_cap,
                                                   int32_t *__restrict__ slist, int32_t *__restrict__ slist_n, int slist_cap) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = b < B;
  const int u = valid ? clamp_quiet(user[b], U) : 0, i = valid ? clamp_quiet(pos[b], I) : 0, j = valid ? clamp_quiet(neg[b], I) : 0;
  if (!ilist && !slist) {
    if (valid && doU) atomicAdd(cntU + u, 1);
    if (valid && doI) { atomicAdd(cntI + i, 1); atomicAdd(cntI + j, 1); }
    return;
  }
  {
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
}

// One group per triplet: forward scores, g = dloss/d(x+ - x-), per-occurrence gradients -> staging tables
// (or, for rows no other triplet of the batch uses, the finished sgd update straight into the table).
// The atomic-staging form (sparse batches, BPRMF shards, exported gradients); segment mode runs k_triplet_seg instead.
// threads of a k_triplet_grad workgroup (its waves meet in LDS when they share a user): same-box A/B of 128 / 256 / 512 -- narrow rows
// (G <= 32: c3shard, k = 128) want 128 (triplet kernel 57.7 -> 53.3 us, step 0.0815 -> 0.0769 ms), a triplet per wave (G = 64:
// k = d = 256) wants 256 (c5list 192 against 200 us); 512 loses everywhere (c3shard 69 us)
template <int G>
struct TripletGradThreads { static constexpr int value = G == 64 ? 256 : 128; };

template <int G, bool VEC>
__global__ __launch_bounds__(TripletGradThreads<G>::value) void k_triplet_grad(SparseArgs a, const int32_t *__restrict__ user,
                                                      const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B) {
  const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  // `full`: every lane group of this workgroup has a triplet (all but the last workgroup): only then may the workgroup
  // meet at barriers (user-row combination below); in a partial workgroup the surplus groups leave here
  constexpr int TG_T = TripletGradThreads<G>::value;
  const bool full = ((int64_t)(blockIdx.x + 1) * TG_T) / G <= B;
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
  if (lane == 0) {
    a.lossb[b] = sp + reg * (nrm + bi * bi + bj * bj * 0.1f);          // BPRMF.py:108-112 / VBPR.py:121-126
    if (a.use_list) {
      // shared rows were listed for the apply pass by k_row_count (the occurrence that found the count at one); exclusive
      // rows (finished by this group alone) reset their multiplicity here -- nobody else looks at it
      if (exU) a.cntU[u] = 0;
      if (exI) { a.wBi[i] = bi - lr * (g + r2 * bi); a.cntI[i] = 0; }
      else atomicAdd(a.dBi + i, g + r2 * bi);
      if (exJ) { a.wBi[j] = bj - lr * (-g + (r2 * 0.1f) * bj); a.cntI[j] = 0; }
      else atomicAdd(a.dBi + j, -g + (r2 * 0.1f) * bj);
    } else {
      if (!exU) a.flagU[u] = 1u;
      if (exI) a.wBi[i] = bi - lr * (g + r2 * bi);
      else { atomicAdd(a.dBi + i, g + r2 * bi); a.flagI[i] = 1u; }
      if (exJ) a.wBi[j] = bj - lr * (-g + (r2 * 0.1f) * bj);
      else { atomicAdd(a.dBi + j, -g + (r2 * 0.1f) * bj); a.flagI[j] = 1u; }
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
  constexpr int NWV = TG_T / 64;
  __shared__ int s_u[NWV];
  __shared__ __attribute__((aligned(16))) float s_du[NWV][WGROW];
  bool wgc = false;
  const int wv = threadIdx.x >> 6;
  if (full && k + d <= WGROW && a.wg_combine) {
    const int u0 = __shfl(u, 0, 64);
    const bool wave_ok = G == 64 ? !exU : comb;
    if ((threadIdx.x & 63) == 0) s_u[wv] = wave_ok ? u0 : -1 - wv;
    __syncthreads();
    wgc = s_u[0] >= 0;
#pragma unroll
    for (int x = 1; x < NWV; ++x) wgc = wgc && s_u[x] == s_u[0];
  }
  // BPRMF with a workgroup-wide user (the common case in epoch order): the user-row gradient goes to LDS, so the backward
  // pass needs NO second read of the rows in the lane = element layout -- it is formed from the forward pass's registers
  // (16 B per lane) and summed across the groups (wide rows only, G >= 32) ...
  // ... and so are the finished rows of EXCLUSIVE items: one 16-B store per lane from the registers instead of four 4-B
  // re-reads and four 4-B stores (C3 shard: ~88 % of the item rows).
  const bool regs_ok = G >= 32 && VEC && k <= 4 * G && d <= 4 * G;
  const bool from_regs = regs_ok && wgc && d == 0;        // (d > 0: the W rows are written by the element loop below)
  // (stored AFTER the element loop, which still re-reads the pre-update item rows for the user-side gradient)
  const bool regI = regs_ok && a.reg_items && d == 0 && exI, regJ = regs_ok && a.reg_items && d == 0 && exJ;
  // ... and the gradients of SHARED item rows on that path (atomic staging): formed from the same registers and turned into
  // the lane = element layout of the atomics (full 128-B requests) through a per-group LDS row, instead of sending the whole
  // wave through the element loop below because one of its four item rows is shared (C3 shard: 12 % of the rows, 40 % of the
  // waves).  Only where the loop would otherwise be skipped (user side from registers).
  const bool shI = regs_ok && wgc && a.reg_items && d == 0 && !exI;
  const bool shJ = regs_ok && wgc && a.reg_items && d == 0 && !exJ;
  const bool doneI = regI || shI, doneJ = regJ || shJ;   // item rows with nothing to do in the element loop below
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
  if (regs_ok && __any(shI || shJ)) {             // wave-uniform entry; the LDS row belongs to this lane group alone
    __shared__ __attribute__((aligned(16))) float s_tr[TG_T / G][4 * G];
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
      if (exI) wi[c] = g * p; else atomicAdd(wi + c, g * p);
      if (exJ) wj[c] = -g * p; else atomicAdd(wj + c, -g * p);
    }
  }
  if (wgc) {                                              // workgroup-uniform
    __syncthreads();
    const int uw = s_u[0];
    for (int e = threadIdx.x; e < k + d; e += TG_T) {
      float sum;                                          // pairwise, in a fixed order
      if constexpr (NWV == 2) sum = s_du[0][e] + s_du[1][e];
      else if constexpr (NWV == 4) sum = (s_du[0][e] + s_du[1][e]) + (s_du[2][e] + s_du[3][e]);
      else sum = ((s_du[0][e] + s_du[1][e]) + (s_du[2][e] + s_du[3][e])) + ((s_du[4 % NWV][e] + s_du[5 % NWV][e]) + (s_du[6 % NWV][e] + s_du[7 % NWV][e]));
      static_assert(NWV == 2 || NWV == 4 || NWV == 8, "k_triplet_grad: 128, 256 or 512 threads");
      atomicAdd(e < k ? a.dGu + (size_t)uw * k + e : a.dTu + (size_t)uw * d + (e - k), sum);
    }
  }
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
  int row = 0;                                             // (B == 0: an empty step of a replicated-user rank only records lr_t)
  if (valid) row = kind == 0 ? clamp_quiet(user[b], T.U) : clamp_quiet(kind == 1 ? pos[b] : neg[b], T.I);
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
__global__ __launch_bounds__(1024) void k_dense_update(float *__restrict__ E, float *__restrict__ Bp, float *mE, float *vE,
                                                      float *mBp, float *vBp, const float *__restrict__ dEp,
                                                      const float *__restrict__ part, int SK, int D, int d, int PS, int adam,
                                                      float lr_t, float reg, float b1, float b2, float eps,
                                                      double *__restrict__ sqpart, float gscale, uint16_t *__restrict__ Et,
                                                      uint16_t *__restrict__ EtF, const int32_t *__restrict__ ilist,
                                                      const int32_t *__restrict__ ilist_n, int32_t *__restrict__ ilist_n_next,
                                                      int bound, float *__restrict__ W, int32_t *__restrict__ cnt_reset,
                                                      uint32_t *__restrict__ absmax_out) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[DU_KB][288];   // PS <= 272
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  if (ilist_n_next && blockIdx.x == 0 && threadIdx.x == 0) *ilist_n_next = 0;
  if (ilist) {
    int n = *ilist_n;
    n = n < bound ? n : bound;
    const int per = PS / 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * per; e += (int64_t)gridDim.x * blockDim.x) {
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
    for (int q = threadIdx.x; q < DU_KB * PQ; q += (int)blockDim.x) {
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
      for (int n = threadIdx.x; n < PS; n += (int)blockDim.x) {
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
  // (blockDim.x is a multiple of 64, at most 1024: a wave-level tree, then the waves' sums in a fixed order)
  __shared__ double red[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) t += red[q];
    sqpart[blockIdx.x] = t;
  }
  if (absmax_out) {
    __shared__ uint32_t wm[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(amax, o, 64); amax = v > amax ? v : amax; }
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t b = wm[0];
      for (int q = 1; q < (int)(blockDim.x >> 6); ++q) b = wm[q] > b ? wm[q] : b;
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
// Segment-mode index pass in ONE launch (round 3; it replaced the returning count atomics of k_row_count -- 130 K memory-side
// atomics, 13 us at C2 -- and the separate k_seg_alloc launch, 8-10 us).
// Owner workgroups: workgroup w OWNS the items [w*R, (w+1)*R).  It reads all 2B item occurrences of the batch (16 B per lane,
// 512 KB at B = 65 536, served by the XCD's L2 after the first workgroup has pulled it), and for the occurrences of ITS items
// counts and ranks in LDS -- every occurrence of an item meets in one workgroup, so no global atomic is needed and the count
// is final when the scan ends.  The workgroup then prefix-sums its counts, reserves its items' entries with ONE atomic on the
// entry cursor, writes seg_cnt / seg_ptr for every item of its range, lists the chunks of its touched items for k_item_seg
// ({item, first entry, entries of the chunk, entries of the item}: that kernel needs no second look-up) and zeroes the bf16 W
// rows of its UNTOUCHED items (7 % of the rows at C2, instead of the whole 8-MB image).
// User workgroups (blockIdx < nuser, sgd segment steps): one thread per triplet; the first lane of every run of equal users
// inside a wave adds the run's length to cntU[u] (the reference's visiting order has ~20 triplets per user: 1 atomic in 20) and
// the add that finds the count at zero names the user's slot for this batch (the batch position of that run head) and appends
// the user to the list of the batch's users, which the finishing lane groups of k_item_seg walk.
// ------------------------------------------------------------------------------------------------------------
constexpr int IX_T = 1024;        // threads of an index workgroup
constexpr int IX_RMAX = 8192;     // items an owner workgroup can own (LDS counters)
constexpr int IX_LPAD = 4;        // chunk-list slots of an owner beyond one per item (hot items' extra chunks; more: overflow list)

struct IndexSegArgs {
  const int32_t *user, *pos, *neg;
  int64_t B;
  int U, I, R, nown, nuser;       // R items per owner workgroup, nown owner workgroups behind nuser user workgroups
  int32_t *seg_rank, *seg_cnt, *seg_ptr;
  // Every owner has a region of its own for its items' entries (Ce entries, twice what it expects) and for its chunk list
  // (Lc = R + IX_LPAD slots): the usual step takes no global atomic.  An owner whose entries do not fit (hot items) reserves
  // ALL of them behind the regions (ent_over + cursor), and chunks that do not fit go to the overflow list behind the regions.
  int Ce, Lc, ent_over, lead_over, lead_cap;
  int32_t *cur, *cur_next;        // this step's cursors (overflow entries, overflow chunks, listed users) and the next step's (cleared here)
  int4 *lead;
  int32_t *cntU, *uslot_of, *ulist;   // user side; nullptr: not wanted (the user gradients stay in the staging tables)
  uint16_t *Wb;                   // bf16 W image (rows of untouched items are zeroed) or nullptr
  int PS;
  int aligned;                    // pos / neg are 16-byte aligned
  // byte planes of the batch's item ids (positives, then negatives; written by bprx_sample_*_h) or nullptr.  With them
  // R == 256 and B % 16 == 0: owner w scans own8 for bytes equal to w, sixteen values per 16-byte load.
  const uint8_t *own8, *loc8;
  int wide;                       // loc8 holds 16-bit locals (R = 2^shift > 256 items per owner: num_items > 65 536)
};

__global__ __launch_bounds__(IX_T) void k_index_seg(IndexSegArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if ((int)blockIdx.x < a.nuser) {
    // ---- user side (first in the grid: its returning atomics travel while the owners scan) ----
    const int64_t b = (int64_t)blockIdx.x * IX_T + tid;
    const bool valid = b < a.B;
    const int u = valid ? clamp_quiet(a.user[b], a.U) : 0;
    const unsigned long long vm = __ballot(valid);                  // (valid lanes are a prefix of the wave)
    const int up = __shfl_up(u, 1, 64);
    const bool uhead = valid && (lane == 0 || up != u);
    const unsigned long long hm = __ballot(uhead);
    const unsigned long long le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);          // lanes 0 .. lane
    const int hl = (hm & le) ? 63 - __clzll((long long)(hm & le)) : lane;                 // first lane of my run
    const unsigned long long stops = (hm | ~vm) & ~(hl == 63 ? ~0ull : ((2ull << hl) - 1ull));
    const int run_end = stops ? __ffsll((long long)stops) - 1 : 64;                       // first lane after my run
    // the add that finds the count at zero names the user's slot and lists the user for the finishing pass (k_item_seg); the
    // listed users of the WORKGROUP take their list positions with one cursor atomic (one per wave put 1 024 same-address
    // returning atomics in a row: +11 us on this kernel)
    __shared__ int u_wcnt[IX_T / 64], u_base;
    int old = -1;
    if (valid && lane == hl) old = atomicAdd(a.cntU + u, run_end - hl);
    const bool firstrun = old == 0;
    const unsigned long long fm = __ballot(firstrun);
    if (lane == 0) u_wcnt[wv] = __popcll(fm);
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int q = 0; q < IX_T / 64; ++q) { const int c = u_wcnt[q]; u_wcnt[q] = t; t += c; }
      u_base = t ? atomicAdd(a.cur + 2, t) : 0;
    }
    __syncthreads();
    if (firstrun) {
      a.uslot_of[u] = (int)b;
      a.ulist[u_base + u_wcnt[wv] + __popcll(fm & (lane ? (~0ull >> (64 - lane)) : 0ull))] = u;   // (at most B users: the list holds max_batch)
    }
    return;
  }
  // ---- item side ----
  extern __shared__ __attribute__((aligned(16))) int cnt[];        // [R] occurrence counters of my items
  __shared__ int wsc[IX_T / 64], wsl[IX_T / 64];
  __shared__ int s_gbase, s_lover, s_ltot;
  const int w = (int)blockIdx.x - a.nuser;
  const int lo = w * a.R;
  const int Rw = a.I - lo < a.R ? a.I - lo : a.R;                   // > 0: nown = ceil(I / R)
  for (int t = tid; t < Rw; t += IX_T) cnt[t] = 0;
  if (w == 0 && tid == 0) { a.cur_next[0] = 0; a.cur_next[1] = 0; a.cur_next[2] = 0; }
  __syncthreads();
  // My items as a SIGNED interval [Lb, Hb]: out-of-range indices are clamped like everywhere else (to item 0 / I-1; reported by
  // k_triplet_seg), so the first owner also takes everything below 0 and the last one everything above.  Four values are
  // rejected together: the smallest of their offsets from Lb against the interval's width -- one compare for 16 bytes; a
  // lane with a match (1 value in `nown`) ranks it on the spot.
  const int I = a.I;
  const int Lb = w == 0 ? (int)0x80000000 : lo, Hb = w == a.nown - 1 ? 0x7fffffff : lo + Rw - 1;
  const unsigned uL = (unsigned)Lb, Wm = (unsigned)Hb - (unsigned)Lb;
  auto one = [&](int v, int64_t occ) {
    if ((unsigned)v - uL <= Wm) a.seg_rank[occ] = atomicAdd(&cnt[clamp_quiet(v, I) - lo], 1);
  };
  auto four = [&](const int4 v, int64_t occ, bool in) {
    const unsigned x0 = (unsigned)v.x - uL, x1 = (unsigned)v.y - uL, x2 = (unsigned)v.z - uL, x3 = (unsigned)v.w - uL;
    const unsigned mn = min(min(x0, x1), min(x2, x3));
    if (in && mn <= Wm) { one(v.x, occ); one(v.y, occ + 1); one(v.z, occ + 2); one(v.w, occ + 3); }
  };
  if (a.own8 && a.wide) {
    // as below with 16-bit locals (two 16-byte loads of the local plane per load of the owner plane; four loads in flight)
    const uint4 *o4 = reinterpret_cast<const uint4 *>(a.own8), *l4 = reinterpret_cast<const uint4 *>(a.loc8);
    const int n16 = (int)((2 * a.B) >> 4);
    const int nblk = (n16 + IX_T - 1) / IX_T;
    const int rot = nblk ? (int)((((unsigned)w * 2654435761u) >> 8) % (unsigned)nblk) : 0;
    const unsigned wp = (unsigned)w * 0x01010101u;
    auto dword16 = [&](unsigned o, unsigned la, unsigned lb, int occ) {       // la: locals of bytes 0, 1; lb: of bytes 2, 3
      const unsigned y = o ^ wp;
      unsigned m = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
      while (m) {
        const int by = (__ffs((int)m) - 1) >> 3;
        const unsigned l = ((by < 2 ? la : lb) >> (16 * (by & 1))) & 0xffffu;
        a.seg_rank[occ + by] = atomicAdd(&cnt[l < (unsigned)Rw ? l : (unsigned)Rw - 1u], 1);
        m &= m - 1;
      }
    };
    for (int blk = 0; blk < nblk; blk += 4) {
      uint4 vo[4], va[4], vb[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int bb = blk + x + rot;
        bb = bb >= nblk ? bb - nblk : bb;
        const int e = bb * IX_T + tid;
        const int ec = e < n16 ? e : n16 - 1;
        vo[x] = o4[ec]; va[x] = l4[2 * ec]; vb[x] = l4[2 * ec + 1];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        int bb = blk + x + rot;
        bb = bb >= nblk ? bb - nblk : bb;
        const int e = bb * IX_T + tid;
        const unsigned y0 = vo[x].x ^ wp, y1 = vo[x].y ^ wp, y2 = vo[x].z ^ wp, y3 = vo[x].w ^ wp;
        const unsigned any = (((y0 - 0x01010101u) & ~y0) | ((y1 - 0x01010101u) & ~y1) | ((y2 - 0x01010101u) & ~y2) |
                              ((y3 - 0x01010101u) & ~y3)) & 0x80808080u;
        if (blk + x < nblk && e < n16 && any) {
          dword16(vo[x].x, va[x].x, va[x].y, 16 * e); dword16(vo[x].y, va[x].z, va[x].w, 16 * e + 4);
          dword16(vo[x].z, vb[x].x, vb[x].y, 16 * e + 8); dword16(vo[x].w, vb[x].z, vb[x].w, 16 * e + 12);
        }
      }
    }
  } else if (a.own8) {
    // one byte per occurrence: y = plane ^ (w in every byte) has a zero byte where the occurrence is mine; the classic
    // (y - 0x01..) & ~y & 0x80.. test is exact for "any zero byte in the dword", and the four dwords of a load are OR-ed before
    // the one compare.  A lane with a match (16 values in `nown`) finds the bytes with the exact per-byte mask and ranks them;
    // their low bytes (item - lo) come from the second plane, loaded beside the first (a dependent load would be waited for).
    const uint4 *o4 = reinterpret_cast<const uint4 *>(a.own8), *l4 = reinterpret_cast<const uint4 *>(a.loc8);
    const int n16 = (int)((2 * a.B) >> 4);
    const int nblk = (n16 + IX_T - 1) / IX_T;
    const int rot = nblk ? (int)((((unsigned)w * 2654435761u) >> 8) % (unsigned)nblk) : 0;
    const unsigned wp = (unsigned)w * 0x01010101u;
    auto dword = [&](unsigned o, unsigned l, int occ) {
      const unsigned y = o ^ wp;
      unsigned m = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);        // 0x80 in every zero byte of y, nowhere else
      while (m) {
        const int by = (__ffs((int)m) - 1) >> 3;
        a.seg_rank[occ + by] = atomicAdd(&cnt[(l >> (8 * by)) & 255u], 1);
        m &= m - 1;
      }
    };
    for (int blk = 0; blk < nblk; blk += 8) {
      uint4 vo[8], vl[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        int bb = blk + x + rot;
        bb = bb >= nblk ? bb - nblk : bb;
        const int e = bb * IX_T + tid;
        vo[x] = o4[e < n16 ? e : n16 - 1];
        vl[x] = l4[e < n16 ? e : n16 - 1];
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        int bb = blk + x + rot;
        bb = bb >= nblk ? bb - nblk : bb;
        const int e = bb * IX_T + tid;
        const unsigned y0 = vo[x].x ^ wp, y1 = vo[x].y ^ wp, y2 = vo[x].z ^ wp, y3 = vo[x].w ^ wp;
        const unsigned any = (((y0 - 0x01010101u) & ~y0) | ((y1 - 0x01010101u) & ~y1) | ((y2 - 0x01010101u) & ~y2) |
                              ((y3 - 0x01010101u) & ~y3)) & 0x80808080u;
        if (blk + x < nblk && e < n16 && any) {
          dword(vo[x].x, vl[x].x, 16 * e); dword(vo[x].y, vl[x].y, 16 * e + 4);
          dword(vo[x].z, vl[x].z, 16 * e + 8); dword(vo[x].w, vl[x].w, 16 * e + 12);
        }
      }
    }
  } else
#pragma unroll 1
  for (int role = 0; role < 2; ++role) {
    const int32_t *arr = role ? a.neg : a.pos;
    const int64_t occ0 = role ? a.B : 0;
    int64_t done = 0;
    if (a.aligned) {
      const int4 *a4 = reinterpret_cast<const int4 *>(arr);
      const int64_t n4 = a.B >> 2;
      // every owner reads the same bytes: each starts somewhere else (rot, in 16-KB blocks) and wraps, so that the CUs of an
      // XCD do not all ask its L2 for the same lines at the same moment
      const int64_t nblk = (n4 + IX_T - 1) / IX_T;                   // 16-KB blocks (the last one may be partial)
      const int64_t rot = nblk ? (int64_t)(((unsigned)w * 2654435761u) >> 8) % nblk : 0;
      for (int64_t blk = 0; blk < nblk; blk += 16) {                 // sixteen 16-B loads in flight per lane (256 KB per workgroup)
        int4 v[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          int64_t bb = blk + x + rot;
          bb = bb >= nblk ? bb - nblk : bb;
          const int64_t e = bb * IX_T + tid;
          v[x] = a4[e < n4 ? e : n4 - 1];                            // (unconditional: a predicated load is waited for on the spot)
        }
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          int64_t bb = blk + x + rot;
          bb = bb >= nblk ? bb - nblk : bb;
          const int64_t e = bb * IX_T + tid;
          four(v[x], occ0 + 4 * e, blk + x < nblk && e < n4);
        }
      }
      done = n4 * 4;
    }
    for (int64_t x = done + tid; x < a.B; x += IX_T) one(arr[x], occ0 + x);
  }
  __syncthreads();
  // counts are final: offsets of my items' segments and of their chunk-list entries
  const int per = (Rw + IX_T - 1) / IX_T;
  const int t0 = tid * per, t1 = t0 + per < Rw ? t0 + per : Rw;
  int cs = 0, ls = 0;
  for (int t = t0; t < t1; ++t) { const int c = cnt[t]; cs += c; ls += (c + SEG_CAP - 1) / SEG_CAP; }
  int ic = cs, il = ls;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int vc = __shfl_up(ic, o, 64), vl = __shfl_up(il, o, 64);
    if (lane >= o) { ic += vc; il += vl; }
  }
  if (lane == 63) { wsc[wv] = ic; wsl[wv] = il; }
  __syncthreads();
  if (tid == 0) {
    int tc = 0, tl = 0;
    for (int q = 0; q < IX_T / 64; ++q) { const int vc = wsc[q], vl = wsl[q]; wsc[q] = tc; wsl[q] = tl; tc += vc; tl += vl; }
    s_gbase = tc <= a.Ce ? w * a.Ce : a.ent_over + atomicAdd(a.cur, tc);
    s_lover = tl > a.Lc ? a.lead_over + atomicAdd(a.cur + 1, tl - a.Lc) : 0;
    s_ltot = tl;
  }
  __syncthreads();
  int e = s_gbase + wsc[wv] + ic - cs, l = wsl[wv] + il - ls;      // l: position in my chunk list
  int4 *const myl = a.lead + (size_t)w * a.Lc;
  for (int t = t0; t < t1; ++t) {
    const int c = cnt[t], item = lo + t;
    a.seg_cnt[item] = c;
    a.seg_ptr[item] = e;
    for (int q = 0; q * SEG_CAP < c; ++q, ++l) {
      const int4 ent = make_int4(item, e + q * SEG_CAP, c - q * SEG_CAP < SEG_CAP ? c - q * SEG_CAP : SEG_CAP, c);
      if (l < a.Lc) myl[l] = ent;
      else if (s_lover + (l - a.Lc) < a.lead_cap) a.lead[s_lover + (l - a.Lc)] = ent;
    }
    e += c;
    if (c == 0 && a.Wb) {
      uint4 *row = reinterpret_cast<uint4 *>(a.Wb + (size_t)item * a.PS);       // PS % 16 == 0: whole 16-B pieces
      for (int x = 0; x < a.PS / 8; ++x) row[x] = make_uint4(0, 0, 0, 0);
    }
  }
  for (int x = s_ltot + tid; x < a.Lc; x += IX_T) myl[x] = make_int4(0, 0, 0, 0);      // unused slots: no work
}

// ------------------------------------------------------------------------------------------------------------
// Segment-mode triplet kernel (round 3).  One lane group per triplet: forward scores, g, the two segment entries -- as
// before -- and the USER side without a staging round trip: the per-occurrence user-row gradients of the workgroup's
// TS_T / G consecutive triplets go to LDS, the runs of equal users inside the workgroup are summed there IN ORDER (one wave
// per run, lane = column), and
//   mode 0 (sgd): every run sum is added to the user's staging row (runs of one user cut by a workgroup boundary, or several
//          runs of a user, meet there), and the run that starts at the user's slot saves the pre-update row to uold[slot] for
//          k_item_seg's gathers.  Nothing is waited for.  The totals are applied by a few finishing lane groups at the front of
//          k_item_seg's grid (they walk the list of the batch's users built by k_index_seg): no apply launch, no claim marks.
//          (A first version finished users inside this kernel -- occurrence counters and a last-arriver read-back of the
//          totals: the waits for the atomics' acknowledgements and the counter round trips stretched every workgroup's tail,
//          32.5 us against 26.7 us in mode 1.)
//   mode 1 (adam_tf23, exported user gradients): the run sums are added to the staging rows and the user is marked.
// In the reference's visiting order (runs of ~20 triplets) that is 2 atomic row adds per workgroup of 32 triplets instead of
// one per wave or per triplet, and every sum inside a run is taken in batch order.
// ------------------------------------------------------------------------------------------------------------
constexpr int TS_T = 256;    // (same-box A/B of 128 / 256 / 512 threads: 256 wins on every shape -- C2 30.1 -> 25.7 us, c5 390 -> 348,
                             //  c4shard 51 -> 43.5, i.i.d. batches 54.9 -> 48.3; 512 was round 3's first choice, 128 loses to both)

struct SegUser {
  int mode;
  const int32_t *uslot_of;
  float *uold;
};

template <int G>
__global__ __launch_bounds__(TS_T) void k_triplet_seg(SparseArgs a, SegUser su, const int32_t *__restrict__ user,
                                                      const int32_t *__restrict__ pos, const int32_t *__restrict__ neg, int64_t B) {
  constexpr int T = TS_T / G;                               // triplets per workgroup (<= 64)
  __shared__ __attribute__((aligned(16))) float rows[4096];  // [T][k + d], k + d <= 8 G
  __shared__ int s_user[T], s_slot[T], s_cnt[T];
  const int tl = threadIdx.x / G, lane = threadIdx.x % G;
  const int64_t b0 = (int64_t)blockIdx.x * T + tl;
  const bool valid = b0 < B;                                // (the surplus groups of the last workgroup run along: barriers)
  const int64_t b = valid ? b0 : B - 1;
  const int u_raw = user[b], i_raw = pos[b], j_raw = neg[b];
  const int u = clamp_idx(u_raw, a.U, a.errflag, 1);
  const int i = clamp_idx(i_raw, a.I, a.errflag, 2), j = clamp_idx(j_raw, a.I, a.errflag, 3);
  const int k = a.k, d = a.d, kd = k + d;
  // everything that depends on the indices alone is requested together with the rows
  const float bi = a.Bi[i], bj = a.Bi[j];
  int spI = a.seg_rank[b] + a.seg_ptr[i], spJ = a.seg_rank[B + b] + a.seg_ptr[j];
  {
    const unsigned top = (unsigned)a.seg_cap - 1u;          // (never beyond the allocation, whatever the index state holds)
    spI = (int)((unsigned)spI < top ? (unsigned)spI : top); spJ = (int)((unsigned)spJ < top ? (unsigned)spJ : top);
  }
  int slot = 0, ucnt = 0;
  if (su.mode == 0) { slot = su.uslot_of[u]; ucnt = a.cntU[u]; }    // (the user's occurrences in the batch: final since k_index_seg)
  const int c4 = lane * 4;
  const bool hk = c4 < k, hd = c4 < d;
  const int ck = hk ? c4 : 0, cd = hd ? c4 : 0;
  const float mk = hk ? 1.f : 0.f, md = hd ? 1.f : 0.f;
  const float *gu = a.Gu + (size_t)u * k, *gi = a.Gi + (size_t)i * k, *gj = a.Gi + (size_t)j * k;
  const float4 p = ld4(gu + ck), q = ld4(gi + ck), r = ld4(gj + ck);
  float4 tp = make_float4(0.f, 0.f, 0.f, 0.f), tq = tp, tr = tp;
  float pid = 0.f, pjd = 0.f;
  if (d) {
    const float *tu = a.Tu + (size_t)u * d, *Pi = a.P + (size_t)i * a.PS, *Pj = a.P + (size_t)j * a.PS;
    tp = ld4(tu + cd); tq = ld4(Pi + cd); tr = ld4(Pj + cd);
    pid = Pi[d]; pjd = Pj[d];
  }
  // ---- forward: the un-differenced per-item scores of the reference (BPRMF.py:101-102) ----
  float si = mk * (p.x * q.x + p.y * q.y + p.z * q.z + p.w * q.w);
  float sj = mk * (p.x * r.x + p.y * r.y + p.z * r.z + p.w * r.w);
  float nrm = mk * (p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w + q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w +
                    r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
  float ti = 0.f, tj = 0.f;
  if (d) {
    ti = md * (tp.x * tq.x + tp.y * tq.y + tp.z * tq.z + tp.w * tq.w);
    tj = md * (tp.x * tr.x + tp.y * tr.y + tp.z * tr.z + tp.w * tr.w);
    nrm += md * (tp.x * tp.x + tp.y * tp.y + tp.z * tp.z + tp.w * tp.w);
  }
  si = group_sum<G>(si); sj = group_sum<G>(sj); nrm = group_sum<G>(nrm);
  float xp = bi + si, xn = bj + sj;
  if (d) {
    ti = group_sum<G>(ti); tj = group_sum<G>(tj);
    xp = xp + ti + pid;
    xn = xn + tj + pjd;
  }
  const float diff = xp - xn;
  const bool inr = (diff >= -80.0f) && (diff <= 1e8f);                 // tf.clip_by_value gradient mask
  const float cl = fminf(fmaxf(diff, -80.0f), 1e8f);
  const float z = -cl;                                                 // softplus(z), stable form
  const float sp = z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z));
  const float g = inr ? -1.0f / (1.0f + expf(diff)) : 0.f;            // -sigmoid(-diff)
  const float reg = a.reg, r2 = 2.f * reg, lr = a.lr;
  if (lane == 0 && valid) {
    a.lossb[b] = sp + reg * (nrm + bi * bi + bj * bj * 0.1f);          // BPRMF.py:108-112 / VBPR.py:121-126
    const int key = su.mode == 0 ? slot : u;                           // where k_item_seg finds the pre-update user row
    if ((unsigned)spI < (unsigned)a.seg_ent_cap) a.seg_ent[spI] = make_int2(key, __float_as_int(g));
    if ((unsigned)spJ < (unsigned)a.seg_ent_cap) a.seg_ent[spJ] = make_int2((int)((unsigned)key | 0x80000000u), __float_as_int(g));
  }
  // ---- user side: per-occurrence gradient rows -> LDS, runs summed in order ----
  float *row = rows + tl * kd;
  if (hk) *reinterpret_cast<float4 *>(row + c4) = make_float4(g * (q.x - r.x) + r2 * p.x, g * (q.y - r.y) + r2 * p.y,
                                                              g * (q.z - r.z) + r2 * p.z, g * (q.w - r.w) + r2 * p.w);
  if (hd) *reinterpret_cast<float4 *>(row + k + c4) = make_float4(g * (tq.x - tr.x) + r2 * tp.x, g * (tq.y - tr.y) + r2 * tp.y,
                                                                  g * (tq.z - tr.z) + r2 * tp.z, g * (tq.w - tr.w) + r2 * tp.w);
  if (lane == 0) { s_user[tl] = valid ? u : -1 - tl; s_slot[tl] = slot; s_cnt[tl] = ucnt; }
  __syncthreads();
  const int wl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int myu = wl < T ? s_user[wl] : 0, pru = (wl > 0 && wl < T) ? s_user[wl - 1] : 0;
  const unsigned long long hm = __ballot(wl < T && (wl == 0 || myu != pru));     // run heads among the workgroup's triplets
  const int nseg = __popcll(hm);
  for (int sg = wv; sg < nseg; sg += TS_T / 64) {                                // wave-uniform
    unsigned long long m = hm;
    for (int x = 0; x < sg; ++x) m &= m - 1;
    const int s0 = __ffsll((long long)m) - 1;
    m &= m - 1;
    const int s1 = m ? __ffsll((long long)m) - 1 : T;
    const int uu = s_user[s0];
    if (uu < 0) continue;
    float acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int c = wl + 64 * x;
      float sum = 0.f;
      if (c < kd)
        for (int t = s0; t < s1; ++t) sum += rows[t * kd + c];
      acc[x] = sum;
    }
    float *const tG = a.wGu + (size_t)uu * k, *const tT = d ? a.wTu + (size_t)uu * d : nullptr;
    float *const sG = a.dGu + (size_t)uu * k, *const sT = d ? a.dTu + (size_t)uu * d : nullptr;
    if (su.mode != 0) {
#pragma unroll
      for (int x = 0; x < 8; ++x) { const int c = wl + 64 * x; if (c < kd) atomicAdd(c < k ? sG + c : sT + (c - k), acc[x]); }
      if (wl == 0) a.flagU[uu] = 1u;
      continue;
    }
    // mode 0: the run's sum meets the user's other runs in the staging row (fire and forget: the totals are applied by the
    // finishing lane groups of k_item_seg, after this kernel); the segment that starts at the user's slot -- the first run
    // head of the user in the batch -- saves the PRE-update row for k_item_seg's gathers
    // A user whose ONLY run this is (its occurrence count equals the run's length: most users of a batch in the reference's
    // visiting order, every user of an i.i.d. batch) takes its update right here -- no staging round trip, nothing left for
    // the finishing groups (count back to zero = "done").  Every triplet of the run has read the row before the barrier above,
    // and no other workgroup holds a triplet of this user.  Same arithmetic as the finishing pass: row - lr * (0 + sum).
    const int sl = s_slot[s0];
    const bool head = (int64_t)blockIdx.x * T + s0 == (int64_t)sl;
    const bool excl = head && s_cnt[s0] == s1 - s0;
    if (!excl) {
#pragma unroll
      for (int x = 0; x < 8; ++x) { const int c = wl + 64 * x; if (c < kd) atomicAdd(c < k ? sG + c : sT + (c - k), acc[x]); }
    }
    if (head) {
      float *const uo = su.uold + (size_t)sl * kd;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int c = wl + 64 * x;
        if (c < kd) {
          float *const tp = c < k ? tG + c : tT + (c - k);
          const float old = *tp;
          uo[c] = old;
          if (excl) *tp = old - a.lr * acc[x];
        }
      }
      if (excl && wl == 0) a.cntU[uu] = 0;
    }
  }
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

constexpr int IS_T = 256;     // threads of an item-segment workgroup

template <int G, int ADAM>
__global__ __launch_bounds__(IS_T) void k_item_seg(SparseArgs a, float *__restrict__ Gi, float *__restrict__ Bi,
                                                  float *__restrict__ Wf, uint16_t *__restrict__ Wb, float lr, AdamFuse af) {
  const int lane = threadIdx.x % G;
  if ((int)blockIdx.x < a.nfin) {
    // ---- finishing lane groups (segment-mode sgd): the batch's users, whose gradient totals k_triplet_seg left in the staging
    // rows, take their update here: table row -= lr * total, staging row and occurrence count back to zero.  (The item groups
    // of this launch gather the PRE-update user rows from uold, not from the tables.)
    const int n = a.ulist_n[0];
    const int ngroups = a.nfin * (IS_T / G);
    const int c4 = lane * 4;
    for (int e = ((int)blockIdx.x * IS_T + (int)threadIdx.x) / G; e < n; e += ngroups) {
      const int u = clamp_quiet(a.ulist[e], a.U);
      if (a.cntU[u] == 0) continue;                      // a user of one run: k_triplet_seg has finished it (group-uniform)
      if (c4 < a.k) {
        float *t = a.wGu + (size_t)u * a.k + c4, *g = a.dGu + (size_t)u * a.k + c4;
        const float4 tv = ld4(t), gv = ld4(g);
        *reinterpret_cast<float4 *>(t) = make_float4(tv.x - lr * gv.x, tv.y - lr * gv.y, tv.z - lr * gv.z, tv.w - lr * gv.w);
        *reinterpret_cast<float4 *>(g) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (c4 < a.d) {
        float *t = a.wTu + (size_t)u * a.d + c4, *g = a.dTu + (size_t)u * a.d + c4;
        const float4 tv = ld4(t), gv = ld4(g);
        *reinterpret_cast<float4 *>(t) = make_float4(tv.x - lr * gv.x, tv.y - lr * gv.y, tv.z - lr * gv.z, tv.w - lr * gv.w);
        *reinterpret_cast<float4 *>(g) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (lane == 0) a.cntU[u] = 0;
    }
    return;
  }
  const int64_t e0 = ((int64_t)((int)blockIdx.x - a.nfin) * blockDim.x + threadIdx.x) / G;
  // one lane group per listed chunk (k_index_seg): an ordinary item is one chunk, a hot one is cut into chunks of SEG_CAP
  // (the owners' regions first -- unused slots hold no work -- then the overflow list, whose length is read on the device)
  if (e0 >= a.seg_lead_over) {
    int nl = a.seg_nlead[0];
    nl = nl < a.seg_lead_cap - a.seg_lead_over ? nl : a.seg_lead_cap - a.seg_lead_over;
    if (e0 - a.seg_lead_over >= nl) return;
  }
  const int4 job = a.seg_lead[e0];
  if (job.z <= 0) return;
  const int item = clamp_quiet(job.x, a.I);
  const int n = job.w;
  int e_first = job.y, ns = job.z;
  if (e_first < 0 || ns < 0 || e_first + ns > a.seg_cap) ns = 0;    // (never beyond the allocation)
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
  const float *const UG = a.uG, *const UT = d ? a.uT : a.uG;          // d == 0: any valid address, md == 0
  const int gs = a.usG, ds = d ? a.usT : a.usG;
  // the item's own row and bias depend on `item` alone: requested now, beside the first entries, not after the loop
  const size_t og = (size_t)item * k + c4, ow = (size_t)item * a.PS;
  float4 q = ld4(Gi + (size_t)item * k + ck);
  const float pb = Bi[item];
  for (; e + 2 <= ns; e += 2) {                                       // two entries in flight
    const int2 r0 = ent[e], r1 = ent[e + 1];
    const int u0 = r0.x & 0x7fffffff, u1 = r1.x & 0x7fffffff;
    const float4 p0 = ld4(UG + (size_t)u0 * gs + ck), p1 = ld4(UG + (size_t)u1 * gs + ck);
    const float4 t0 = ld4(UT + (size_t)u0 * ds + cd), t1 = ld4(UT + (size_t)u1 * ds + cd);
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
    const float4 p0 = ld4(UG + (size_t)u0 * gs + ck);
    const float4 t0 = ld4(UT + (size_t)u0 * ds + cd);
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
    __shared__ __attribute__((aligned(16))) float s_hot[IS_T / G][4 * G];
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
}

// segment-mode step whose users are finished inside k_triplet_seg (sgd, gradients not exported): no apply pass for them
inline bool seg_finishes_users(const bprx_handle *h) {
  return h->item_mode && h->cfg.optimizer == BPRX_OPT_SGD && !(h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD);
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
  if (h->item_mode) {
    // every item row is finished by k_item_seg, which gathers PRE-update user rows after k_triplet_grad: the user
    // side may therefore not be updated in place either (staging + k_apply_sgd)
    a.fastI = 0; a.fastU = 0; a.fast = 0;
  }
  a.seg_rank = h->seg_rank; a.seg_ptr = h->seg_ptr; a.seg_ent = (int2 *)h->seg_ent; a.seg_ent_cap = (int)h->seg_ent_cap; a.hot_done = h->hot_done;
  a.seg_lead = (const int4 *)h->seg_lead; a.seg_lead_cap = (int)h->seg_lead_cap;
  a.seg_nlead = h->seg_cursor ? h->seg_cursor + 3 * h->seg_cur_slot + 1 : nullptr;
  a.ulist = h->ulist; a.ulist_n = h->seg_cursor ? h->seg_cursor + 3 * h->seg_cur_slot + 2 : nullptr;
  a.nfin = 0;
  a.seg_lead_over = h->seg_lead_over;
  a.seg_cap = (int)h->seg_ent_cap;
  // where k_item_seg finds the pre-update user rows: a segment-mode sgd step finishes its users inside k_triplet_seg and
  // keeps their old rows in uold (entry key = user slot); otherwise the tables are untouched until the apply pass
  if (seg_finishes_users(h)) { a.uG = h->uold; a.uT = h->uold + a.k; a.usG = a.usT = a.k + a.d; }
  else { a.uG = h->t.Gu; a.uT = h->t.Tu; a.usG = a.k; a.usT = a.d; }
  // shared-row list: both sides on the exclusive-row fast path (sgd, atomic staging, no exported gradients)
  a.use_list = (h->slist && a.fastU && a.fastI) ? 1 : 0;
  a.reg_items = 1;
  a.wg_combine = 1;
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

// One adam_tf23 step of a whole row shard from its summed gradient (g is returned to zero): what the OWNER of routed rows does
// with the gradients the all-to-all brought back (bprx_route_scatter_add into g with scale 1).  The element function of the
// handle's own sweeps and replays, so a sharded table moves bit for bit like an unsharded one given the same gradient sums.
extern "C" int bprx_adam_rows(float *p, float *m, float *v, float *g, int64_t n, float lr_t, float beta1, float beta2, float eps,
                              void *stream) {
  if (!p || !m || !v || !g || n < 0) return BPRX_E_INVALID;
  if (n == 0) return BPRX_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_adam_sparse, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, m, v, g, (size_t)n, beta1, beta2,
                     lr_t, eps);
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
  if (!h || !msg || B < 0 || cap <= 0 || (B > 0 && !user)) return BPRX_E_INVALID;
  if (!(h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD)) BPRX_FAIL(h, BPRX_E_STATE, "pack_user_msg needs BPRX_FLAG_EXPORT_USER_GRAD");
  if (!h->pending_stage) BPRX_FAIL(h, BPRX_E_STATE, "pack_user_msg outside a step (after bprx_step_begin[_sparse])");
  if (B != h->pending_B) BPRX_FAIL(h, BPRX_E_INVALID, "pack_user_msg: B differs from the pending step's");
  hipStream_t s = (hipStream_t)stream;
  const int k = h->cfg.embed_k, d = h->cfg.embed_d;
  const dim3 grid((unsigned)((B + 255) / 256));
  if (B == 0) BPRX_HIP(h, hipMemsetAsync(msg, 0, 4 * sizeof(float), s));          // an empty message: count 0
  else if (vec_ok(h) && ((uintptr_t)msg & 15) == 0)
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
  if (h->item_mode) {
    // segment mode: ranks, counts, segment offsets, the chunk list of k_item_seg, the users' counts and slots -- one launch
    BprxProfScope pc(h, BPRX_PHASE_ROW_COUNT, s);
    IndexSegArgs x;
    x.user = u; x.pos = i; x.neg = j; x.B = B; x.U = a.U; x.I = a.I;
    int nown = h->num_cu > 0 ? h->num_cu : 256;                                  // one owner workgroup per CU ...
    if (nown > 1024) nown = 1024;
    if ((a.I + nown - 1) / nown > IX_RMAX) nown = (a.I + IX_RMAX - 1) / IX_RMAX;   // ... more when a range would not fit LDS
    if (nown > a.I) nown = a.I;
    x.R = (a.I + nown - 1) / nown;
    x.own8 = x.loc8 = nullptr; x.wide = 0;
    h->idx_kind = 1;
    if (h->idx8_use && h->idx8_shift) {                                          // the sampler left byte planes of this batch
      x.R = 1 << h->idx8_shift; x.own8 = h->own8; x.loc8 = h->loc8; x.wide = h->idx8_shift > 8; h->idx_kind = 2;
    }    // the sampler left byte planes of this batch
    x.nown = (a.I + x.R - 1) / x.R;
    x.seg_rank = h->seg_rank; x.seg_cnt = h->seg_cnt; x.seg_ptr = h->seg_ptr;
    x.Ce = (int)(2 * ((2 * B + x.nown - 1) / x.nown) + 64);
    x.Lc = x.R + IX_LPAD;
    x.ent_over = x.nown * x.Ce; x.lead_over = x.nown * x.Lc; x.lead_cap = (int)h->seg_lead_cap;
    if ((int64_t)x.ent_over + 2 * B > h->seg_ent_cap || (int64_t)x.lead_over + 2 * B / SEG_CAP + 64 > h->seg_lead_cap)
      BPRX_FAIL(h, BPRX_E_STATE, "index pass: segment buffers too small (I=%d, B=%lld)", a.I, (long long)B);
    h->seg_lead_over = x.lead_over;
    h->seg_cur_slot = h->seg_slot;                                               // the pair this step's kernels read
    x.cur = h->seg_cursor + 3 * h->seg_slot; x.cur_next = h->seg_cursor + 3 * (h->seg_slot ^ 1);
    h->seg_slot ^= 1;
    x.lead = (int4 *)h->seg_lead;
    const bool users = seg_finishes_users(h);
    x.cntU = users ? h->cntU : nullptr; x.uslot_of = users ? h->uslot_of : nullptr; x.ulist = users ? h->ulist : nullptr;
    const bool zw = a.d && h->cfg.feat_dtype != BPRX_F_FP32;                      // bf16 W image: rows of untouched items
    x.Wb = zw ? (uint16_t *)h->Wb : nullptr; x.PS = a.PS;
    x.aligned = (((uintptr_t)i | (uintptr_t)j) & 15) == 0;
    x.nuser = users ? (int)((B + IX_T - 1) / IX_T) : 0;
    hipLaunchKernelGGL(k_index_seg, dim3((unsigned)(x.nown + x.nuser)), dim3(IX_T), (size_t)x.R * sizeof(int), s, x);
    BPRX_LAUNCH_CHECK(h, "k_index_seg");
    return BPRX_OK;
  }
  if (h->fast_rows || h->list_mode) {
    BprxProfScope pc(h, BPRX_PHASE_ROW_COUNT, s);
    const int64_t cap = 2 * B < (int64_t)a.I ? 2 * B : (int64_t)a.I;
    hipLaunchKernelGGL(k_row_count, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, u, i, j, B, a.U, a.I, h->cntU, h->cntI,
                       a.fastU, (a.fastI || h->list_mode) ? 1 : 0,
                       h->list_mode ? h->ilist : (int32_t *)nullptr, h->list_cur, (int)cap,
                       a.use_list ? a.slist : (int32_t *)nullptr, a.slist_n, (int)(3 * h->cfg.max_batch));
  }
  BPRX_LAUNCH_CHECK(h, "k_row_count");
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
  const bool leaves_dirty = a.d && !h->list_mode && !bf;
  if (leaves_dirty || (a.d && h->W_dirty))
    BPRX_HIP(h, hipMemsetAsync(h->W, 0, (size_t)a.I * a.PS * sizeof(float), s));
  h->W_dirty = leaves_dirty;
  if (h->item_mode) {                                   // (segment mode implies the 16-B-per-lane layout: k % 4 == d % 4 == 0)
    const SegUser su = {seg_finishes_users(h) ? 0 : 1, h->uslot_of, h->uold};
    const int Gs = pick_group(a.k, a.d, true);
    const dim3 grid((unsigned)((B * Gs + TS_T - 1) / TS_T));
    switch (Gs) {
      case 8: hipLaunchKernelGGL((k_triplet_seg<8>), grid, dim3(TS_T), 0, s, a, su, u, i, j, B); break;
      case 16: hipLaunchKernelGGL((k_triplet_seg<16>), grid, dim3(TS_T), 0, s, a, su, u, i, j, B); break;
      case 32: hipLaunchKernelGGL((k_triplet_seg<32>), grid, dim3(TS_T), 0, s, a, su, u, i, j, B); break;
      default: hipLaunchKernelGGL((k_triplet_seg<64>), grid, dim3(TS_T), 0, s, a, su, u, i, j, B); break;
    }
  } else {
#define TG_LAUNCH(GG, VV)                                                                                                \
  hipLaunchKernelGGL((k_triplet_grad<GG, VV>), dim3((unsigned)((B * GG + TripletGradThreads<GG>::value - 1) / TripletGradThreads<GG>::value)), \
                     dim3(TripletGradThreads<GG>::value), 0, s, a, u, i, j, B)
    if (vec) { switch (G) { case 8: TG_LAUNCH(8, true); break; case 16: TG_LAUNCH(16, true); break; case 32: TG_LAUNCH(32, true); break; default: TG_LAUNCH(64, true); break; } }
    else { switch (G) { case 8: TG_LAUNCH(8, false); break; case 16: TG_LAUNCH(16, false); break; case 32: TG_LAUNCH(32, false); break; default: TG_LAUNCH(64, false); break; } }
#undef TG_LAUNCH
  }
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
  // (k_index_seg zeroed the bf16 rows of this batch's untouched items)
  // one lane group per chunk-list slot: the owners' regions (~one slot per item) + the overflow list (hot items' extra chunks)
  const int64_t bound = (int64_t)h->seg_lead_over + 2 * B / SEG_CAP + 64;
  // (the finishing workgroups stride over the batch's users -- a few thousand in the reference's visiting order, up to B for
  //  i.i.d. batches: sized for a quarter of B, the surplus ones leave after one load)
  int64_t nfin = B * G / IS_T / 4;
  nfin = nfin < 16 ? 16 : (nfin > 2048 ? 2048 : nfin);
  a.nfin = seg_finishes_users(h) ? (int)nfin : 0;
  const dim3 grid((unsigned)((bound * G + IS_T - 1) / IS_T + a.nfin));
#define LAUNCH_SEG(GG)                                                                                                   \
  do {                                                                                                                   \
    if (adam == 2) hipLaunchKernelGGL((k_item_seg<GG, 2>), grid, dim3(IS_T), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, lr_t, af); \
    else if (adam) hipLaunchKernelGGL((k_item_seg<GG, 1>), grid, dim3(IS_T), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, lr_t, af); \
    else hipLaunchKernelGGL((k_item_seg<GG, 0>), grid, dim3(IS_T), 0, s, a, h->t.Gi, h->t.Bi, Wf, Wb, lr_t, af);          \
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
  if (seg_finishes_users(h)) return BPRX_OK;             // k_item_seg finished the items, k_triplet_seg the users
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
    // (segment mode: k_item_seg finished the items and k_triplet_seg the users: nothing is left to apply)
    const int fk = ((h->cfg.flags & BPRX_FLAG_EXPORT_USER_GRAD) || h->item_mode) ? 1 : 0;
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
    const int ek = (h->item_mode || (h->cfg.flags & BPRX_FLAG_EXPORT_ITEM_GRAD)) ? 1 : 3;   // segments: k_item_seg took the items' steps;
                                                                             // exported item gradients: their owner does
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
  DISPATCH_G(G, vec, k_adam_catchup, grid_for(B ? 3 * B : 1, G), s, T, al, u, i, j, B, (int)h->adam_t, lr_t, h->lr_hist);
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
  const bool images = h->cfg.feat_dtype == BPRX_F_BF16;
  const bool lm = h->list_mode != 0;
  const int64_t bound = lm ? h->list_bound : 0;
  // one round per tile: a tile is DU_KB rows x PS / 4 float4 columns -- 160 threads' worth at PS = 80, 544 at PS = 272 (a 256-thread
  // block took three dependent rounds of slab loads there: c5small 21.5 us)
  int threads = (DU_KB * (h->PS / 4) + 63) / 64 * 64;
  threads = threads < 256 ? 256 : (threads > 1024 ? 1024 : threads);
  hipLaunchKernelGGL(k_dense_update, dim3(blocks), dim3((unsigned)threads), 0, s, h->t.E, h->t.Bp, h->t.m_E, h->t.v_E, h->t.m_Bp,
                     h->t.v_Bp, h->dEp, part, h->SK_step, D, h->cfg.embed_d, h->PS,
                     h->cfg.optimizer == BPRX_OPT_ADAM_TF23 ? 1 : 0, lr_t, h->cfg.reg, h->cfg.beta1, h->cfg.beta2,
                     h->cfg.epsilon, h->loss_acc, gscale, images ? (uint16_t *)h->Et : (uint16_t *)nullptr, (uint16_t *)h->EtF,
                     lm ? (const int32_t *)h->ilist : (const int32_t *)nullptr, (const int32_t *)h->list_cur,
                     lm ? h->ilist_n + (h->list_slot ^ 1) : (int32_t *)nullptr, (int)bound, h->W,
                     (lm && h->list_reset_cnt) ? h->cntI : (int32_t *)nullptr,
                     // fp8: the slot the next k_cast_Et8 reads (cleared by the last one)
                     h->cfg.feat_dtype == BPRX_F_FP8 ? (uint32_t *)h->qs + 2 + h->qs_slot : (uint32_t *)nullptr);
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
