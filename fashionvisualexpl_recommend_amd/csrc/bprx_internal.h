// bprx_internal.h -- private state of libbprx.so (C ABI: include/bprx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "bprx.h"

#define BPRX_DENSE_BLOCKS 2048

struct bprx_handle {
  bprx_config cfg;
  bprx_tables t;
  bool bound;
  int64_t adam_t;          // optimizer.iterations
  // lazy-exact adam_tf23 (bprx_sparse.hip): rows are brought up to date when they are read
  bool adam_lazy;
  int32_t *lastU, *lastI;  // [U], [I] step up to which the row (Gu/Tu resp. Gi/Bi and their slots) is current
  float *lr_hist;          // ring of the last ADAM_HIST steps' lr_t
  int64_t adam_synced;     // every row is current at least up to this step
  char err[512];

  // ---- scratch owned by the handle (device) ----
  float *dGu, *dGi, *dBi, *dTu;   // dense fp32 gradient staging, same shapes as the tables; all-zero between steps
  uint32_t *flagU, *flagI;        // "row touched this step" marks (sgd claim)
  float *lossb;                   // [max_batch] per-triplet loss (data + per-occurrence regularisation)
  double *loss_acc;               // [BPRX_DENSE_BLOCKS] per-block partial sums of ||E||^2+||Bp||^2 (k_dense_update)
  int dense_blocks;               // blocks of the last k_dense_update launch
  bool proj_fresh;                // bprx_step_project already ran for the coming step
  bool et_valid;                  // the bf16/fp8 image Et matches the bound E/Bp (cleared by every dense update, bind, tables_dirty)
  bool p_valid;                   // P holds the projections of ALL items for the bound E/Bp (bprx_score_block reuses it)
  bool fused_reduce;              // bprx_step: k_dense_update sums the split-K slabs itself (no k_reduce_parts)
  int32_t *errflag;               // device-side deferred error (index out of range)
  // VBPR projection state
  int PS;                         // padded row stride of P/W/Et: 16*ceil((d+1)/16)
  float *P;                       // [I][PS]  item projections f_i.[E|Bp]
  float *W;                       // [I][PS]  sum_b +-g_b*[theta_u|1] per item; all-zero between steps
  void *Wb;                       // bf16 [I][PS] copy of W for the backward MFMA
  float *Ppair;                   // [max_batch][PS] projections for bprx_score_pairs
  void *Ft;                       // tiled copy of F (bf16 / fp8 features): 8-KB blocks of 32 items x 256 B, see k_tile_F
  void *Et;                       // bf16 [PS][D]: [E|Bp|0]^T, refreshed every step
  void *EtF;                      // the same values in MFMA-fragment-major order (k_proj_fwd_rows), see k_cast_Et
  void *EtS;                      // fp8 features, PS/16 >= 10: the codes in the order of k_proj_fwd_f8s (scaled fp8 MFMA)
  float *dEp;                     // [D*d + D] dense gradient of E then Bp (no regularisation term)
  float *part;                    // [SK][D][PS] split-K slabs of the backward projection
  int SK;
  float *qs;                      // fp8 features: [1] = 1/(feat_scale*sE) for P, [2], [3] = max|E,Bp| bits (uint32, atomicMax;
                                  //               two slots used alternately, the idle one is cleared by k_cast_Et8)
  int qs_slot;
  bool absmax_valid;              // fp8: qs[2 + qs_slot] already holds max|E,Bp| of the bound values (left by k_dense_update)
  int fast_rows;                  // sgd: rows used by exactly one triplet of the batch are updated in place
  int32_t *cntU, *cntI;           // [U], [I] row multiplicities of the current batch (all-zero between steps)
  int seg_policy;                 // 0 never, 1 per step (2B >= I), 2 always (env BPRX_ITEM_MODE)
  int item_mode;                  // this step: 1: item-side gradients by per-item occurrence segments (k_item_seg), 0: global
                                  //    float atomics into the staging tables + claim-apply
  // Occurrence segments (segment mode), built by ONE launch of k_index_seg (bprx_sparse.hip): workgroup w OWNS the item
  // range [w*R, (w+1)*R): it scans all 2B item occurrences, counts and ranks those of its items in LDS (no global atomics),
  // prefix-sums its counts, reserves the entries with one cursor atomic and lists its items' chunks for k_item_seg.
  int32_t *seg_rank;              // [2 * max_batch] rank of occurrence (role*B + b) among its item's occurrences
  int32_t *seg_cnt;               // [I] occurrences of the item in this batch (rewritten for every item by each index pass)
  int32_t *seg_ptr;               // [I] start of the item's segment in seg_ent
  int32_t *seg_cursor;            // [6] two (overflow entries, overflow chunks, listed users) cursor triples used by alternate
                                  //     steps: an index pass clears the triple of the NEXT step
  int seg_slot;                   // cursor pair of the next segment-mode step
  int seg_cur_slot;               // cursor pair of the step in flight
  void *seg_lead;                 // int4 [seg_lead_cap] {item, first entry, entries of the chunk, entries of the item}: k_item_seg's work list
  int64_t seg_lead_cap;
  int seg_lead_over;              // this step: slots of the owners' regions (the overflow list follows)
  int64_t seg_ent_cap;            // entries allocated in seg_ent
  // user side of a segment-mode sgd step: k_triplet_seg sums the runs of equal users in LDS and adds the run sums to the staging
  // rows; finishing lane groups at the front of k_item_seg's grid apply the totals (no apply launch).  k_item_seg's item groups,
  // which need the PRE-update user rows, read them from uold (saved by the run that starts at the user's slot).
  int32_t *uslot_of;              // [U] batch position of the user's first run head = the user's slot (valid for users of the batch)
  int32_t *ulist;                 // [max_batch] the batch's users (first-run order): walked by k_item_seg's finishing groups
  float *uold;                    // [max_batch][k + d] pre-update [gamma_u | theta_u] of the slot's user
  // byte planes of the item ids of the NEXT step's batch, written by the library's own device samplers (bprx_sample_*_h) when
  // own8 = id >> idx8_shift (the owner workgroup of k_index_seg, 2^shift items each: at most 256 owners), loc8 = the rest; [2 * max_batch]
  // each (positives, then negatives).  idx8_pos / idx8_neg / idx8_B: the buffers and batch size they belong to; idx8_n: triplets
  // filled so far (-1: invalid); consumed (idx8_n = 0) by the step that uses them.
  uint8_t *own8, *loc8;
  int idx8_shift;                 // own8 = id >> idx8_shift (8: loc8 holds bytes; 9..13, num_items up to 2 M: loc8 holds uint16 id & (2^shift - 1))
  const int32_t *idx8_pos, *idx8_neg;
  int64_t idx8_B, idx8_n;
  bool idx8_use;                  // this step's index pass scans the byte planes
  int idx_kind;                   // bprx_index_pass_kind
  bool idx8_ready(const int32_t *pos, const int32_t *neg, int64_t B) const {
    return item_mode && own8 && B > 0 && idx8_n == B && idx8_B == B && idx8_pos == pos && idx8_neg == neg && B % 16 == 0;
  }
  int32_t *hot_done;              // [I] finished chunks of a hot item (k_item_seg), all-zero between steps
  void *seg_ent;                  // [seg_ent_cap] 8-byte entries {user or user slot | role << 31, g_b}: the owners' regions (twice
                                  //     the expected occupancy each) + 2 * max_batch for the owners that overflow theirs
  // touched-item list (sparse batches, 2B < I): both projections run over the batch's DISTINCT items only
  int list_policy;                // 0 never, 1 per step (2B < I), 2 always (env BPRX_LIST_MODE)
  int list_mode;                  // this step
  int32_t *ilist;                 // [min(2*max_batch, I)] distinct items of the batch, in arrival order (k_row_count)
  int32_t *ilist_n;               // [2] their number, two cursors used by alternate list-mode steps: k_dense_update (the
                                  //     last kernel of a step) clears the cursor the NEXT list-mode step appends through
  int list_slot;                  // cursor of the next list-mode step
  int32_t *list_cur;              // ilist_n + slot of the step in flight
  int64_t list_bound;             // host-side bound of the list length of the step in flight: min(2B, I)
  bool list_reset_cnt;            // list mode: k_cast_W_rows resets cntI (no exclusive-row fast path on the item side)
  bool W_dirty;                   // the fp32 W table is not all-zero (left so by a dense fp32-feature step)
  int32_t *slist, *slist_n;       // sgd fast path: list of the batch's SHARED rows (kind << 30 | row), two alternating cursors
  int slist_slot;
  int SK_step;                    // split-K slabs written by this step's backward projection (<= SK)
  int num_cu;                     // compute units of the device (balanced forward grid)
  int fwd_variant;                // 0: the plain forward kernel (env BPRX_FWD_VARIANT, read at create), else the per-shape policy
  int64_t pending_B;              // B of the step between _begin and _end (0 = none)
  int pending_stage;              // 1 = bprx_step_begin_sparse done (user gradients final), 2 = whole _begin done
  const int32_t *pend_u, *pend_i, *pend_j;   // the pending step's index buffers (bprx_step_begin_dense)
  float pend_lr;
  // replicated-user message exchange (bprx_pack_user_msg / bprx_apply_user_msgs)
  int32_t *msg_cursor;            // [2] next free slot of the message being packed, workgroups done (both zero between calls)
  int32_t *msg_next;              // [nranks*cap] chain links of the occurrences of one user across the ranks' messages
  size_t msg_next_n;
  // side stream: the sparse optimizer pass (k_apply_sgd / adam sweeps: factor tables only) runs beside the backward
  // projection (F, W, slabs only); forked after k_triplet_grad, joined in bprx_step_end
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
  bool side_pending;
  int side_mode;                  // BPRX_SIDE_STREAM bit mask: 1 = sparse optimizer pass beside proj_bwd, 4 = lazy-Adam catch-up beside proj_fwd
  // hipGraph of the whole sgd step (bprx_step): captured on first use, replayed while the call's arguments repeat
  // hipGraphs of the whole sgd step (bprx_step): captured when a call repeats the previous call's arguments, replayed while
  // they keep repeating.  A captured launch sequence depends on the host-side state below (cursor slots that alternate from
  // step to step, validity of the derived images), so an exec is stored with the state it was captured in and the state
  // it leaves, and is replayed only from the same state; a steady training loop alternates between two execs.
  int graph_mode;                 // env BPRX_GRAPH: 0 (default) = never, 1 = always, 2 = small steps only (B <= 8192)
  struct GraphSig { int list_slot, slist_slot, qs_slot, seg_slot; bool et_valid, p_valid, absmax_valid, W_dirty, idx8; };
  struct GraphEnt { hipGraphExec_t exec; GraphSig in, out; };
  GraphEnt graph_ents[4];
  int graph_n;
  struct { const void *u, *i, *j, *loss; int64_t B; void *stream; float lr, reg; } graph_key;
  // per-kernel HIP-event timing (bprx_profile_*)
  bool prof;
  struct ProfRec { int phase; hipEvent_t a, b; };
  std::vector<ProfRec> *prof_pending;
  std::vector<hipEvent_t> *prof_free;
};

// RAII: records an event pair around one kernel launch when profiling is on.
struct BprxProfScope {
  bprx_handle *h; hipStream_t s; hipEvent_t a, b; int phase; bool on;
  BprxProfScope(bprx_handle *h_, int phase_, hipStream_t s_) : h(h_), s(s_), phase(phase_), on(h_->prof) {
    if (!on) return;
    auto get = [&]() { hipEvent_t e; if (!h->prof_free->empty()) { e = h->prof_free->back(); h->prof_free->pop_back(); }
                       else (void)hipEventCreate(&e); return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, s);
  }
  ~BprxProfScope() {
    if (!on) return;
    (void)hipEventRecord(b, s);
    h->prof_pending->push_back({phase, a, b});
  }
};

#define BPRX_FAIL(h, code, ...)                                   \
  do {                                                            \
    snprintf((h)->err, sizeof((h)->err), __VA_ARGS__);            \
    return (code);                                                \
  } while (0)

#define BPRX_HIP(h, call)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess) BPRX_FAIL(h, BPRX_E_HIP, "%s: %s", #call, hipGetErrorString(e__));      \
  } while (0)

#define BPRX_LAUNCH_CHECK(h, name)                                                                 \
  do {                                                                                             \
    hipError_t e__ = hipGetLastError();                                                            \
    if (e__ != hipSuccess) BPRX_FAIL(h, BPRX_E_HIP, "launch %s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- launchers implemented in the kernel translation units ----
// sparse part (bprx_sparse.hip)
int bprx_launch_score(bprx_handle *h, const int32_t *u, const int32_t *i, int64_t B, const float *Prow,
                      int p_by_pair, float *x, hipStream_t s);
int bprx_launch_index_pass(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, hipStream_t s);
int bprx_launch_triplet_grad(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B,
                             hipStream_t s);
int bprx_launch_item_seg(bprx_handle *h, const int32_t *i, const int32_t *j, int64_t B, float lr_t, hipStream_t s);
int bprx_launch_apply(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B,
                      float lr_t, hipStream_t s);
int bprx_launch_dense_update(bprx_handle *h, float lr_t, hipStream_t s);
int bprx_launch_adam_catchup(bprx_handle *h, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, float lr_t,
                             hipStream_t s);
int bprx_launch_adam_sync(bprx_handle *h, int64_t t, hipStream_t s);
int bprx_launch_adam_reset(bprx_handle *h, int64_t t, hipStream_t s);
int bprx_adam_hist(void);
int bprx_launch_loss_reduce(bprx_handle *h, int64_t B, float *loss_out, hipStream_t s);
int bprx_launch_score_block(bprx_handle *h, int32_t u0, int32_t u1, float *out, hipStream_t s);
int bprx_launch_score_gemm(bprx_handle *h, int32_t u0, int32_t u1, float *out, hipStream_t s);
// projection part (bprx_proj.hip)
int bprx_launch_tile_F(bprx_handle *h);
int bprx_launch_cast_Et(bprx_handle *h, hipStream_t s);
// rows == nullptr: items 0..nrows; else the listed items.  nrows_dev (device, optional): the actual row count (<= nrows, the
// host-side bound the grid is sized for).  scatter: row t of the result goes to Pout[rows[t]] instead of Pout[t].
int bprx_launch_proj_fwd(bprx_handle *h, const int32_t *rows, int64_t nrows, const int32_t *nrows_dev, int scatter, float *Pout,
                         hipStream_t s);
int bprx_launch_proj_bwd(bprx_handle *h, int64_t B, hipStream_t s);
