/*
 * bprx.h -- C ABI of libbprx.so: the MI355X (gfx950) BPRMF / VBPR training hot path.
 *
 * The reference (peternara/FashionVisualExpl-recommend) has NO native plugin/FFI interface for this
 * path: it is TensorFlow-2.3 eager Python.  Its boundary is the Python class surface
 *     src/recommender/models/BPRMF.py:55  call            :78  predict_all   :87  train_step
 *     src/recommender/models/VBPR.py:59   call            :88  predict_all   :99  train_step
 *     src/dataset/dataset.py:83           all_triple_batches (index stream)
 *     src/recommender/Evaluator.py:82     _eval_by_user     (HR/nDCG/AUC/P/R definition)
 * Each entry point below names the reference lines it replaces.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.  Plain pointers and sizes only: no torch / TF types.
 *
 * Conventions
 *   - Every function returns 0 on success, a negative BPRX_E_* code otherwise; bprx_last_error() gives text.
 *     No exception crosses the ABI.  Index range errors are detected on the device and reported by the next
 *     bprx_sync_check() (indices are clamped so that no kernel ever faults).
 *   - All table / index / output pointers are DEVICE pointers owned by the caller (e.g. torch tensors); the
 *     handle owns only its scratch.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     all work is enqueued, nothing synchronises unless stated.
 *   - A handle is bound to one device and is not re-entrant.  One process per GPU.
 *   - Tables are row-major fp32 exactly like the reference's tf.Variables (BPRMF.py:48-50, VBPR.py:44-54).
 *     F may be fp32 [I,D] or bf16 [I,D] (raw uint16 bit patterns).
 */
#ifndef BPRX_H_
#define BPRX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPRX_ABI_VERSION 6

#if defined(__GNUC__)
#define BPRX_API __attribute__((visibility("default")))
#else
#define BPRX_API
#endif

enum { BPRX_MODEL_BPRMF = 0, BPRX_MODEL_VBPR = 1 };
enum { BPRX_OPT_SGD = 0, BPRX_OPT_ADAM_TF23 = 1 };
enum { BPRX_F_FP32 = 0, BPRX_F_BF16 = 1, BPRX_F_FP8 = 2 };   /* FP8: OCP e4m3fn codes, see feat_scale */

enum {
  BPRX_OK = 0,
  BPRX_E_INVALID = -1,   /* bad argument / unsupported configuration */
  BPRX_E_STATE = -2,     /* call order (tables not bound, ...) */
  BPRX_E_HIP = -3,       /* HIP runtime error */
  BPRX_E_RANGE = -4,     /* an index was out of range (seen by bprx_sync_check) */
  BPRX_E_NOMEM = -5
};

typedef struct bprx_handle bprx_handle;
typedef struct bprx_sampler bprx_sampler;

typedef struct {
  int32_t abi_version;   /* BPRX_ABI_VERSION */
  int32_t model;         /* BPRX_MODEL_*            train_rec.py:75-78 (--rec bprmf|vbpr) */
  int32_t num_users;     /* rows of Gu/Tu held by THIS handle (a shard when user-sharded) */
  int32_t num_items;     /* rows of Gi/Bi/F held by THIS handle (a shard when item-sharded) */
  int32_t embed_k;       /* --embed_k  train_rec.py:42 */
  int32_t embed_d;       /* --embed_d  train_rec.py:43 (VBPR only, else 0) */
  int32_t feat_dim;      /* D = F.shape[1]  visual_loader_mixin.py:31 (VBPR only, else 0) */
  int32_t feat_dtype;    /* BPRX_F_* */
  int32_t optimizer;     /* BPRX_OPT_* ; reference = Adam (BPRMF.py:52, VBPR.py:56) */
  int32_t device;        /* HIP device ordinal */
  int64_t max_batch;     /* largest B any bprx_step/bprx_score_pairs call will use */
  float lr;              /* --lr   train_rec.py:28 */
  float reg;             /* --reg  train_rec.py:44,69 */
  float beta1, beta2, epsilon; /* adam_tf23: 0.9, 0.999, 1e-7 (tf.optimizers.Adam defaults) */
  int32_t flags;         /* BPRX_FLAG_* */
  float feat_scale;      /* BPRX_F_FP8: F holds e4m3fn(f * feat_scale), i.e. f ~ F / feat_scale (448 for max-abs-normalised
                            features, visual_loader_mixin.py:30); ignored otherwise */
} bprx_config;

/* BPRX_FLAG_EXPORT_USER_GRAD (item-sharded multi-GPU; with adam_tf23 see bprx_adam_rows): the bound Gu/Tu are per-step STAGING rows fetched from
   their owner ranks (row b = the user row of triplet b); the step leaves their summed gradients in the buffers of
   bprx_user_grad() instead of applying them, and the caller routes those rows back to the owners
   (bprx_scatter_add with scale = -lr) and clears them with bprx_clear_user_grad(). */
enum { BPRX_FLAG_EXPORT_USER_GRAD = 1, BPRX_FLAG_EXPORT_ITEM_GRAD = 2, BPRX_FLAG_DENSE_ALLREDUCE = 4,
       BPRX_FLAG_ADAM_SWEEP = 8, BPRX_FLAG_ADAM_LAZY = 16 };
/* BPRX_FLAG_ADAM_SWEEP / BPRX_FLAG_ADAM_LAZY: the caller's choice of adam_tf23's form (whole-table sweeps / lazily-exact replay:
   the same arithmetic) instead of bprx_create's estimate from num_users and max_batch -- a caller that knows the real batch size
   and the number of training interactions knows the replay depth (interactions / batch) exactly.  See bprx_adam_is_lazy. */
/* BPRX_FLAG_DENSE_ALLREDUCE (with BPRX_FLAG_EXPORT_USER_GRAD, replicated-user step): the per-rank message carries NO dense
   part; the caller all-reduces (sum, RCCL) the buffer of bprx_dense_grad() between bprx_pack_user_msg and bprx_step_end --
   the "RCCL all-reduce on E / beta'" form of SURVEY 8(e).  Without it dE|dBp rides in the all-gathered message and is summed
   in rank order (bit-identical replicas whatever the collective's reduction order). */
/* BPRX_FLAG_EXPORT_ITEM_GRAD (user-sharded multi-GPU BPRMF; with adam_tf23 see bprx_adam_rows): the mirror image -- the bound Gi/Bi are per-step
   STAGING rows fetched from the item owners (row b = the positive item row of triplet b, row B+b its negative item row);
   their gradients are left in the buffers of bprx_item_grad() and cleared with bprx_clear_item_grad(). */

/* Device pointers to the model state.  Unused entries (BPRMF: Tu,F,E,Bp; sgd: every m_/v_) are NULL. */
typedef struct {
  float *Gu;       /* [U,k]  BPRMF.py:49 */
  float *Gi;       /* [I,k]  BPRMF.py:50 */
  float *Bi;       /* [I]    BPRMF.py:48 */
  float *Tu;       /* [U,d]  VBPR.py:46  */
  const void *F;   /* [I,D]  VBPR.py:49, frozen; fp32, bf16 or fp8 (e4m3fn) per feat_dtype */
  float *E;        /* [D,d]  VBPR.py:52  */
  float *Bp;       /* [D]    VBPR.py:44 ([D,1]) */
  float *m_Gu, *v_Gu, *m_Gi, *v_Gi, *m_Bi, *v_Bi, *m_Tu, *v_Tu, *m_E, *v_E, *m_Bp, *v_Bp; /* Adam slots */
} bprx_tables;

BPRX_API int bprx_abi_version(void);

/* Model(data, params) constructor scratch (BPRMF.py:23-53, VBPR.py:19-57). */
BPRX_API int bprx_create(const bprx_config *cfg, bprx_handle **out);
BPRX_API int bprx_destroy(bprx_handle *h);
BPRX_API const char *bprx_last_error(const bprx_handle *h); /* h == NULL: error of the last failed bprx_create */

/* Binds caller-owned device tables (they stay owned by the caller and are updated in place by the steps).
   bf16 / fp8 features: F is frozen (visual_loader_mixin.py:22-31) -- this call copies it ONCE into a tiled layout the
   projections read (synchronous; work is enqueued on the null stream, so F must be complete with respect to it); the
   caller's F is not read again until the next bprx_bind_tables and may be released.
   A handle is not thread-safe; all other entry points only enqueue work on the stream they are given. */
BPRX_API int bprx_bind_tables(bprx_handle *h, const bprx_tables *t);
/* The bound tables stay owned by the caller, who may write them between calls (restoring a snapshot: the reference
   deep-copies / checkpoints the whole model, BPRMF.py:156-160,177-179).  The handle keeps images DERIVED from them across
   calls -- the bf16/fp8 image of [E|Bp]^T and the item projections P = F.[E|Bp] that bprx_score_block / bprx_score_pairs
   reuse until a step changes E/Bp -- so after writing any bound table from outside the library call bprx_tables_dirty()
   before the next library call.  (bprx_bind_tables implies it.)  `stream`: the stream the outside writes were enqueued on --
   the lazy-Adam bookkeeping reset is ordered behind them there (NULL: the null stream, synchronised). */
BPRX_API int bprx_tables_dirty(bprx_handle *h, void *stream);
BPRX_API int bprx_set_hyper(bprx_handle *h, float lr, float reg);           /* train_rec.py:69 (args.reg = reg) */
BPRX_API int bprx_set_adam_step(bprx_handle *h, int64_t iterations, void *stream);   /* optimizer.iterations (resume); stream as above */
/* adam_tf23 is implemented LAZILY but exactly: TF-2.3's Adam moves every row of every table every step (non-lazy sparse
   apply, BPRMF.py:123 / VBPR.py:142); here a row that received no gradient is brought up to date -- by replaying the
   skipped steps with the arithmetic of the whole-table sweep, bit for bit -- when it is next read.  The library does that
   itself wherever IT reads the tables (steps, bprx_score_pairs, bprx_score_block); a caller that reads the bound tensors
   directly (snapshot, inspection) calls bprx_sync_adam first: afterwards every row holds what TF's tables would hold after
   optimizer.iterations steps.  No-op for sgd, or when nothing is pending.  (BPRX_ADAM_LAZY=0: the sweeps, for A/B.) */
BPRX_API int bprx_sync_adam(bprx_handle *h, void *stream);
BPRX_API int64_t bprx_get_adam_step(const bprx_handle *h);
/* 1: adam_tf23 runs lazily-exact on this handle (per-row replay, bprx_sync_adam meaningful), 0: by whole-table sweeps.  Chosen at
   bprx_create from the table sizes and max_batch (BPRX_ADAM_LAZY=0 / 1 forces it); the arithmetic is the same. */
BPRX_API int bprx_adam_is_lazy(const bprx_handle *h);

/* Model.call((user,item)) -> xui        BPRMF.py:55-76 / VBPR.py:59-86.   x: fp32 [B] */
BPRX_API int bprx_score_pairs(bprx_handle *h, const int32_t *user, const int32_t *item, int64_t B, float *x, void *stream);

/* Model.train_step((user,pos,neg)) -> loss   BPRMF.py:87-125 / VBPR.py:99-144.
   Batch-synchronous: all gradients from pre-update values, duplicate rows summed, one optimizer update.
   loss_out: device fp32 scalar (data term + regularisation, as the reference's loss.numpy()); may be NULL. */
BPRX_API int bprx_step(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B,
              float *loss_out, void *stream);

/* The same step in two halves, for item-sharded multi-GPU VBPR: after _begin the dense gradient of the
   shared parameters ([D,d] dE followed by [D] dBp, fp32, WITHOUT the 2*reg*E term) sits in the buffer
   returned by bprx_dense_grad(); the caller all-reduces it (RCCL) and calls _end, which adds the
   regularisation term and applies the optimizer to E/Bp.  bprx_step == _begin + _end. */
BPRX_API int bprx_step_begin(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B,
                    void *stream);
BPRX_API int bprx_dense_grad(bprx_handle *h, float **ptr, int64_t *count);
BPRX_API int bprx_step_end(bprx_handle *h, float *loss_out, void *stream);
/* _begin itself in two halves (bprx_step_begin == _begin_sparse + _begin_dense), so that a collective on the user-side
   gradients overlaps the backward projection:
     _begin_sparse  index pass, item projections P = F.[E|Bp], per-triplet gradients: the USER-side gradients of the batch
                    are final afterwards (bprx_user_grad / bprx_pack_user_msg may follow at once)
     _begin_dense   item rows, W, dE|dBp = F^T W (bprx_dense_grad() is final afterwards); reads the index buffers of
                    _begin_sparse again: they must stay valid and unchanged until it returns */
BPRX_API int bprx_step_begin_sparse(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B,
                           void *stream);
BPRX_API int bprx_step_begin_dense(bprx_handle *h, void *stream);
/* Item-sharded multi-GPU helpers.
   bprx_step_project: the item-projection prologue of the step (P = F.[E|Bp]) on its own, so that it can overlap the
   all-to-all that fetches the user rows; a following bprx_step_begin does not repeat it.
   bprx_user_grad / bprx_clear_user_grad: see BPRX_FLAG_EXPORT_USER_GRAD.
   bprx_scatter_add: table[idx[r], :] += scale * rows[r, :] for r < n (fp32 atomics; duplicates in idx are summed):
   the owner-side application of routed gradient rows.  Stateless; all pointers are device pointers. */
BPRX_API int bprx_step_project(bprx_handle *h, void *stream);
BPRX_API int bprx_user_grad(bprx_handle *h, float **dGu, float **dTu);
BPRX_API int bprx_clear_user_grad(bprx_handle *h, int64_t n_rows, int32_t marks_only, void *stream);   /* marks_only: the gradient
   rows were already returned to zero (bprx_route_pack); only the touched-row marks are cleared */
/* Replicated-user multi-GPU step (item-sharded VBPR with every rank holding ALL user rows; needs
   BPRX_FLAG_EXPORT_USER_GRAD and a handle created with num_users = the GLOBAL user count): ONE fixed-size all-gather per
   step, no data-dependent routing, no host synchronisation.
     bprx_user_msg_floats   size (in 4-byte words, a multiple of 4) of one rank's message for `cap` distinct users per batch
     bprx_pack_user_msg     after bprx_step_begin (without the dense part, BPRX_FLAG_DENSE_ALLREDUCE: already after
                            bprx_step_begin_sparse): moves the summed gradient rows of the batch's distinct users out of the
                            staging tables (which are left all-zero) into msg = [count,0,0,0 | ids[cap rounded up to 4] |
                            dGu[cap,k] | dTu[cap,d] | dE|dBp]; more than `cap` distinct users are reported by
                            bprx_sync_check (BPRX_E_RANGE).  msg should be 16-byte aligned (vector copies).
     bprx_apply_user_msgs   after the all-gather (msgs = nranks messages back to back): Gu/Tu[id] += scale * row for
                            every rank's rows, per user in ascending rank order (the occurrences of a user across the
                            messages are chained and one lane group applies them one after the other: every replica
                            performs the same additions in the same order and the replicas stay bit-identical; two
                            launches whatever nranks is), and bprx_dense_grad() = sum over ranks of their dE|dBp parts in
                            rank order; then bprx_step_end.
     bprx_sum_dense_parts   BPRX_FLAG_DENSE_ALLREDUCE handles that want the ORDERED sum instead of an RCCL all-reduce:
                            parts = an all-gather of bprx_dense_grad() (nranks x count floats); bprx_dense_grad() becomes
                            their sum in rank order. */
BPRX_API int64_t bprx_user_msg_floats(const bprx_handle *h, int64_t cap);
BPRX_API int bprx_pack_user_msg(bprx_handle *h, const int32_t *user, int64_t B, int64_t cap, float *msg, void *stream);
BPRX_API int bprx_apply_user_msgs(bprx_handle *h, const float *msgs, int32_t nranks, int64_t cap, float scale, void *stream);
BPRX_API int bprx_sum_dense_parts(bprx_handle *h, const float *parts, int32_t nranks, void *stream);
/* adam_tf23 in the all-to-all modes: the handle (lazy form) takes the Adam steps of the rows it keeps and of E|Bp; the rows whose
   gradients it exports are stepped by their OWNER rank, over its whole shard (TF-2.3's Adam is not lazy: every row decays and
   moves every step, BPRMF.py:123 / VBPR.py:142): the owner adds the returned gradient rows into a zero gradient table
   (bprx_route_scatter_add, scale 1) and calls
     bprx_adam_rows  p, m, v, g: n floats each (the shard, its two moment tables, the gradient table: returned to zero);
                     lr_t from bprx_step_lr of this step -- every rank steps every global step (an empty batch is a step with
                     B = 0), so all ranks hold the same step count;
     bprx_step_lr    the bias-corrected learning rate of the step begun last (sgd: lr). */
BPRX_API int bprx_adam_rows(float *p, float *m, float *v, float *g, int64_t n, float lr_t, float beta1, float beta2, float eps,
                            void *stream);
BPRX_API int bprx_step_lr(const bprx_handle *h, float *lr_t);
BPRX_API int bprx_item_grad(bprx_handle *h, float **dGi, float **dBi);
BPRX_API int bprx_clear_item_grad(bprx_handle *h, int64_t n_rows, int32_t marks_only, void *stream);
BPRX_API int bprx_scatter_add(float *table, int32_t num_rows, int32_t num_cols, const int32_t *idx, const float *rows,
                              int64_t n, float scale, void *stream);

/* Fixed-capacity row routing for the all-to-all multi-GPU modes (user-sharded BPRMF moves item rows, the
   partitioned-user form of item-sharded VBPR moves user rows).  Stateless, device pointers only.  Every rank sends exactly `cap`
   slots to every rank: the collectives take equal splits and nothing is read back to the host to size them.  A routed row is
   [w0 floats | w1 floats | pad] with a stride of (w0 + w1 + 3) & ~3 floats (buffers of nranks*cap such rows).  Per step:
     requester  bprx_route_reset(send_idx, nranks*cap, cursor, nranks)   send_idx <- -1 (unused slot), cursors <- 0
                bprx_route_plan(ids[n] global row ids, rows_per_rank = rows of a full shard, ...) -> slot[n] (owner*cap + position,
                  -1 and *overflow = 1 when the owner's bucket is full or the id is out of range), send_idx[slot] = owner-local row id
                all-to-all(send_idx) -> recv_idx: the rows the other ranks ask this rank for
     owner      bprx_route_gather(t0, w0, t1, w1, num_rows, recv_idx, nranks*cap, out): out[q] = [t0[idx] | t1[idx]] (w1 = 0: one table)
                all-to-all(out) -> got
     requester  bprx_route_unpack(got, slot, n, dst0, w0, dst1, w1): row r of the staging tables = got[slot[r]] (zero row for -1)
                ... local step (BPRX_FLAG_EXPORT_*_GRAD) ...
                bprx_route_pack(grad0, w0, grad1, w1, slot, n, send): send[slot[r]] = [grad0[r] | grad1[r]]; the gradient rows are
                  returned to zero (replaces bprx_clear_*_grad); all-to-all(send) -> back, aligned with recv_idx
     owner      bprx_route_scatter_add(t0, w0, t1, w1, num_rows, recv_idx, back, nranks*cap, scale): t[idx] += scale * row
                  (duplicates summed; unused slots skipped).
   Rows the requesting rank owns itself (my_rank; -1: none) never enter the send buffers: bprx_route_plan gives them slot
   -2 - local row id, bprx_route_unpack copies them from own0 / own1 (the rank's shard tables, own_rows rows) and bprx_route_pack
   adds scale * gradient into own0 / own1 directly.
   bprx_route_plan takes the ids as two arrays back to back (ids[n], then ids_b[n_b]; either may be empty): a triplet batch's
   positives and negatives need no concatenation.
   Row multiplicities (own_cnt / cnt: optional, int32 [rows_per_rank] each, all-zero at the start; NULL = every row is added with
   fp32 atomics): plan counts the requester's own rows into own_cnt, gather counts the rows the other ranks ask for into cnt; pack
   (own rows) and scatter_add (returned rows) then add a row that occurs ONCE with plain 16-byte read-modify-writes and leave the
   counts all-zero again by themselves (see bprx_route.hip).  Pass the same array to the pair (plan, pack) and another one to the
   pair (gather, scatter_add).
   bprx_route_pack with send_idx != NULL also returns send_idx (nslots entries) to -1 and the nranks cursors to 0 for the next
   step's plan: bprx_route_reset is then needed once, before the first step. */
BPRX_API int bprx_route_reset(int32_t *send_idx, int64_t nslots, int32_t *cursor, int32_t nranks, void *stream);
BPRX_API int bprx_route_plan(const int32_t *ids, int64_t n, const int32_t *ids_b, int64_t n_b, int32_t rows_per_rank, int32_t nranks,
                             int32_t cap, int32_t my_rank, int32_t *slot, int32_t *send_idx, int32_t *cursor, int32_t *overflow,
                             int32_t *own_cnt, void *stream);
BPRX_API int bprx_route_gather(const float *t0, int32_t w0, const float *t1, int32_t w1, int32_t num_rows, const int32_t *idx,
                               int64_t n, float *out, int32_t *cnt, void *stream);
BPRX_API int bprx_route_unpack(const float *got, const int32_t *slot, int64_t n, float *dst0, int32_t w0, float *dst1, int32_t w1,
                               const float *own0, const float *own1, int32_t own_rows, void *stream);
BPRX_API int bprx_route_pack(float *src0, int32_t w0, float *src1, int32_t w1, const int32_t *slot, int64_t n, float *send,
                             float *own0, float *own1, int32_t own_rows, float scale, int32_t *own_cnt, int32_t *send_idx,
                             int64_t nslots, int32_t *cursor, int32_t nranks, void *stream);
BPRX_API int bprx_route_scatter_add(float *t0, int32_t w0, float *t1, int32_t w1, int32_t num_rows, const int32_t *idx,
                                    const float *rows, int64_t n, float scale, int32_t *cnt, void *stream);

/* Model.predict_all() rows [u0,u1)   BPRMF.py:78-85 / VBPR.py:88-97.   out: fp32 [(u1-u0), I] */
BPRX_API int bprx_score_block(bprx_handle *h, int32_t u0, int32_t u1, float *out, void *stream);

/* Per-kernel timing with HIP events recorded on the caller's stream around every kernel of bprx_step
   (used by bench.py for the roofline figure; off by default, costs two event records per kernel when on).
   bprx_profile_read synchronises the pending events, ADDS the elapsed milliseconds and launch counts of each
   phase into ms[BPRX_PHASE_COUNT] / launches[BPRX_PHASE_COUNT] and clears the pending list. */
enum {
  BPRX_PHASE_CAST_ET = 0, BPRX_PHASE_PROJ_FWD = 1, BPRX_PHASE_TRIPLET = 2, BPRX_PHASE_PROJ_BWD = 3,
  BPRX_PHASE_REDUCE = 4, BPRX_PHASE_APPLY = 5, BPRX_PHASE_DENSE = 6, BPRX_PHASE_LOSS = 7, BPRX_PHASE_ITEM_SEG = 8,
  BPRX_PHASE_SEG_ALLOC = 9, BPRX_PHASE_ROW_COUNT = 10, BPRX_PHASE_ADAM_CATCHUP = 11, BPRX_PHASE_COUNT = 12
};
BPRX_API int bprx_profile_enable(bprx_handle *h, int on);
BPRX_API int bprx_profile_read(bprx_handle *h, double *ms, int64_t *launches);

/* Evaluator._eval_by_user on the device (Evaluator.py:82-128) for users [u0,u1) from their score rows
   (`scores` = the output of bprx_score_block for the same range).  CSR lists are DEVICE pointers: indptr int64 [U+1]
   indexed by the global user id, items int32.  out: double [(u1-u0), 5] = hr, prec, rec, auc, ndcg per user;
   out[.][0] == -1: the user has no held-out item (skipped by the reference, :88-89); == -2: more than 32 held-out
   items (use the host evaluator).  Exact integer rank counting: for identical fp32 scores the values equal the
   reference's, ties included. */
BPRX_API int bprx_eval_users(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, const int64_t *train_ptr,
                             const int32_t *train_items, const int64_t *eval_ptr, const int32_t *eval_items, int32_t K,
                             double *out, void *stream);

/* The same metrics for an ITEM-SHARDED model (train_rec --world_size N --shard item): every rank holds the score columns of its
   own items [item_lo, item_lo + num_items) of items_total; `scores` = bprx_score_block of the same user range on that rank.
   The CSR lists carry GLOBAL item ids.  Everything Evaluator._eval_by_user counts is additive over item shards:
     bprx_eval_pos     sp fp32 [(u1-u0), 32]: the score of held-out item t where this rank owns it, 0 elsewhere
                       -> the caller all-reduces (sum) sp: exact, one rank contributes each value
     bprx_eval_counts  counts int32 [(u1-u0), 65]: per held-out item #(own items >= sp_t) [0..32), #(own train-only items >= sp_t)
                       [32..64), and #(own train-only items) [64]        -> the caller all-reduces (sum) counts
     bprx_eval_finish  out double [(u1-u0), 5] from the summed counts, as bprx_eval_users (same markers -1 / -2): equal to
                       bprx_eval_users on the concatenated score row (Evaluator.py:96-126). */
BPRX_API int bprx_eval_pos(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, int32_t item_lo, int32_t items_total,
                           const int64_t *eval_ptr, const int32_t *eval_items, float *sp, void *stream);
BPRX_API int bprx_eval_counts(bprx_handle *h, int32_t u0, int32_t u1, const float *scores, int32_t item_lo, int32_t items_total,
                              const int64_t *train_ptr, const int32_t *train_items, const int64_t *eval_ptr,
                              const int32_t *eval_items, const float *sp, int32_t *counts, void *stream);
BPRX_API int bprx_eval_finish(bprx_handle *h, int32_t u0, int32_t u1, int32_t items_total, const int64_t *eval_ptr,
                              const float *sp, const int32_t *counts, int32_t K, double *out, void *stream);

/* Evaluator.store_recommendation on the device (Evaluator.py:225-239) for users [u0,u1): the train items of each user are
   overwritten with -inf IN `scores` (the output of bprx_score_block for the same range; the reference does the same to its
   score matrix, :233) and the K (<= 1024) largest remaining scores are returned best first: idx int32 [(u1-u0), K] (-1 past
   min(K, I)), val fp32 same shape.  flag int32 [(u1-u0)]: 1 = the row's list depends on how EQUAL scores are ordered (ties
   inside the list or at its boundary, or fewer than K unmasked items) -- the reference's order there is numpy's unstable
   argsort; the caller redoes flagged rows with numpy on the (already masked) row.  Unflagged rows equal the reference's
   output exactly.  CSR as in bprx_eval_users. */
BPRX_API int bprx_topk(bprx_handle *h, int32_t u0, int32_t u1, float *scores, const int64_t *train_ptr,
                       const int32_t *train_items, int32_t K, int32_t *idx, float *val, int32_t *flag, void *stream);

/* Measurement helper (bench.py): one launch of a plain streaming-read kernel over buf[0, bytes) (device memory, >= 64 MiB;
   sink: >= 8 KiB of device scratch).  Returns the number of bytes the launch reads, or a negative BPRX_E_* code.  Timed by
   the caller on `stream`: the rate this device's HBM delivers to a streaming kernel, quoted beside the 8 TB/s spec. */
BPRX_API int64_t bprx_probe_stream_read(const void *buf, int64_t bytes, void *sink, void *stream);
/* ... with `nt` (streaming, no-retain) loads, the policy of the bf16 feature passes */
BPRX_API int64_t bprx_probe_stream_read_nt(const void *buf, int64_t bytes, void *sink, void *stream);
/* Measurement helper: n lane groups each move one row of table[num_rows][row_floats] (fp32, row_floats a multiple of 4)
   picked by idx[] -- mode 0: read; mode 1: read and write back in place (idx distinct).  The access shape of the sparse
   kernels (a table row per index): the rate quoted beside their gather/scatter roofline.  Returns the bytes moved. */
BPRX_API int64_t bprx_probe_row_gather(void *table, int64_t num_rows, int32_t row_floats, const int32_t *idx, int64_t n,
                                       int32_t mode, void *sink, void *stream);

/* Synchronise `stream` and report deferred device-side errors (index out of range). */
BPRX_API int bprx_sync_check(bprx_handle *h, void *stream);

/* ---- index stream (HOST side) --------------------------------------------------------------------------
   DataLoader.all_triple_batches  dataset.py:83-114: per epoch random.shuffle(users) (Python MT19937), walk
   every positive of each user, negative by rejection on np.random.randint (NumPy-legacy MT19937).
   CSR of training_list on the HOST.  bprx_sampler_count = floor(N/bs)*bs*epochs (all epochs when that is 0).
   Output arrays are HOST int32 (the stream is inherently sequential; 12 B/triplet). */
BPRX_API int bprx_sampler_create(const int64_t *indptr, const int32_t *items, int32_t num_users, int32_t num_items,
                        bprx_sampler **out);
BPRX_API int bprx_sampler_destroy(bprx_sampler *s);
BPRX_API int64_t bprx_sampler_count(const bprx_sampler *s, int32_t batch_size, int32_t epochs);
BPRX_API int64_t bprx_sampler_ref_stream(bprx_sampler *s, int32_t batch_size, int32_t epochs, uint32_t py_seed,
                                uint32_t np_seed, int32_t *user, int32_t *pos, int32_t *neg, int64_t cap);

/* ---- throughput sampler (DEVICE side; not in the reference: its sampler tops out at ~4e5 triplets/s) ---------
   Stateless counter-based Philox4x32-10: triplet n = first + b of stream `seed` depends on (seed, n) only.
   Positive: uniform over the num_pos training interactions (pos_user[p], items_sorted[p]); negative: uniform over the
   items that are not positives of that user (rejection with binary search, <= 1024 attempts).  Replaces the role of
   dataset.py:83-122 for throughput runs; its distribution differs from the reference's epoch-permutation walk.
   All pointers are DEVICE pointers: indptr int64 [U+1], items_sorted int32 [num_pos] (ascending inside each user),
   pos_user int32 [num_pos]. */
BPRX_API int bprx_sample_philox(const int64_t *indptr, const int32_t *items_sorted, const int32_t *pos_user,
                                int64_t num_pos, int32_t num_items, uint64_t seed, uint64_t first, int64_t B,
                                int32_t *user, int32_t *pos, int32_t *neg, void *stream);

/* Epoch-walk mode of the device sampler (the reference's visiting order, dataset.py:93-107, as a stateless stream):
   in epoch `epoch` users come in the order perm[0..U) (see bprx_epoch_prepare) and every positive of a user is emitted once, consecutively;
   position n = first + b of the epoch belongs to the user a with epoch_ptr[a] <= n < epoch_ptr[a+1], epoch_ptr being the
   exclusive prefix sums of the list lengths in perm order (int64 [U+1], epoch_ptr[U] = number of interactions).
   pos_slot (optional, int32 [epoch_ptr[U]]): pos_slot[n] = that a, precomputed once per epoch; NULL = binary search per
   triplet (17 dependent loads at U = 100 000: 11.6 vs 6 us per batch of 65 536).
   The caller must keep first + B <= epoch_ptr[U] (a batch that crosses an epoch boundary is two calls).  Negatives as in
   bprx_sample_philox, keyed by (seed; n, epoch).  All pointers are DEVICE pointers. */
/* The user order of epoch `epoch`, prepared on the device without a sort and without the host:
     bprx_epoch_prepare  perm[a] (int32 [num_users]) = the user in slot a: a keyed permutation evaluated pointwise (4-round Feistel
                         network over 2*ceil(bits(U-1)/2) bits with Philox round functions keyed by (seed, epoch), cycle-walked
                         into [0, U); CPU twin: the oracle's orc_epoch_perm), and lens[a] (int64 [num_users]) = the length of
                         that user's list; the caller's inclusive prefix sums of lens behind a leading 0 are epoch_ptr;
     bprx_epoch_slots    pos_slot[p] = a for the positions p in [epoch_ptr[a], epoch_ptr[a+1]) (pos_slot: int32 [num_pos]).
   Three launches and one scan per epoch; an epoch switch never waits for the host. */
BPRX_API int bprx_epoch_prepare(uint64_t seed, uint32_t epoch, int32_t num_users, const int64_t *indptr, int32_t *perm,
                                int64_t *lens, void *stream);
BPRX_API int bprx_epoch_slots(const int64_t *epoch_ptr, int32_t num_users, int32_t *pos_slot, int64_t num_pos, void *stream);
BPRX_API int bprx_sample_epoch(const int64_t *indptr, const int32_t *items_sorted, const int32_t *perm,
                               const int64_t *epoch_ptr, const int32_t *pos_slot, int32_t num_users, int32_t num_items, uint64_t seed,
                               uint32_t epoch, int64_t first, int64_t B, int32_t *user, int32_t *pos, int32_t *neg,
                               void *stream);

/* The same two samplers, told which handle's NEXT step will consume the batch (h may be NULL: exactly the calls above).
   When that handle steps in segment mode (num_items up to 2 M), the sampler also writes the owner byte (id >> shift, shift = 8 up to
   65 536 items) and the local part (a byte, or 16 bits) of every sampled item id into planes the handle owns (2-3 B per occurrence); the index pass of the step called with exactly these
   pos / neg pointers and B == batch_size (a multiple of 16) then scans one byte per occurrence instead of four (its owner
   workgroups each read the whole batch).  batch_offset / batch_size: this call fills triplets [batch_offset, batch_offset + B)
   of a batch of batch_size, whose arrays start at user - batch_offset, pos - batch_offset, neg - batch_offset (an epoch
   crossing fills a batch in two calls; the call with batch_offset 0 comes first).  The caller must not change pos / neg between
   the sampler and the step; any other step on the handle simply ignores (and drops) the planes.  Results are identical. */
BPRX_API int bprx_sample_philox_h(bprx_handle *h, const int64_t *indptr, const int32_t *items_sorted, const int32_t *pos_user,
                                  int64_t num_pos, int32_t num_items, uint64_t seed, uint64_t first, int64_t B, int32_t *user,
                                  int32_t *pos, int32_t *neg, int64_t batch_offset, int64_t batch_size, void *stream);
BPRX_API int bprx_sample_epoch_h(bprx_handle *h, const int64_t *indptr, const int32_t *items_sorted, const int32_t *perm,
                                 const int64_t *epoch_ptr, const int32_t *pos_slot, int32_t num_users, int32_t num_items,
                                 uint64_t seed, uint32_t epoch, int64_t first, int64_t B, int32_t *user, int32_t *pos, int32_t *neg,
                                 int64_t batch_offset, int64_t batch_size, void *stream);
/* What the index pass of the handle's last step read: 0 = no segment-mode step yet, 1 = the int32 index arrays,
   2 = the sampler's byte planes.  (Introspection for tests and benchmarks.) */
BPRX_API int bprx_index_pass_kind(const bprx_handle *h);

#ifdef __cplusplus
}
#endif
#endif /* BPRX_H_ */
