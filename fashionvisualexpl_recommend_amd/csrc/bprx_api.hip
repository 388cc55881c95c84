// bprx_api.hip -- host side of the C ABI (include/bprx.h): handle, scratch, step orchestration.
#include <math.h>
#include <stdlib.h>
#include <new>

#include "bprx_internal.h"
#include <utility>

static char g_create_err[512] = "";

extern "C" int bprx_abi_version(void) { return BPRX_ABI_VERSION; }

extern "C" const char *bprx_last_error(const bprx_handle *h) { return h ? h->err : g_create_err; }

template <typename T>
static hipError_t dalloc_zero(T **p, size_t n) {
  *p = nullptr;
  if (n == 0) return hipSuccess;
  hipError_t e = hipMalloc((void **)p, n * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemset(*p, 0, n * sizeof(T));
}

static void graph_drop(bprx_handle *h) {
  for (int q = 0; q < h->graph_n; ++q)
    if (h->graph_ents[q].exec) (void)hipGraphExecDestroy(h->graph_ents[q].exec);
  h->graph_n = 0;
}

static bprx_handle::GraphSig graph_sig(const bprx_handle *h) {
  return {h->list_slot, h->slist_slot, h->qs_slot, h->seg_slot, h->et_valid, h->p_valid, h->absmax_valid, h->W_dirty,
          h->idx8_ready((const int32_t *)h->graph_key.i, (const int32_t *)h->graph_key.j, h->graph_key.B)};
}
static void graph_sig_apply(bprx_handle *h, const bprx_handle::GraphSig &g) {
  h->list_slot = g.list_slot; h->slist_slot = g.slist_slot; h->qs_slot = g.qs_slot; h->seg_slot = g.seg_slot;
  h->et_valid = g.et_valid; h->p_valid = g.p_valid; h->absmax_valid = g.absmax_valid; h->W_dirty = g.W_dirty;
}
static bool graph_sig_eq(const bprx_handle::GraphSig &a, const bprx_handle::GraphSig &b) {
  return a.list_slot == b.list_slot && a.slist_slot == b.slist_slot && a.qs_slot == b.qs_slot && a.seg_slot == b.seg_slot &&
         a.idx8 == b.idx8 &&
         a.et_valid == b.et_valid &&
         a.p_valid == b.p_valid && a.absmax_valid == b.absmax_valid && a.W_dirty == b.W_dirty;
}

static void free_scratch(bprx_handle *h) {
  void *ptrs[] = {h->dGu, h->dGi, h->dBi, h->dTu, h->flagU, h->flagI, h->lossb, h->loss_acc, h->errflag,
                  h->P,   h->W,   h->Wb, h->Ppair, h->Et, h->EtF, h->EtS, h->dEp, h->part, h->qs, h->Ft, h->seg_rank, h->seg_cnt, h->seg_ptr, h->seg_cursor, h->seg_lead, h->seg_ent, h->hot_done, h->uslot_of, h->ulist, h->uold, h->own8, h->loc8, h->cntU, h->cntI, h->ilist, h->ilist_n, h->lastU, h->lastI, h->lr_hist, h->slist, h->slist_n, h->msg_cursor, h->msg_next};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
}

extern "C" int bprx_create(const bprx_config *cfg, bprx_handle **out) {
#define CFAIL(code, ...)                                        \
  do {                                                          \
    snprintf(g_create_err, sizeof(g_create_err), __VA_ARGS__);  \
    return (code);                                              \
  } while (0)
  if (!cfg || !out) CFAIL(BPRX_E_INVALID, "bprx_create: null argument");
  *out = nullptr;
  if (cfg->abi_version != BPRX_ABI_VERSION)
    CFAIL(BPRX_E_INVALID, "bprx_create: abi_version %d, library is %d", cfg->abi_version, BPRX_ABI_VERSION);
  if (cfg->model != BPRX_MODEL_BPRMF && cfg->model != BPRX_MODEL_VBPR) CFAIL(BPRX_E_INVALID, "unknown model %d", cfg->model);
  if (cfg->optimizer != BPRX_OPT_SGD && cfg->optimizer != BPRX_OPT_ADAM_TF23)
    CFAIL(BPRX_E_INVALID, "unknown optimizer %d", cfg->optimizer);
  if (cfg->num_users <= 0 || cfg->num_items <= 0 || cfg->embed_k <= 0 || cfg->max_batch <= 0)
    CFAIL(BPRX_E_INVALID, "num_users, num_items, embed_k and max_batch must be positive");
  const bool vb = cfg->model == BPRX_MODEL_VBPR;
  if (vb) {
    if (cfg->embed_d <= 0 || cfg->feat_dim <= 0) CFAIL(BPRX_E_INVALID, "VBPR needs embed_d > 0 and feat_dim > 0");
    if (cfg->embed_d > 271) CFAIL(BPRX_E_INVALID, "embed_d %d > 271 unsupported", cfg->embed_d);
    if (cfg->feat_dtype != BPRX_F_FP32 && cfg->feat_dtype != BPRX_F_BF16 && cfg->feat_dtype != BPRX_F_FP8)
      CFAIL(BPRX_E_INVALID, "unknown feat_dtype");
    if (cfg->feat_dtype == BPRX_F_BF16 && cfg->feat_dim % 128 != 0)
      CFAIL(BPRX_E_INVALID, "bf16 features need feat_dim %% 128 == 0 (got %d)", cfg->feat_dim);
    if (cfg->feat_dtype == BPRX_F_FP8 && cfg->feat_dim % 256 != 0)
      CFAIL(BPRX_E_INVALID, "fp8 features need feat_dim %% 256 == 0 (got %d)", cfg->feat_dim);
    if (cfg->feat_dtype == BPRX_F_FP8 && !(cfg->feat_scale > 0.f)) CFAIL(BPRX_E_INVALID, "fp8 features need feat_scale > 0");
  }
  // exported gradients + adam_tf23: the handle takes the Adam steps of the rows it keeps; the exported side's owner applies its
  // rows' steps (bprx_apply_user_msgs, or bprx_adam_rows over its shard).  A rank may see an empty batch and must still move
  // every row it owns: the lazy form only (a sweep per step would have to run on ranks that launch nothing else).
  if ((cfg->flags & (BPRX_FLAG_EXPORT_USER_GRAD | BPRX_FLAG_EXPORT_ITEM_GRAD)) && cfg->optimizer != BPRX_OPT_SGD &&
      ((getenv("BPRX_ADAM_LAZY") && atoi(getenv("BPRX_ADAM_LAZY")) == 0) ||
       (!getenv("BPRX_ADAM_LAZY") && (cfg->flags & BPRX_FLAG_ADAM_SWEEP) && !(cfg->flags & BPRX_FLAG_ADAM_LAZY))))
    CFAIL(BPRX_E_INVALID, "BPRX_FLAG_EXPORT_*_GRAD with adam_tf23 needs the lazy form (BPRX_ADAM_LAZY != 0)");
  if ((cfg->flags & BPRX_FLAG_EXPORT_ITEM_GRAD) && cfg->model != BPRX_MODEL_BPRMF)
    CFAIL(BPRX_E_INVALID, "BPRX_FLAG_EXPORT_ITEM_GRAD is for BPRMF (VBPR keeps its items and features local)");
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) CFAIL(BPRX_E_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));

  bprx_handle *h = new (std::nothrow) bprx_handle();
  if (!h) CFAIL(BPRX_E_NOMEM, "out of host memory");
  memset(h, 0, sizeof(*h));
  h->cfg = *cfg;
  if (!vb) { h->cfg.embed_d = 0; h->cfg.feat_dim = 0; }
  const size_t U = cfg->num_users, I = cfg->num_items, k = cfg->embed_k, d = h->cfg.embed_d, D = h->cfg.feat_dim;
  const size_t MB = cfg->max_batch;
  bool ok = true;
#define A(call) ok = ok && ((e = (call)) == hipSuccess)
  A(dalloc_zero(&h->dGu, U * k));
  A(dalloc_zero(&h->dGi, I * k));
  A(dalloc_zero(&h->dBi, I));
  A(dalloc_zero(&h->flagU, U));
  A(dalloc_zero(&h->flagI, I));
  A(dalloc_zero(&h->lossb, MB));
  A(dalloc_zero(&h->loss_acc, (size_t)BPRX_DENSE_BLOCKS));
  A(dalloc_zero(&h->errflag, (size_t)1));
  if (cfg->flags & BPRX_FLAG_EXPORT_USER_GRAD) A(dalloc_zero(&h->msg_cursor, (size_t)2));
  A(dalloc_zero(&h->cntU, U));
  A(dalloc_zero(&h->cntI, I));
  if (vb) {
    h->PS = 16 * (int)((d + 1 + 15) / 16);
    const size_t PS = h->PS;
    // split-K of the backward projection: ONE 8-wave workgroup per CU over the D/256 column ranges (C2: 16 ranges x 16
    // item splits on 256 CUs).  More splits only add slab traffic (SK*D*PS*4 B written, then read by the dense update):
    // measured on C2 SK 16 / 32 with 3 / 2 tiles in flight: backward 73.8 / 77.0 us, dense update 9.7 / 12.1 us.
    {
      hipDeviceProp_t prop;
      const int ncu = hipGetDeviceProperties(&prop, cfg->device) == hipSuccess ? prop.multiProcessorCount : 256;
      const int mr = (int)((D + 255) / 256);
      h->SK = (ncu + mr - 1) / mr;
      // fp8 tiles are half the bytes, wide projections (more than 9 column tiles) have no registers for a third tile in
      // flight: both keep two workgroups per CU with two tiles in flight (measured: c2fp8 53 vs 58 us, c5 124 vs 362 us)
      // (wide projections, re-measured with the 8-wave kernel at two tiles in flight -- c5: SK 16 / 24 / 32 = 120 / 154 / 127 us
      //  backward and 24 / 30 / 38 us dense update: one workgroup per CU there as well)
      if (cfg->feat_dtype == BPRX_F_FP8 && PS / 16 <= 9) h->SK *= 2;
    }
    if (h->SK > 64) h->SK = 64;
    if (h->SK < 1) h->SK = 1;
    if (h->SK > 256) h->SK = 256;
    // BPRX_FWD_VARIANT=0: the plain forward kernel (the reference the streaming kernels are tested against); anything else:
    // the per-shape policy of bprx_proj.hip (launch_fwd_nt / launch_bwd_nt)
    h->fwd_variant = 4;
    if (const char *e = getenv("BPRX_FWD_VARIANT")) h->fwd_variant = atoi(e);
    A(dalloc_zero(&h->dTu, U * d));
    A(dalloc_zero(&h->P, I * PS));
    A(dalloc_zero(&h->W, I * PS));
    A(dalloc_zero((uint16_t **)&h->Wb, I * PS));
    A(dalloc_zero(&h->Ppair, MB * PS));
    A(dalloc_zero((uint16_t **)&h->Et, PS * D));
    A(dalloc_zero((uint16_t **)&h->EtF, PS * D));
    if (cfg->feat_dtype == BPRX_F_FP8 && PS / 16 >= 10) A(dalloc_zero((uint8_t **)&h->EtS, PS * D));
    A(dalloc_zero(&h->dEp, D * d + D));
    A(dalloc_zero(&h->part, (size_t)h->SK * D * PS));
    A(dalloc_zero(&h->qs, (size_t)4));
    if (cfg->feat_dtype != BPRX_F_FP32)   // tiled copy of F, item count padded to whole 32-item blocks
      A(dalloc_zero((uint8_t **)&h->Ft, (size_t)((I + 31) / 32 * 32) * D * (cfg->feat_dtype == BPRX_F_FP8 ? 1 : 2)));
  }
#undef A
  if (!ok) {
    snprintf(g_create_err, sizeof(g_create_err), "scratch allocation failed: %s", hipGetErrorString(e));
    free_scratch(h);
    delete h;
    return BPRX_E_NOMEM;
  }
  {
    // Item-side gradients through per-item occurrence segments (k_item_seg) instead of global float atomics: the
    // default whenever the row widths fit the 16-B-per-lane layout and the item rows are not staging rows of a sharded
    // run.  BPRX_ITEM_MODE=0 forces the atomic staging path (A/B measurements: profiles/r01_sweeps.md).
    const bool fits = k % 4 == 0 && d % 4 == 0 && k <= 256 && d <= 256;
    // seg_policy: 0 never, 1 per step (segments when the batch revisits items: 2B >= I; sparse batches keep the atomic
    // staging path with its in-place update of exclusive rows -- C3 shard: 0.104 vs 0.121 ms/step), 2 always
    h->seg_policy = (fits && !(cfg->flags & BPRX_FLAG_EXPORT_ITEM_GRAD)) ? 1 : 0;
    if (const char *e = getenv("BPRX_ITEM_MODE")) { const int v = atoi(e); h->seg_policy = h->seg_policy ? (v < 0 ? 0 : (v > 2 ? 2 : v)) : 0; }
    h->item_mode = 0;
    if (h->seg_policy) {
      // chunk list: the owners' regions (one slot per item + 4 per owner, <= 1024 owners; a last partial range) + the overflow list
      h->seg_lead_cap = (int64_t)(I + 8192 + 4 * 1024 + 2 * MB / 64 + 64 + 64);
      h->seg_ent_cap = (int64_t)(6 * MB + 64 * 1024 + 2048);
    // byte planes for the index pass (bprx_sample_*_h): at most 256 owners of 2^shift items; BPRX_IDX8=0: none (A/B)
    h->idx8_shift = 0;
    if (!(getenv("BPRX_IDX8") && atoi(getenv("BPRX_IDX8")) == 0))
      for (int sh = 8; sh <= 13 && !h->idx8_shift; ++sh)
        if ((((int64_t)I - 1) >> sh) <= 255) h->idx8_shift = sh;
      bool ok2 = dalloc_zero(&h->seg_rank, (size_t)2 * MB) == hipSuccess && dalloc_zero(&h->seg_cnt, I) == hipSuccess &&
                 dalloc_zero(&h->seg_ptr, I) == hipSuccess && dalloc_zero(&h->seg_cursor, (size_t)6) == hipSuccess &&
                 dalloc_zero((int4 **)&h->seg_lead, (size_t)h->seg_lead_cap) == hipSuccess &&
                 dalloc_zero(&h->hot_done, I) == hipSuccess && dalloc_zero((int2 **)&h->seg_ent, (size_t)h->seg_ent_cap) == hipSuccess &&
                 dalloc_zero(&h->uslot_of, U) == hipSuccess && dalloc_zero(&h->ulist, MB) == hipSuccess &&
                 dalloc_zero(&h->uold, MB * (k + d)) == hipSuccess &&
                 (!h->idx8_shift ||
                  (dalloc_zero(&h->own8, (size_t)2 * MB) == hipSuccess && dalloc_zero(&h->loc8, (size_t)2 * MB * (h->idx8_shift > 8 ? 2 : 1)) == hipSuccess));
      if (!ok2) {
        snprintf(g_create_err, sizeof(g_create_err), "segment scratch allocation failed");
        free_scratch(h);
        delete h;
        return BPRX_E_NOMEM;
      }
    }
  }
  // Touched-item list (sparse batches): when the batch touches few of the items, both projections run over the batch's
  // distinct items only (the reference gathers 2B feature rows per step, VBPR.py:78; its own default is --batch_size 256,
  // train_rec.py:23) instead of streaming all of F twice.  Per step: list mode iff 2B < I, i.e. whenever the batch cannot touch every item (measured on C2's tables: B = 16 384 / 24 576:
  // 0.182 / 0.223 ms with the list, 0.231 / 0.239 ms streaming; B = 32 768 = 2B >= I: 0.287 vs 0.240 ms with occurrence segments);
  // BPRX_LIST_MODE = 0 never / 1 per step / 2 always.
  h->list_policy = vb ? 1 : 0;
  if (const char *e = getenv("BPRX_LIST_MODE")) { const int v = atoi(e); h->list_policy = vb ? (v < 0 ? 0 : (v > 2 ? 2 : v)) : 0; }
  if (h->list_policy) {
    const size_t cap = 2 * MB < I ? 2 * MB : I;
    if (dalloc_zero(&h->ilist, cap) != hipSuccess || dalloc_zero(&h->ilist_n, (size_t)2) != hipSuccess) {
      snprintf(g_create_err, sizeof(g_create_err), "item list allocation failed");
      free_scratch(h);
      delete h;
      return BPRX_E_NOMEM;
    }
  }
  h->SK_step = h->SK;
  // adam_tf23: lazy-exact form by default (rows are replayed when read; BPRX_ADAM_LAZY=0 = the whole-table sweeps)
  // adam_tf23: lazily exact (per-row replay on touch) or by whole-table sweeps -- the same arithmetic either way.  A row's replay
  // is a serial recurrence over the steps since its last touch: with ~20 positives per user that is 20 U / B steps of ~0.45 us
  // each, against a sweep that moves every row's (p, m, v) once per step.  Large batches (C2: 30 steps, 14 us, against a 96-us
  // sweep) want the replay; the reference's own defaults (batch 256: 1 562 steps = 700 us, against a 25-us sweep of its small
  // tables) want the sweeps: measured on the CLI, 20 000 x 10 000, an epoch of 1 562 steps takes 0.76 s lazily and 0.22 s with
  // sweeps (BPRMF 0.46 / 0.10).  BPRX_ADAM_LAZY=0 / 1 forces either; exported user gradients (multi-GPU) need the lazy form.
  h->adam_lazy = false;
  if (cfg->optimizer == BPRX_OPT_ADAM_TF23) {
    // (0.45 us per replayed step: k_adam_catchup on C2's tables takes 86 / 235 / 702 us at replay depths 122 / 488 / 1953 =
    //  batches of 16 384 / 4 096 / 1 024, where the sweeps take 81-88 us: measured crossover between 16 384 and 4 096)
    const double chain_us = 20.0 * (double)cfg->num_users / (double)cfg->max_batch * 0.45;
    const double sweep_us = ((double)cfg->num_users * (cfg->embed_k + h->cfg.embed_d) + (double)cfg->num_items * (cfg->embed_k + 1)) * 24.0 / 4e6;
    h->adam_lazy = (cfg->flags & (BPRX_FLAG_EXPORT_USER_GRAD | BPRX_FLAG_EXPORT_ITEM_GRAD)) ? true : chain_us < sweep_us;
    if (cfg->flags & BPRX_FLAG_ADAM_SWEEP) h->adam_lazy = false;
    if (cfg->flags & BPRX_FLAG_ADAM_LAZY) h->adam_lazy = true;
    if (const char *e = getenv("BPRX_ADAM_LAZY")) h->adam_lazy = atoi(e) != 0;
  }
  if (h->adam_lazy) {
    if (dalloc_zero(&h->lastU, U) != hipSuccess || dalloc_zero(&h->lastI, I) != hipSuccess ||
        dalloc_zero(&h->lr_hist, (size_t)bprx_adam_hist()) != hipSuccess) {
      snprintf(g_create_err, sizeof(g_create_err), "adam bookkeeping allocation failed");
      free_scratch(h);
      delete h;
      return BPRX_E_NOMEM;
    }
  }
  // exclusive-row fast path: sgd only (adam sweeps every row anyway); not with exported user gradients
  h->fast_rows = (cfg->optimizer == BPRX_OPT_SGD && !(cfg->flags & BPRX_FLAG_EXPORT_USER_GRAD)) ? 1 : 0;   // per side: make_args
  if (h->fast_rows && !(cfg->flags & (BPRX_FLAG_EXPORT_USER_GRAD | BPRX_FLAG_EXPORT_ITEM_GRAD))) {
    if (dalloc_zero(&h->slist, (size_t)3 * MB) != hipSuccess || dalloc_zero(&h->slist_n, (size_t)2) != hipSuccess) {
      snprintf(g_create_err, sizeof(g_create_err), "shared-row list allocation failed");
      free_scratch(h);
      delete h;
      return BPRX_E_NOMEM;
    }
  }
  if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "side stream / event creation failed");
    free_scratch(h);
    delete h;
    return BPRX_E_HIP;
  }
  // BPRX_SIDE_STREAM (bit mask).  Measured on C2:
  //   1  the sparse optimizer pass beside the backward projection: SLOWER (0.385 vs 0.363 ms/step; both are
  //      bandwidth-bound and interfere: proj_bwd 93 -> 144 us, apply 35 -> 64 us)
  //   (2, removed: the segment-mode index pass beside the forward projection.  Round 2's pass, 130 K memory-side count atomics:
  //      0.2599 vs 0.2600 ms, both kernels stretched; round 3's k_index_seg, LDS counting: 0.238 vs 0.229 ms on a plain side
  //      stream -- its workgroups share every CU's memory pipeline with the projection's streaming loads and take 84 us instead
  //      of 19 -- and 0.250 / 0.246 vs 0.224 ms on a stream whose CU mask confines it to 16 / 32 CUs: with few owner workgroups
  //      each of them matches 16x / 8x more of the scanned values and the rare-match path of the scan becomes its bulk)
  //   4  lazy Adam's catch-up (ALU-bound: correctly rounded sqrt / divide per replayed element and step) beside the
  //      HBM-bound forward projection of a streaming step: adam_tf23 0.330 -> 0.317 ms/step.  The default with adam_tf23.
  {
    const char *e = getenv("BPRX_SIDE_STREAM");
    h->side_mode = e ? atoi(e) : ((vb && cfg->optimizer == BPRX_OPT_ADAM_TF23 && h->adam_lazy) ? 4 : 0);
    if (!h->side_mode) { (void)hipStreamDestroy(h->side); h->side = nullptr; }
  }
  {
    hipDeviceProp_t prop;
    h->num_cu = hipGetDeviceProperties(&prop, cfg->device) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  // Measured (ROCm 7.0, bench.py on a non-default stream): replaying the step as a hipGraph is SLOWER than the plain
  // launches it replaces, for large batches (C2, B = 65 536: 0.2828 vs 0.2747 ms/step) and for small ones alike (C2 tables,
  // B = 256: 0.0576 vs 0.0529; B = 4096: 0.0944 vs 0.0906; BPRMF B = 256: 0.0256 vs 0.0261 ms/step) -- the host is not the
  // limiter: a small step is six dependent kernels of 5-10 us each (chains of 3-5 memory round trips), and a graph
  // launch costs more than the dispatch gaps it removes.  Off by default.  BPRX_GRAPH=1: always; 2: steps of B <= 8192.
  h->graph_mode = 0;
  if (const char *e = getenv("BPRX_GRAPH")) h->graph_mode = atoi(e);
  if (h->graph_mode < 0 || h->graph_mode > 2) h->graph_mode = 0;
  h->prof_pending = new std::vector<bprx_handle::ProfRec>();
  h->prof_free = new std::vector<hipEvent_t>();
  *out = h;
  return BPRX_OK;
#undef CFAIL
}

extern "C" int bprx_destroy(bprx_handle *h) {
  if (!h) return BPRX_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->side) (void)hipStreamSynchronize(h->side);
  free_scratch(h);
  graph_drop(h);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->prof_pending) { for (auto &r : *h->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } delete h->prof_pending; }
  if (h->prof_free) { for (auto e : *h->prof_free) (void)hipEventDestroy(e); delete h->prof_free; }
  delete h;
  return BPRX_OK;
}

extern "C" int bprx_profile_enable(bprx_handle *h, int on) {
  if (!h) return BPRX_E_INVALID;
  h->prof = on != 0;
  return BPRX_OK;
}

extern "C" int bprx_profile_read(bprx_handle *h, double *ms, int64_t *launches) {
  if (!h || !ms || !launches) return BPRX_E_INVALID;
  for (auto &r : *h->prof_pending) {
    BPRX_HIP(h, hipEventSynchronize(r.b));
    float t = 0.f;
    BPRX_HIP(h, hipEventElapsedTime(&t, r.a, r.b));
    ms[r.phase] += (double)t;
    launches[r.phase] += 1;
    h->prof_free->push_back(r.a);
    h->prof_free->push_back(r.b);
  }
  h->prof_pending->clear();
  return BPRX_OK;
}

extern "C" int bprx_bind_tables(bprx_handle *h, const bprx_tables *t) {
  if (!h || !t) return BPRX_E_INVALID;
  if (!t->Gu || !t->Gi || !t->Bi) BPRX_FAIL(h, BPRX_E_INVALID, "bind_tables: Gu, Gi, Bi are required");
  const bool vb = h->cfg.model == BPRX_MODEL_VBPR;
  if (vb && (!t->Tu || !t->F || !t->E || !t->Bp)) BPRX_FAIL(h, BPRX_E_INVALID, "bind_tables: VBPR needs Tu, F, E, Bp");
  if (h->cfg.optimizer == BPRX_OPT_ADAM_TF23) {
    if (!t->m_Gu || !t->v_Gu || !t->m_Gi || !t->v_Gi || !t->m_Bi || !t->v_Bi)
      BPRX_FAIL(h, BPRX_E_INVALID, "bind_tables: adam_tf23 needs m_/v_ slots for Gu, Gi, Bi");
    if (vb && (!t->m_Tu || !t->v_Tu || !t->m_E || !t->v_E || !t->m_Bp || !t->v_Bp))
      BPRX_FAIL(h, BPRX_E_INVALID, "bind_tables: adam_tf23 needs m_/v_ slots for Tu, E, Bp");
  }
  if (((uintptr_t)t->Gu | (uintptr_t)t->Gi | (uintptr_t)t->Tu | (uintptr_t)t->F | (uintptr_t)t->E) & 15)
    BPRX_FAIL(h, BPRX_E_INVALID, "bind_tables: table base pointers must be 16-byte aligned");
  h->t = *t;
  h->et_valid = h->p_valid = h->absmax_valid = false;
  graph_drop(h);                            // a captured step holds the old table pointers
  {
    const int rc = bprx_launch_tile_F(h);   // the projections read a tiled copy of the frozen F (made here, once)
    if (rc) return rc;
  }
  h->bound = true;
  return bprx_launch_adam_reset(h, h->adam_t, 0);   // every bound row counts as current at optimizer.iterations
}

extern "C" int bprx_tables_dirty(bprx_handle *h, void *stream) {
  if (!h) return BPRX_E_INVALID;
  h->et_valid = h->p_valid = h->absmax_valid = false;
  // outside values are current by definition; the bookkeeping reset is ordered on the caller's stream, behind whatever
  // wrote the tables there (a NULL stream is synchronised instead)
  return h->bound ? bprx_launch_adam_reset(h, h->adam_t, (hipStream_t)stream) : BPRX_OK;
}

extern "C" int bprx_set_hyper(bprx_handle *h, float lr, float reg) {
  if (!h) return BPRX_E_INVALID;
  h->cfg.lr = lr;
  h->cfg.reg = reg;
  return BPRX_OK;
}

extern "C" int bprx_set_adam_step(bprx_handle *h, int64_t it, void *stream) {
  if (!h || it < 0) return BPRX_E_INVALID;
  h->adam_t = it;
  return h->bound ? bprx_launch_adam_reset(h, it, (hipStream_t)stream) : BPRX_OK;
}

extern "C" int64_t bprx_get_adam_step(const bprx_handle *h) { return h ? h->adam_t : -1; }
extern "C" int bprx_adam_is_lazy(const bprx_handle *h) { return h ? (h->adam_lazy ? 1 : 0) : BPRX_E_INVALID; }

static int check_ready(bprx_handle *h, int64_t B) {
  if (!h) return BPRX_E_INVALID;
  if (!h->bound) BPRX_FAIL(h, BPRX_E_STATE, "tables not bound (call bprx_bind_tables first)");
  if (B < 0 || B > h->cfg.max_batch) BPRX_FAIL(h, BPRX_E_INVALID, "B=%lld outside [0, max_batch=%lld]", (long long)B, (long long)h->cfg.max_batch);
  return BPRX_OK;
}

extern "C" int bprx_sync_adam(bprx_handle *h, void *stream) {
  int rc = check_ready(h, 0);
  if (rc) return rc;
  return bprx_launch_adam_sync(h, h->adam_t, (hipStream_t)stream);
}

extern "C" int bprx_score_pairs(bprx_handle *h, const int32_t *user, const int32_t *item, int64_t B, float *x, void *stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if (B == 0) return BPRX_OK;
  if (!user || !item || !x) BPRX_FAIL(h, BPRX_E_INVALID, "score_pairs: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if ((rc = bprx_launch_adam_sync(h, h->adam_t, s))) return rc;          // lazy adam: the rows must be current
  if (h->cfg.model == BPRX_MODEL_VBPR) {
    if (h->p_valid) return bprx_launch_score(h, user, item, B, nullptr, 0, x, s);      // every item's projection is at hand
    if ((rc = bprx_launch_cast_Et(h, s))) return rc;
    if ((rc = bprx_launch_proj_fwd(h, item, B, nullptr, 0, h->Ppair, s))) return rc;     // one projection row per pair
    return bprx_launch_score(h, user, item, B, h->Ppair, 1, x, s);
  }
  return bprx_launch_score(h, user, item, B, nullptr, 0, x, s);
}

// First half of bprx_step_begin: index pass, item projections, per-triplet gradients.  Afterwards the USER-side gradients
// of the batch are final (staging tables with BPRX_FLAG_EXPORT_USER_GRAD): a replicated-user step packs and all-gathers
// them while bprx_step_begin_dense (item rows, W, dE|dBp = F^T W) is still running.
extern "C" int bprx_step_begin_sparse(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B, void *stream) {
  int rc = check_ready(h, B);
  if (rc) return rc;
  if (h->pending_stage) BPRX_FAIL(h, BPRX_E_STATE, "step_begin called twice without step_end");
  hipStream_t s = (hipStream_t)stream;
  const bool vb = h->cfg.model == BPRX_MODEL_VBPR;
  if (B == 0) {
    // A rank of a replicated-user multi-GPU step whose item shard holds no positive of this global batch: it contributes an
    // empty message and a zero dense gradient, but takes part in every collective and takes the same optimizer step as the
    // other replicas (bprx_pack_user_msg -> count 0, bprx_step_begin_dense -> dE|dBp = 0, bprx_apply_user_msgs, bprx_step_end).
    if (!(h->cfg.flags & (BPRX_FLAG_EXPORT_USER_GRAD | BPRX_FLAG_EXPORT_ITEM_GRAD))) BPRX_FAIL(h, BPRX_E_INVALID, "step: empty batch");
    h->list_mode = 0; h->item_mode = 0;
    if (h->cfg.optimizer == BPRX_OPT_ADAM_TF23) {
      h->adam_t += 1;
      const float t = (float)h->adam_t;
      const float lr_t = h->cfg.lr * sqrtf(1.0f - powf(h->cfg.beta2, t)) / (1.0f - powf(h->cfg.beta1, t));
      if (h->adam_lazy) {
        if (h->adam_t - h->adam_synced >= bprx_adam_hist() - 2 && (rc = bprx_launch_adam_sync(h, h->adam_t - 1, s))) return rc;
        if ((rc = bprx_launch_adam_catchup(h, nullptr, nullptr, nullptr, 0, lr_t, s))) return rc;   // records lr_t of this step
      }
      h->pend_lr = lr_t;
    } else h->pend_lr = h->cfg.lr;
    h->pending_B = 0; h->pending_stage = 1;
    h->pend_u = h->pend_i = h->pend_j = nullptr;
    return BPRX_OK;
  }
  if (!user || !pos || !neg) BPRX_FAIL(h, BPRX_E_INVALID, "step: null index pointer");
  // list mode: both projections over the batch's distinct items only (needs the index pass BEFORE the forward projection)
  h->list_mode = vb && !h->proj_fresh && (h->list_policy == 2 || (h->list_policy == 1 && 2 * B < (int64_t)h->cfg.num_items));
  h->item_mode = !h->list_mode && (h->seg_policy == 2 || (h->seg_policy == 1 && 2 * B >= (int64_t)h->cfg.num_items));
  h->list_reset_cnt = !(h->fast_rows && !(h->cfg.flags & BPRX_FLAG_EXPORT_ITEM_GRAD));
  // byte planes of this very batch, left by bprx_sample_*_h (consumed here, whatever this step does with them)
  h->idx8_use = h->idx8_ready(pos, neg, B);
  h->idx8_n = 0;
  float lr_t = h->cfg.lr;
  bool catchup_aside = false;     // BPRX_SIDE_STREAM & 4: the (ALU-bound) lazy-Adam catch-up runs beside the (HBM-bound) projection
  if (h->cfg.optimizer == BPRX_OPT_ADAM_TF23) {
    h->adam_t += 1;
    float t = (float)h->adam_t;
    lr_t = h->cfg.lr * sqrtf(1.0f - powf(h->cfg.beta2, t)) / (1.0f - powf(h->cfg.beta1, t));
    if (h->adam_lazy) {
      // the ring holds lr_s of the last ADAM_HIST steps: before it would wrap, everything is caught up (amortised: one
      // sweep per ~8000 steps); then the rows of THIS batch are brought to step t-1 for the forward pass
      if (h->adam_t - h->adam_synced >= bprx_adam_hist() - 2 && (rc = bprx_launch_adam_sync(h, h->adam_t - 1, s))) return rc;
      catchup_aside = vb && !h->proj_fresh && !h->list_mode && !h->p_valid && h->side && (h->side_mode & 4);
      if (!catchup_aside && (rc = bprx_launch_adam_catchup(h, user, pos, neg, B, lr_t, s))) return rc;
    }
  }
  if (catchup_aside) {                                     // work that does not depend on P, beside the projection
    BPRX_HIP(h, hipEventRecord(h->ev_fork, s));
    BPRX_HIP(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if ((rc = bprx_launch_adam_catchup(h, user, pos, neg, B, lr_t, h->side))) return rc;
    BPRX_HIP(h, hipEventRecord(h->ev_join, h->side));
  }
  if (h->list_mode) {
    h->list_cur = h->ilist_n + h->list_slot;
    h->list_bound = 2 * B < (int64_t)h->cfg.num_items ? 2 * B : (int64_t)h->cfg.num_items;
    if ((rc = bprx_launch_cast_Et(h, s))) return rc;
    if ((rc = bprx_launch_index_pass(h, user, pos, neg, B, s))) return rc;              // counts + the distinct-item list
    if (!h->p_valid &&                                                                   // P rows of the listed items only
        (rc = bprx_launch_proj_fwd(h, h->ilist, h->list_bound, h->list_cur, 1, h->P, s))) return rc;
  } else if (vb && !h->proj_fresh) {
    if ((rc = bprx_launch_cast_Et(h, s))) return rc;
    if (!h->p_valid && (rc = bprx_launch_proj_fwd(h, nullptr, h->cfg.num_items, nullptr, 0, h->P, s))) return rc;  // every item
  }
  h->proj_fresh = false;
  if (catchup_aside) BPRX_HIP(h, hipStreamWaitEvent(s, h->ev_join, 0));
  if (!h->list_mode && (rc = bprx_launch_index_pass(h, user, pos, neg, B, s))) return rc;
  if ((rc = bprx_launch_triplet_grad(h, user, pos, neg, B, s))) return rc;
  h->pending_B = B;
  h->pending_stage = 1;
  h->pend_u = user; h->pend_i = pos; h->pend_j = neg; h->pend_lr = lr_t;
  return BPRX_OK;
}

extern "C" int bprx_step_begin_dense(bprx_handle *h, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (h->pending_stage != 1) BPRX_FAIL(h, BPRX_E_STATE, "step_begin_dense without step_begin_sparse");
  hipStream_t s = (hipStream_t)stream;
  const bool vb = h->cfg.model == BPRX_MODEL_VBPR;
  if (h->pending_B == 0) {                                 // empty batch of a replicated-user rank: zero dense gradient
    if (vb) BPRX_HIP(h, hipMemsetAsync(h->dEp, 0, ((size_t)h->cfg.feat_dim * (h->cfg.embed_d + 1)) * sizeof(float), s));
    h->pending_stage = 2;
    return BPRX_OK;
  }
  const int32_t *user = h->pend_u, *pos = h->pend_i, *neg = h->pend_j;
  const int64_t B = h->pending_B;
  const float lr_t = h->pend_lr;
  int rc;
  if ((rc = bprx_launch_item_seg(h, pos, neg, B, lr_t, s))) return rc;                  // item rows + W, no float atomics
  // sparse tables are final now: their optimizer pass does not depend on the dense all-reduce, nor on the backward
  // projection -- with VBPR it runs on the side stream beside it
  if (vb && h->side && (h->side_mode & 1)) {
    BPRX_HIP(h, hipEventRecord(h->ev_fork, s));
    BPRX_HIP(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    if ((rc = bprx_launch_apply(h, user, pos, neg, B, lr_t, h->side))) return rc;
    BPRX_HIP(h, hipEventRecord(h->ev_join, h->side));
    h->side_pending = true;
    if ((rc = bprx_launch_proj_bwd(h, B, s))) return rc;                                // dE|dBp = F^T W
  } else {
    if (vb && (rc = bprx_launch_proj_bwd(h, B, s))) return rc;
    if ((rc = bprx_launch_apply(h, user, pos, neg, B, lr_t, s))) return rc;
  }
  h->pending_stage = 2;
  return BPRX_OK;
}

extern "C" int bprx_step_begin(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B, void *stream) {
  int rc = bprx_step_begin_sparse(h, user, pos, neg, B, stream);
  if (!rc && (rc = bprx_step_begin_dense(h, stream))) { h->pending_B = 0; h->pending_stage = 0; }   // a failed _begin leaves no step pending
  return rc;
}

extern "C" int bprx_step_project(bprx_handle *h, void *stream) {
  int rc = check_ready(h, 0);
  if (rc) return rc;
  if (h->cfg.model != BPRX_MODEL_VBPR) return BPRX_OK;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = bprx_launch_cast_Et(h, s))) return rc;
  if (!h->p_valid && (rc = bprx_launch_proj_fwd(h, nullptr, h->cfg.num_items, nullptr, 0, h->P, s))) return rc;
  h->p_valid = true;
  h->proj_fresh = true;
  return BPRX_OK;
}

extern "C" int bprx_user_grad(bprx_handle *h, float **dGu, float **dTu) {
  if (!h || !dGu || !dTu) return BPRX_E_INVALID;
  *dGu = h->dGu;
  *dTu = h->dTu;
  return BPRX_OK;
}

extern "C" int bprx_clear_user_grad(bprx_handle *h, int64_t n_rows, int32_t marks_only, void *stream) {
  if (!h || n_rows < 0 || n_rows > h->cfg.num_users) return BPRX_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (!marks_only) {                                       // (bprx_route_pack has already returned the gradient rows to zero)
    BPRX_HIP(h, hipMemsetAsync(h->dGu, 0, (size_t)n_rows * h->cfg.embed_k * sizeof(float), s));
    if (h->cfg.embed_d) BPRX_HIP(h, hipMemsetAsync(h->dTu, 0, (size_t)n_rows * h->cfg.embed_d * sizeof(float), s));
  }
  BPRX_HIP(h, hipMemsetAsync(h->flagU, 0, (size_t)n_rows * sizeof(uint32_t), s));
  return BPRX_OK;
}

extern "C" int bprx_item_grad(bprx_handle *h, float **dGi, float **dBi) {
  if (!h || !dGi || !dBi) return BPRX_E_INVALID;
  *dGi = h->dGi;
  *dBi = h->dBi;
  return BPRX_OK;
}

extern "C" int bprx_clear_item_grad(bprx_handle *h, int64_t n_rows, int32_t marks_only, void *stream) {
  if (!h || n_rows < 0 || n_rows > h->cfg.num_items) return BPRX_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (!marks_only) {
    BPRX_HIP(h, hipMemsetAsync(h->dGi, 0, (size_t)n_rows * h->cfg.embed_k * sizeof(float), s));
    BPRX_HIP(h, hipMemsetAsync(h->dBi, 0, (size_t)n_rows * sizeof(float), s));
  }
  BPRX_HIP(h, hipMemsetAsync(h->flagI, 0, (size_t)n_rows * sizeof(uint32_t), s));
  return BPRX_OK;
}

extern "C" int bprx_dense_grad(bprx_handle *h, float **ptr, int64_t *count) {
  if (!h || !ptr || !count) return BPRX_E_INVALID;
  *ptr = h->dEp;
  *count = h->cfg.model == BPRX_MODEL_VBPR ? (int64_t)h->cfg.feat_dim * (h->cfg.embed_d + 1) : 0;
  return BPRX_OK;
}

extern "C" int bprx_step_end(bprx_handle *h, float *loss_out, void *stream) {
  if (!h) return BPRX_E_INVALID;
  if (!h->pending_stage) BPRX_FAIL(h, BPRX_E_STATE, "step_end without step_begin");
  if (h->pending_stage != 2) BPRX_FAIL(h, BPRX_E_STATE, "step_end before step_begin_dense");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  float lr_t = h->cfg.lr;
  if (h->cfg.optimizer == BPRX_OPT_ADAM_TF23) {
    float t = (float)h->adam_t;
    lr_t = h->cfg.lr * sqrtf(1.0f - powf(h->cfg.beta2, t)) / (1.0f - powf(h->cfg.beta1, t));
  }
  int64_t B = h->pending_B;
  h->pending_B = 0;
  h->pending_stage = 0;
  if (h->side_pending) {                                  // join the side stream (sparse optimizer pass)
    BPRX_HIP(h, hipStreamWaitEvent(s, h->ev_join, 0));
    h->side_pending = false;
  }
  if (h->cfg.model == BPRX_MODEL_VBPR && (rc = bprx_launch_dense_update(h, lr_t, s))) return rc;
  if (loss_out && (rc = bprx_launch_loss_reduce(h, B, loss_out, s))) return rc;
  return BPRX_OK;
}

static int step_plain(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B, float *loss_out,
                      void *stream) {
  h->fused_reduce = true;          // no all-reduce in between: the dense update sums the split-K slabs itself
  int rc = bprx_step_begin(h, user, pos, neg, B, stream);
  if (!rc) rc = bprx_step_end(h, loss_out, stream);
  h->fused_reduce = false;
  return rc;
}

extern "C" int bprx_step(bprx_handle *h, const int32_t *user, const int32_t *pos, const int32_t *neg, int64_t B,
                         float *loss_out, void *stream) {
  if (!h) return BPRX_E_INVALID;
  // The sgd step is a fixed sequence of launches whose arguments repeat from call to call (index buffers, loss scalar,
  // stream): captured into a hipGraph and replayed, it is ONE launch per step (see graph_mode in bprx_internal.h for
  // when that pays).  Not for adam (lr_t changes every step), not while per-kernel profiling is on, not on the legacy
  // default stream (cannot be captured): those take the plain path.
  const bool can_graph = (h->graph_mode == 1 || (h->graph_mode == 2 && B <= 8192)) && h->cfg.optimizer == BPRX_OPT_SGD &&
                         !h->prof && stream != nullptr && !h->side_mode && h->bound && B > 0 && B <= h->cfg.max_batch && user && pos &&
                         neg && !h->proj_fresh && !h->pending_stage;
  if (!can_graph) return step_plain(h, user, pos, neg, B, loss_out, stream);
  hipStream_t s = (hipStream_t)stream;
  const bool same = h->graph_key.u == user && h->graph_key.i == pos && h->graph_key.j == neg && h->graph_key.loss == loss_out &&
                    h->graph_key.B == B && h->graph_key.stream == stream && h->graph_key.lr == h->cfg.lr &&
                    h->graph_key.reg == h->cfg.reg;
  if (!same) {
    // capture only when a call repeats the previous call's arguments (a caller that walks through a pre-generated
    // stream passes new pointers every step and must not pay for a capture each time)
    graph_drop(h);
    h->graph_key.u = user; h->graph_key.i = pos; h->graph_key.j = neg; h->graph_key.loss = loss_out; h->graph_key.B = B;
    h->graph_key.stream = stream; h->graph_key.lr = h->cfg.lr; h->graph_key.reg = h->cfg.reg;
    return step_plain(h, user, pos, neg, B, loss_out, stream);
  }
  const bprx_handle::GraphSig in = graph_sig(h);
  for (int q = 0; q < h->graph_n; ++q)
    if (graph_sig_eq(h->graph_ents[q].in, in)) {
      BPRX_HIP(h, hipGraphLaunch(h->graph_ents[q].exec, s));
      h->idx8_n = 0;                                         // (byte planes of this batch, if any: consumed)
      graph_sig_apply(h, h->graph_ents[q].out);              // what the launches of the captured step left on the host
      return BPRX_OK;
    }
  if (h->graph_n == 4) graph_drop(h);                        // the host state wandered (outside reads between steps): start over
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    h->graph_mode = 0;                                       // this stream cannot be captured: plain launches from now on
    return step_plain(h, user, pos, neg, B, loss_out, stream);
  }
  const int rc = step_plain(h, user, pos, neg, B, loss_out, stream);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &g);
  hipGraphExec_t exec = nullptr;
  const bool ok = !rc && e == hipSuccess && g && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess && exec;
  if (g) (void)hipGraphDestroy(g);
  if (!ok) {
    (void)hipGetLastError();
    h->graph_mode = 0;
    h->pending_B = 0; h->pending_stage = 0;
    graph_sig_apply(h, in);                                  // nothing ran: back to the state before the capture
    return rc ? rc : step_plain(h, user, pos, neg, B, loss_out, stream);
  }
  h->graph_ents[h->graph_n++] = {exec, in, graph_sig(h)};
  BPRX_HIP(h, hipGraphLaunch(exec, s));
  return BPRX_OK;
}

extern "C" int bprx_score_block(bprx_handle *h, int32_t u0, int32_t u1, float *out, void *stream) {
  int rc = check_ready(h, 0);
  if (rc) return rc;
  if (u0 < 0 || u1 > h->cfg.num_users || u0 > u1 || !out) BPRX_FAIL(h, BPRX_E_INVALID, "score_block: bad user range [%d,%d)", u0, u1);
  if (u0 == u1) return BPRX_OK;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = bprx_launch_adam_sync(h, h->adam_t, s))) return rc;          // lazy adam: predict_all reads every row
  if (h->cfg.model == BPRX_MODEL_VBPR && !h->p_valid) {   // P = F.[E|Bp] once per parameter state, not once per user block
    if ((rc = bprx_launch_cast_Et(h, s))) return rc;
    if ((rc = bprx_launch_proj_fwd(h, nullptr, h->cfg.num_items, nullptr, 0, h->P, s))) return rc;
    h->p_valid = true;
  }
  if (h->cfg.embed_k % 2 == 0 && h->cfg.embed_d % 2 == 0)
    return bprx_launch_score_gemm(h, u0, u1, out, s);          // fp32 MFMA GEMM (K step 2)
  return bprx_launch_score_block(h, u0, u1, out, s);
}

extern "C" int bprx_step_lr(const bprx_handle *h, float *lr_t) {
  if (!h || !lr_t) return BPRX_E_INVALID;
  *lr_t = h->pend_lr;
  return BPRX_OK;
}

extern "C" int bprx_index_pass_kind(const bprx_handle *h) { return h ? h->idx_kind : 0; }

extern "C" int bprx_sync_check(bprx_handle *h, void *stream) {
  if (!h) return BPRX_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  int32_t flag = 0;
  BPRX_HIP(h, hipMemcpyAsync(&flag, h->errflag, sizeof(flag), hipMemcpyDeviceToHost, s));
  BPRX_HIP(h, hipStreamSynchronize(s));
  if (flag) {
    BPRX_HIP(h, hipMemsetAsync(h->errflag, 0, sizeof(flag), s));
    BPRX_FAIL(h, BPRX_E_RANGE, "a user/item index was out of range (code %d); it was clamped, results are invalid", flag);
  }
  return BPRX_OK;
}
