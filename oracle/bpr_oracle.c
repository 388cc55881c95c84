/*
 * oracle/bpr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's BPRMF / VBPR training hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (fashionvisualexpl_recommend_amd + libbprx.so)
 * never links, imports or falls back to it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - index stream (orc_sample_ref_stream): PINNED against fixtures produced by
 *     running the reference's own DataLoader.all_triple_batches
 *     (src/dataset/dataset.py:83-114) in the build container
 *     (tests/golden/gen_golden.py).
 *   - metrics (orc_eval): PINNED against fixtures produced by the reference's own
 *     Evaluator (src/recommender/Evaluator.py:82-128,149-223).
 *   - train step / scores (orc_step, orc_score_pairs, orc_predict_all):
 *     "parity unpinned" at the TensorFlow boundary: tensorflow==2.3.1
 *     (requirements.txt:42) is not installable here and the reference holds no
 *     tests or golden vectors.  Restated from source text
 *     (BPRMF.py:55-125, VBPR.py:59-144) and cross-checked against torch-CPU
 *     autograd (tests/test_oracle_step.py), a neutral check, not the reference.
 *   - quant = 1 / 2 (bf16 / fp8 operand rounding): twins of the DEVICE's reduced-precision projections, not reference
 *     semantics (the reference is fp32 throughout).  The e4m3 rounding is cross-checked against torch's CPU
 *     float8_e4m3fn cast (identical on 2e5 random values inside the finite range; saturation instead of NaN beyond).
 *   - device samplers' twins (orc_sample_philox, orc_sample_epoch): definitions of this build, no reference counterpart.
 *
 * Conventions: all parameters are fp32 row-major exactly like the reference's
 * tf.Variables; long sums are accumulated in double and rounded once to fp32 so
 * that the oracle is independent of summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* MT19937 (the generator behind both Python's `random` and NumPy's legacy     */
/* RandomState, which dataset.py:84,95 interleave).                            */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint32_t mt[624];
  int mti;
} orc_mt;

static void mt_init_genrand(orc_mt *s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; i++)
    s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->mti = 624;
}

static void mt_init_by_array(orc_mt *s, const uint32_t *key, int klen) {
  mt_init_genrand(s, 19650218u);
  int i = 1, j = 0;
  int k = 624 > klen ? 624 : klen;
  for (; k; k--) {
    s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    i++; j++;
    if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (k = 623; k; k--) {
    s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    i++;
    if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
  }
  s->mt[0] = 0x80000000u;
}

static uint32_t mt_next(orc_mt *s) {
  if (s->mti >= 624) {
    uint32_t *mt = s->mt;
    for (int kk = 0; kk < 624; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % 624] & 0x7fffffffu);
      mt[kk] = mt[(kk + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->mti = 0;
  }
  uint32_t y = s->mt[s->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* Python: random.seed(int) -> init_by_array over the 32-bit digits of |seed| (one digit 0 for seed 0). */
static void py_random_seed(orc_mt *s, uint32_t seed) {
  uint32_t key[1] = {seed};
  mt_init_by_array(s, key, 1);
}

static int bit_length_u32(uint32_t n) {
  int b = 0;
  while (n) { b++; n >>= 1; }
  return b;
}

/* Python: Random._randbelow_with_getrandbits(n), n >= 1, n < 2^32. */
static uint32_t py_randbelow(orc_mt *s, uint32_t n) {
  int k = bit_length_u32(n);
  uint32_t r = mt_next(s) >> (32 - k);
  while (r >= n) r = mt_next(s) >> (32 - k);
  return r;
}

/* Python: random.shuffle(x) (3.x, no `random` argument): Fisher-Yates from the top. */
static void py_shuffle_i32(orc_mt *s, int32_t *x, int32_t n) {
  for (int32_t i = n - 1; i >= 1; i--) {
    uint32_t j = py_randbelow(s, (uint32_t)i + 1u);
    int32_t t = x[i]; x[i] = x[j]; x[j] = t;
  }
}

/* NumPy legacy: np.random.randint(n) (int64 dtype, masked rejection on 32-bit draws, n-1 <= 0xFFFFFFFF). */
static uint32_t np_randint(orc_mt *s, uint32_t n) {
  uint32_t rng = n - 1u;
  if (rng == 0) return 0; /* consumes no draw */
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  while ((v = (mt_next(s) & mask)) > rng) {}
  return v;
}

/* Exposed for neutral cross-checks against CPython / NumPy themselves. */
void orc_py_shuffle(uint32_t seed, int32_t *x, int32_t n) {
  orc_mt s; py_random_seed(&s, seed); py_shuffle_i32(&s, x, n);
}
void orc_np_randint(uint32_t seed, uint32_t n, int64_t count, int64_t *out) {
  orc_mt s; mt_init_genrand(&s, seed);
  for (int64_t c = 0; c < count; c++) out[c] = np_randint(&s, n);
}

/* ------------------------------------------------------------------------- */
/* A2: DataLoader.all_triple_batches (dataset.py:83-114).                      */
/* training_list is CSR: items of user u are items[indptr[u] .. indptr[u+1]).  */
/* Returns the number of triplets written (<= cap); -1 if cap is too small.    */
/* ------------------------------------------------------------------------- */
int64_t orc_sample_count(const int64_t *indptr, int32_t U, int32_t bs, int32_t epochs) {
  int64_t N = indptr[U];
  int64_t actual = (N / bs) * bs * (int64_t)epochs; /* dataset.py:89-91 */
  /* counter starts at 1 and the early return fires when counter == actual (dataset.py:87,109):
     with actual == 0 it never fires and every epoch is emitted in full. */
  return actual > 0 ? actual : N * (int64_t)epochs;
}

int64_t orc_sample_ref_stream(const int64_t *indptr, const int32_t *items, int32_t U, int32_t I,
                              int32_t bs, int32_t epochs, uint32_t py_seed, uint32_t np_seed,
                              int32_t *out_u, int32_t *out_i, int32_t *out_j, int64_t cap) {
  orc_mt pyr, npr;
  py_random_seed(&pyr, py_seed);     /* BPRMF.py:15  random.seed(0)    */
  mt_init_genrand(&npr, np_seed);    /* BPRMF.py:16  np.random.seed(0) */
  int64_t N = indptr[U];
  int64_t actual = (N / bs) * bs * (int64_t)epochs;
  int64_t counter = 1, n = 0;
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(U > 0 ? U : 1));
  for (int32_t ep = 0; ep < epochs; ep++) {
    for (int32_t a = 0; a < U; a++) order[a] = a;       /* dataset.py:94 */
    py_shuffle_i32(&pyr, order, U);                      /* dataset.py:95 */
    for (int32_t a = 0; a < U; a++) {
      int32_t u = order[a];
      const int32_t *uis = items + indptr[u];
      int64_t len = indptr[u + 1] - indptr[u];
      for (int64_t p = 0; p < len; p++) {
        uint32_t j;
        for (;;) {                                       /* dataset.py:101-103 */
          j = np_randint(&npr, (uint32_t)I);
          int found = 0;
          for (int64_t q = 0; q < len; q++) if ((uint32_t)uis[q] == j) { found = 1; break; }
          if (!found) break;
        }
        if (n >= cap) { free(order); return -1; }
        out_u[n] = u; out_i[n] = uis[p]; out_j[n] = (int32_t)j; n++;
        if (counter == actual) { free(order); return n; } /* dataset.py:109-110 */
        counter++;
      }
    }
  }
  free(order);
  return n;
}

/* ------------------------------------------------------------------------- */
/* Model state. BPRMF: d == 0, D == 0, Tu/F/E/Bp NULL.                         */
/* ------------------------------------------------------------------------- */
typedef struct {
  int32_t U, I, k, d, D;
  float *Gu;       /* [U,k]  BPRMF.py:49 */
  float *Gi;       /* [I,k]  BPRMF.py:50 */
  float *Bi;       /* [I]    BPRMF.py:48 */
  float *Tu;       /* [U,d]  VBPR.py:46  */
  const float *F;  /* [I,D]  VBPR.py:49 (frozen) */
  float *E;        /* [D,d]  VBPR.py:52  */
  float *Bp;       /* [D]    VBPR.py:44 ([D,1]) */
  /* Adam slots (adam_tf23), same shapes; may be NULL for sgd */
  float *mGu, *vGu, *mGi, *vGi, *mBi, *vBi, *mTu, *vTu, *mE, *vE, *mBp, *vBp;
  int64_t adam_t;  /* optimizer.iterations */
  /* operand rounding that mirrors the device's reduced-precision projection:
     0 = none (reference fp32 semantics)
     1 = bf16: E|Bp rounded to bf16 in the forward product, W=(g*[theta|1]) rounded to bf16 in dE
     2 = fp8:  E|Bp rounded to OCP e4m3fn after scaling by qscale = 448 / max|E,Bp| (set by the caller before every
               call; F is expected to hold already-dequantised fp8 values), W rounded to bf16 in dE */
  int32_t quant;
  float qscale;
} orc_model;

static float bf16_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  float r; memcpy(&r, &u, 4); return r;
}
float orc_bf16_round(float x) { return bf16_round(x); }

/* value of the OCP e4m3fn code nearest to x (round-to-nearest-even, saturating at +-448); x finite */
static float e4m3_round(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  uint32_t a = u & 0x7fffffffu;
  float ax; memcpy(&ax, &a, 4);
  float r;
  if (ax < 0.015625f) {                     /* below 2^-6: multiples of 2^-9 */
    r = rintf(ax * 512.0f) * (1.0f / 512.0f);
  } else {
    uint32_t rr = a + 0x7ffffu + ((a >> 20) & 1u);   /* keep 3 mantissa bits, RNE */
    int e = (int)(rr >> 23) - 127;
    uint32_t mm = (rr >> 20) & 7u;
    if (e > 8 || (e == 8 && mm == 7u)) r = 448.0f;
    else { rr &= 0xfff00000u; memcpy(&r, &rr, 4); }
  }
  return (u >> 31) ? -r : r;
}
float orc_e4m3_round(float x) { return e4m3_round(x); }
void orc_e4m3_round_array(const float *in, float *out, int64_t n) {
  for (int64_t e = 0; e < n; e++) out[e] = e4m3_round(in[e]);
}

/* operand rounding of E|Bp in the forward projection, and of W in the dense gradient */
static inline float qz(const orc_model *m, float x) {
  if (m->quant == 1) return bf16_round(x);
  if (m->quant == 2) return e4m3_round(x * m->qscale) / m->qscale;
  return x;
}
static inline float qw(const orc_model *m, float x) { return m->quant ? bf16_round(x) : x; }

/* P[0..d-1] = f_item . E[:,c],  P[d] = f_item . Bp      (VBPR.py:83-84) */
static void project_item(const orc_model *m, int32_t item, float *P) {
  const int d = m->d, D = m->D;
  const float *f = m->F + (size_t)item * D;
  double acc[d + 1];
  for (int c = 0; c <= d; c++) acc[c] = 0.0;
  for (int r = 0; r < D; r++) {
    double fr = f[r];
    if (fr == 0.0) continue;
    const float *e = m->E + (size_t)r * d;
    for (int c = 0; c < d; c++) acc[c] += fr * (double)qz(m, e[c]);
    acc[d] += fr * (double)qz(m, m->Bp[r]);
  }
  for (int c = 0; c <= d; c++) P[c] = (float)acc[c];
}

/* x_ui for one pair given the item's projection row (NULL for BPRMF).  BPRMF.py:74 / VBPR.py:82-84 */
static float score_one(const orc_model *m, int32_t u, int32_t i, const float *P) {
  const int k = m->k, d = m->d;
  double s = 0.0;
  const float *gu = m->Gu + (size_t)u * k, *gi = m->Gi + (size_t)i * k;
  for (int c = 0; c < k; c++) s += (double)gu[c] * (double)gi[c];
  float x = m->Bi[i] + (float)s;
  if (d > 0) {
    const float *tu = m->Tu + (size_t)u * d;
    double t = 0.0;
    for (int c = 0; c < d; c++) t += (double)tu[c] * (double)P[c];
    x = x + (float)t + P[d];
  }
  return x;
}

/* A5/A8: Model.call((user,item)) -> x_ui  (BPRMF.py:55-76, VBPR.py:59-86) */
void orc_score_pairs(const orc_model *m, const int32_t *u, const int32_t *i, int64_t B, float *x) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    float P[m->d + 1];
    if (m->d > 0) project_item(m, i[b], P);
    x[b] = score_one(m, u[b], i[b], m->d > 0 ? P : NULL);
  }
}

/* A11: predict_all -> [U,I] fp32 (BPRMF.py:85, VBPR.py:95-97) */
void orc_predict_all(const orc_model *m, float *out) {
  const int d = m->d;
  float *P = NULL;
  if (d > 0) {
    P = (float *)malloc(sizeof(float) * (size_t)m->I * (d + 1));
#pragma omp parallel for schedule(static)
    for (int32_t it = 0; it < m->I; it++) project_item(m, it, P + (size_t)it * (d + 1));
  }
#pragma omp parallel for schedule(static)
  for (int32_t u = 0; u < m->U; u++)
    for (int32_t it = 0; it < m->I; it++)
      out[(size_t)u * m->I + it] = score_one(m, u, it, d > 0 ? P + (size_t)it * (d + 1) : NULL);
  free(P);
}

/* TF-2.3 Keras Adam, sparse (IndexedSlices) path, NON-lazy: every row decays and moves every step.
   m_t = m*b1 + g*(1-b1) ; v_t = v*b2 + g*g*(1-b2) ; var -= lr_t * m_t / (sqrt(v_t) + eps)
   (optimizer_v2/adam.py::_resource_apply_sparse; call sites BPRMF.py:123, VBPR.py:142) */
static void adam_sparse_table(float *p, float *mm, float *vv, const double *g, size_t n,
                              float b1, float b2, float lr_t, float eps) {
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#pragma omp parallel for schedule(static)
  for (size_t e = 0; e < n; e++) {
    float gg = (float)g[e];
    float mt = mm[e] * b1 + gg * omb1;
    float vt = vv[e] * b2 + (gg * gg) * omb2;
    mm[e] = mt; vv[e] = vt;
    p[e] = p[e] - lr_t * mt / (sqrtf(vt) + eps);
  }
}
/* dense path (training_ops ApplyAdam): m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= lr_t*m/(sqrt(v)+eps) */
static void adam_dense_table(float *p, float *mm, float *vv, const double *g, size_t n,
                             float b1, float b2, float lr_t, float eps) {
  const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
  for (size_t e = 0; e < n; e++) {
    float gg = (float)g[e];
    float mt = mm[e] + (gg - mm[e]) * omb1;
    float vt = vv[e] + (gg * gg - vv[e]) * omb2;
    mm[e] = mt; vv[e] = vt;
    p[e] = p[e] - lr_t * mt / (sqrtf(vt) + eps);
  }
}
static void sgd_table(float *p, const double *g, size_t n, float lr) {
  for (size_t e = 0; e < n; e++) p[e] = p[e] - lr * (float)g[e];
}

/* Optional taps for tests: per-triplet x+, x-, g (any may be NULL). */
typedef struct { float *xp, *xn, *g; double *dE; double *dBp; } orc_taps;

/* A6/A9: train_step (BPRMF.py:87-125, VBPR.py:99-144). optimizer: 0 = sgd, 1 = adam_tf23.
   Batch-synchronous: every gradient from pre-update values, duplicate rows summed, one update.
   Returns the scalar loss (data term + regularisation) as the reference's loss.numpy(). */
double orc_step(orc_model *m, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B,
                int optimizer, float lr, float reg, const orc_taps *taps) {
  const int k = m->k, d = m->d, D = m->D, d1 = d + 1;
  const int32_t U = m->U, I = m->I;
  double *dGu = (double *)calloc((size_t)U * k, sizeof(double));
  double *dGi = (double *)calloc((size_t)I * k, sizeof(double));
  double *dBi = (double *)calloc((size_t)I, sizeof(double));
  double *dTu = NULL, *W = NULL, *dE = NULL, *dBp = NULL;
  float *P = NULL; int32_t *slot = NULL, *uniq = NULL; int32_t nT = 0;
  if (d > 0) {
    dTu = (double *)calloc((size_t)U * d, sizeof(double));
    slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    uniq = (int32_t *)malloc(sizeof(int32_t) * (size_t)I);
    for (int32_t t = 0; t < I; t++) slot[t] = -1;
    for (int64_t b = 0; b < B; b++) { slot[i[b]] = 0; slot[j[b]] = 0; }
    for (int32_t t = 0; t < I; t++) if (slot[t] == 0) { slot[t] = nT; uniq[nT++] = t; }
    P = (float *)malloc(sizeof(float) * (size_t)nT * d1);
    W = (double *)calloc((size_t)nT * d1, sizeof(double));
#pragma omp parallel for schedule(dynamic, 16)
    for (int32_t t = 0; t < nT; t++) project_item(m, uniq[t], P + (size_t)t * d1);
  }
  double loss = 0.0, regsum = 0.0;
  for (int64_t b = 0; b < B; b++) {
    const int32_t uu = u[b], ii = i[b], jj = j[b];
    const float *Pi = d > 0 ? P + (size_t)slot[ii] * d1 : NULL;
    const float *Pj = d > 0 ? P + (size_t)slot[jj] * d1 : NULL;
    float xp = score_one(m, uu, ii, Pi), xn = score_one(m, uu, jj, Pj);
    float diff = xp - xn;
    int inr = (diff >= -80.0f) && (diff <= 1e8f);
    float cl = diff < -80.0f ? -80.0f : (diff > 1e8f ? 1e8f : diff);           /* BPRMF.py:104 */
    double z = -(double)cl;
    loss += z > 0 ? z + log1p(exp(-z)) : log1p(exp(z));                        /* softplus, BPRMF.py:105 */
    double g = inr ? -1.0 / (1.0 + exp((double)diff)) : 0.0;                   /* -sigmoid(-diff) */
    if (taps) { if (taps->xp) taps->xp[b] = xp; if (taps->xn) taps->xn[b] = xn; if (taps->g) taps->g[b] = (float)g; }
    const float *gu = m->Gu + (size_t)uu * k, *gi = m->Gi + (size_t)ii * k, *gj = m->Gi + (size_t)jj * k;
    double *au = dGu + (size_t)uu * k, *ai = dGi + (size_t)ii * k, *aj = dGi + (size_t)jj * k;
    for (int c = 0; c < k; c++) {
      au[c] += g * ((double)gi[c] - (double)gj[c]) + 2.0 * reg * gu[c];
      ai[c] += g * gu[c] + 2.0 * reg * gi[c];
      aj[c] += -g * gu[c] + 2.0 * reg * gj[c];
      regsum += (double)gu[c] * gu[c] + (double)gi[c] * gi[c] + (double)gj[c] * gj[c]; /* BPRMF.py:108-110 */
    }
    dBi[ii] += g + 2.0 * reg * m->Bi[ii];
    dBi[jj] += -g + 2.0 * (reg / 10.0) * m->Bi[jj];
    regsum += (double)m->Bi[ii] * m->Bi[ii] + (double)m->Bi[jj] * m->Bi[jj] / 10.0;      /* BPRMF.py:111-112 */
    if (d > 0) {
      const float *tu = m->Tu + (size_t)uu * d;
      double *at = dTu + (size_t)uu * d;
      double *wi = W + (size_t)slot[ii] * d1, *wj = W + (size_t)slot[jj] * d1;
      for (int c = 0; c < d; c++) {
        at[c] += g * ((double)Pi[c] - (double)Pj[c]) + 2.0 * reg * tu[c];
        wi[c] += g * tu[c]; wj[c] -= g * tu[c];
        regsum += (double)tu[c] * tu[c];                                               /* VBPR.py:124 */
      }
      wi[d] += g; wj[d] -= g;
    }
  }
  if (d > 0) {
    /* dE = sum_t F[t]^T W[t] + 2 reg E ; dBp likewise (VBPR.py:127,141) */
    dE = (double *)calloc((size_t)D * d, sizeof(double));
    dBp = (double *)calloc((size_t)D, sizeof(double));
    float *Wq = (float *)malloc(sizeof(float) * (size_t)nT * d1);
    for (size_t e = 0; e < (size_t)nT * d1; e++) Wq[e] = qw(m, (float)W[e]);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < D; r++) {
      double acc[d1];
      for (int c = 0; c < d1; c++) acc[c] = 0.0;
      for (int32_t t = 0; t < nT; t++) {
        double fr = m->F[(size_t)uniq[t] * D + r];
        if (fr == 0.0) continue;
        const float *w = Wq + (size_t)t * d1;
        for (int c = 0; c < d1; c++) acc[c] += fr * (double)w[c];
      }
      for (int c = 0; c < d; c++) dE[(size_t)r * d + c] = acc[c] + 2.0 * reg * m->E[(size_t)r * d + c];
      dBp[r] = acc[d] + 2.0 * reg * m->Bp[r];
    }
    free(Wq);
    double er = 0.0;
    for (size_t e = 0; e < (size_t)D * d; e++) er += (double)m->E[e] * m->E[e];
    for (int r = 0; r < D; r++) er += (double)m->Bp[r] * m->Bp[r];
    regsum += er;                                                                        /* VBPR.py:127 */
    if (taps && taps->dE) memcpy(taps->dE, dE, sizeof(double) * (size_t)D * d);
    if (taps && taps->dBp) memcpy(taps->dBp, dBp, sizeof(double) * (size_t)D);
  }
  loss += (double)reg * regsum;

  if (optimizer == 0) {
    sgd_table(m->Gu, dGu, (size_t)U * k, lr);
    sgd_table(m->Gi, dGi, (size_t)I * k, lr);
    sgd_table(m->Bi, dBi, (size_t)I, lr);
    if (d > 0) {
      sgd_table(m->Tu, dTu, (size_t)U * d, lr);
      sgd_table(m->E, dE, (size_t)D * d, lr);
      sgd_table(m->Bp, dBp, (size_t)D, lr);
    }
  } else {
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-7f;
    m->adam_t += 1;
    float t = (float)m->adam_t;
    float lr_t = lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));                  /* adam.py _prepare_local */
    adam_sparse_table(m->Bi, m->mBi, m->vBi, dBi, (size_t)I, b1, b2, lr_t, eps);
    adam_sparse_table(m->Gu, m->mGu, m->vGu, dGu, (size_t)U * k, b1, b2, lr_t, eps);
    adam_sparse_table(m->Gi, m->mGi, m->vGi, dGi, (size_t)I * k, b1, b2, lr_t, eps);
    if (d > 0) {
      adam_sparse_table(m->Tu, m->mTu, m->vTu, dTu, (size_t)U * d, b1, b2, lr_t, eps);
      adam_dense_table(m->E, m->mE, m->vE, dE, (size_t)D * d, b1, b2, lr_t, eps);
      adam_dense_table(m->Bp, m->mBp, m->vBp, dBp, (size_t)D, b1, b2, lr_t, eps);
    }
  }
  free(dGu); free(dGi); free(dBi); free(dTu); free(W); free(dE); free(dBp); free(P); free(slot); free(uniq);
  return loss;
}

/* ------------------------------------------------------------------------- */
/* A13: Evaluator._eval_by_user + Evaluator.eval means (Evaluator.py:82-128,   */
/* 181-193).  scores = predict_all() [U,I]; lists are CSR.                     */
/* out10 = hr_v, p_v, r_v, auc_v, ndcg_v, hr_t, p_t, r_t, auc_t, ndcg_t        */
/* (true auc_t; the 'auc_t': auc_v aliasing of Evaluator.py:220 is applied by  */
/* the Python mirror when it builds the results dict).                         */
/* ------------------------------------------------------------------------- */
static int eval_user(const float *s, int32_t I, const int32_t *train, int64_t ntrain,
                     const int32_t *ev, int64_t nev, int K, uint8_t *mark, double out[5]) {
  if (nev <= 0) return 0;                                                   /* Evaluator.py:88-89 */
  /* candidates = all items - train - eval, eval appended last (Evaluator.py:36-53) */
  memset(mark, 0, (size_t)I);
  for (int64_t q = 0; q < ntrain; q++) mark[train[q]] = 1;
  for (int64_t q = 0; q < nev; q++) mark[ev[q]] = 2;
  int64_t nneg = 0;
  for (int32_t it = 0; it < I; it++) if (!mark[it]) nneg++;
  int64_t position = 0;                                                      /* Evaluator.py:96-98 */
  for (int64_t t = 0; t < nev; t++) {
    float sp = s[ev[t]];
    for (int32_t it = 0; it < I; it++) if (!mark[it] && s[it] >= sp) position++;
  }
  double auc = 1.0 - (double)position / ((double)nneg * (double)nev);        /* Evaluator.py:100 */
  /* top-K by heapq.nlargest == stable descending sort: ties keep iteration order
     (negatives ascending by id, then the eval items)  (Evaluator.py:104-115) */
  int64_t ncand = nneg + nev;
  int64_t topn = ncand < K ? ncand : K;
  int hits = 0;
  for (int64_t t = 0; t < nev; t++) {
    float sp = s[ev[t]];
    int64_t rank = 0;
    for (int32_t it = 0; it < I; it++) if (!mark[it] && s[it] >= sp) rank++;   /* negatives come first on ties */
    for (int64_t q = 0; q < nev; q++) {
      if (q == t) continue;
      float sq = s[ev[q]];
      if (sq > sp || (sq == sp && q < t)) rank++;
    }
    if (rank < topn) hits++;
  }
  out[0] = hits > 0 ? 1.0 : 0.0;                                             /* hr   :117 */
  out[1] = topn > 0 ? (double)hits / (double)topn : 0.0;                     /* prec :123 */
  out[2] = (double)hits / (double)nev;                                       /* rec  :126 */
  out[3] = auc;
  out[4] = position < K ? log(2.0) / log((double)position + 2.0) : 0.0;      /* ndcg :120 */
  return 1;
}

void orc_eval(const float *scores, int32_t U, int32_t I,
              const int64_t *tr_ptr, const int32_t *tr_items,
              const int64_t *va_ptr, const int32_t *va_items,   /* may be NULL */
              const int64_t *te_ptr, const int32_t *te_items, int K, double *out10) {
  double sv[5] = {0, 0, 0, 0, 0}, st[5] = {0, 0, 0, 0, 0};
  int64_t nv = 0, nt = 0;
  uint8_t *mark = (uint8_t *)malloc((size_t)I);
  for (int32_t u = 0; u < U; u++) {
    double r[5];
    const float *s = scores + (size_t)u * I;
    if (eval_user(s, I, tr_items + tr_ptr[u], tr_ptr[u + 1] - tr_ptr[u],
                  te_items + te_ptr[u], te_ptr[u + 1] - te_ptr[u], K, mark, r)) {
      for (int c = 0; c < 5; c++) st[c] += r[c];
      nt++;
    }
    if (va_ptr && eval_user(s, I, tr_items + tr_ptr[u], tr_ptr[u + 1] - tr_ptr[u],
                            va_items + va_ptr[u], va_ptr[u + 1] - va_ptr[u], K, mark, r)) {
      for (int c = 0; c < 5; c++) sv[c] += r[c];
      nv++;
    }
  }
  free(mark);
  for (int c = 0; c < 5; c++) {
    out10[c] = nv ? sv[c] / (double)nv : 0.0;
    out10[5 + c] = nt ? st[c] / (double)nt : 0.0;
  }
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* Throughput sampler twin (NOT in the reference: SURVEY 8(f) N2).             */
/* Counter-based Philox4x32-10: triplet n of stream `seed` is a pure function  */
/* of (seed, n), so the device kernel and this loop produce the same triplets. */
/*   block(n, a) = philox(key = seed, ctr = (n_lo, n_hi, a, 0))                */
/*   positive p  = mulhi64(block(n,0).xy, N)          -> (user_of[p], item_of[p]) */
/*   negative j  = mulhi32(block(n,a).z, I), a = 0,1,.. until j not in train(u) */
/*   (train lists sorted ascending per user; at most 1024 attempts).             */
/* ------------------------------------------------------------------------- */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the user order of an epoch (twin of k_epoch_prepare / epoch_perm_at in bprx_philox.hip): a 4-round Feistel network over
   2*half bits, half = ceil(bits(U-1)/2), round function philox(key = seed, ctr = (R, r, 0xFFFFFFFE, epoch))[0] & mask,
   cycle-walked into [0, U) */
void orc_epoch_perm(uint64_t seed, uint32_t epoch, int32_t U, int32_t *perm) {
  int nb = 1;
  while (nb < 32 && ((uint32_t)U - 1u) >> nb) nb++;
  const int half = (nb + 1) / 2;
  const uint32_t mask = (1u << half) - 1u;
  for (int32_t a = 0; a < U; a++) {
    uint32_t x = (uint32_t)a;
    do {
      uint32_t L = x >> half, R = x & mask;
      for (uint32_t r = 0; r < 4; r++) {
        uint32_t o[4];
        philox4x32_10(R, r, 0xFFFFFFFEu, epoch, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        uint32_t t = L ^ (o[0] & mask);
        L = R; R = t;
      }
      x = (L << half) | R;
    } while (x >= (uint32_t)U);
    perm[a] = (int32_t)x;
  }
}

void orc_sample_philox(const int64_t *indptr, const int32_t *items_sorted, const int32_t *pos_user, int64_t N,
                       int32_t I, uint64_t seed, uint64_t first, int64_t B, int32_t *u, int32_t *i, int32_t *j) {
  for (int64_t b = 0; b < B; b++) {
    uint64_t n = first + (uint64_t)b;
    uint32_t r[4];
    philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    uint64_t x = ((uint64_t)r[1] << 32) | r[0];
    uint64_t p = (uint64_t)(((unsigned __int128)x * (unsigned __int128)(uint64_t)N) >> 64);
    int32_t uu = pos_user[p];
    const int32_t *lst = items_sorted + indptr[uu];
    int64_t len = indptr[uu + 1] - indptr[uu];
    int32_t jj = 0;
    for (uint32_t a = 0; a < 1024; a++) {
      if (a) philox4x32_10((uint32_t)n, (uint32_t)(n >> 32), a, 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
      jj = (int32_t)(((uint64_t)r[2] * (uint64_t)(uint32_t)I) >> 32);
      int64_t lo = 0, hi = len;                     /* binary search in the sorted positives */
      while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (lst[mid] < jj) lo = mid + 1; else hi = mid; }
      if (!(lo < len && lst[lo] == jj)) break;
    }
    u[b] = uu; i[b] = items_sorted[p]; j[b] = jj;
  }
}

/* CPU twin of the epoch-walk device sampler (bprx_sample_epoch): position n of epoch `epoch` -> user perm[a] with
   epoch_ptr[a] <= n < epoch_ptr[a+1], its positive number n - epoch_ptr[a]; negative = Philox rejection keyed by
   (seed; n, epoch). */
void orc_sample_epoch(const int64_t *indptr, const int32_t *items_sorted, const int32_t *perm, const int64_t *epoch_ptr,
                      int32_t U, int32_t I, uint64_t seed, uint32_t epoch, int64_t first, int64_t B,
                      int32_t *u, int32_t *i, int32_t *j) {
  for (int64_t b = 0; b < B; b++) {
    int64_t n = first + b;
    int32_t lo = 0, hi = U;
    while (hi - lo > 1) { int32_t mid = (lo + hi) >> 1; if (epoch_ptr[mid] <= n) lo = mid; else hi = mid; }
    int32_t uu = perm[lo];
    const int32_t *lst = items_sorted + indptr[uu];
    int64_t len = indptr[uu + 1] - indptr[uu];
    uint32_t r[4];
    int32_t jj = 0;
    for (uint32_t a = 0; a < 1024; a++) {
      philox4x32_10((uint32_t)n, (uint32_t)((uint64_t)n >> 32), a, epoch, (uint32_t)seed, (uint32_t)(seed >> 32), r);
      jj = (int32_t)(((uint64_t)r[2] * (uint64_t)(uint32_t)I) >> 32);
      int64_t l = 0, h = len;
      while (l < h) { int64_t mid = (l + h) >> 1; if (lst[mid] < jj) l = mid + 1; else h = mid; }
      if (!(l < len && lst[l] == jj)) break;
    }
    u[b] = uu; i[b] = lst[n - epoch_ptr[lo]]; j[b] = jj;
  }
}
