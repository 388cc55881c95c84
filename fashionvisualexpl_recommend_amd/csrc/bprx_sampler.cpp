// bprx_sampler.cpp -- host-side reference-compatible index stream (C ABI: bprx_sampler_*).
//
// Replaces DataLoader.all_triple_batches (src/dataset/dataset.py:83-114).  The stream is defined by two
// interleaved MT19937 front-ends and a data-dependent rejection loop, so it is inherently sequential and is
// produced on the host (12 B per triplet), then uploaded; the throughput sampler is the device Philox one.
//   * users:     random.shuffle(list(range(U)))  -- CPython: seed(int) = init_by_array, Fisher-Yates from
//                the top with _randbelow = getrandbits(bit_length(n)) rejection       (dataset.py:94-95)
//   * negatives: np.random.randint(I)           -- NumPy legacy RandomState: seed(int) = init_genrand,
//                32-bit draw & mask, reject > I-1; j re-drawn while j in training_list[u] (dataset.py:101-103)
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

#include "bprx.h"

namespace {

class Mt19937 {
 public:
  void seed_genrand(uint32_t s) {
    mt_[0] = s;
    for (int i = 1; i < N; ++i) mt_[i] = 1812433253u * (mt_[i - 1] ^ (mt_[i - 1] >> 30)) + static_cast<uint32_t>(i);
    idx_ = N;
  }
  void seed_by_array(const std::vector<uint32_t> &key) {
    seed_genrand(19650218u);
    const int klen = static_cast<int>(key.size());
    int i = 1, j = 0;
    for (int n = (N > klen ? N : klen); n > 0; --n) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1664525u)) + key[j] + static_cast<uint32_t>(j);
      if (++i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
      if (++j >= klen) j = 0;
    }
    for (int n = N - 1; n > 0; --n) {
      mt_[i] = (mt_[i] ^ ((mt_[i - 1] ^ (mt_[i - 1] >> 30)) * 1566083941u)) - static_cast<uint32_t>(i);
      if (++i >= N) { mt_[0] = mt_[N - 1]; i = 1; }
    }
    mt_[0] = 0x80000000u;
  }
  uint32_t next() {
    if (idx_ >= N) refill();
    uint32_t y = mt_[idx_++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }

 private:
  static constexpr int N = 624, M = 397;
  void refill() {
    for (int k = 0; k < N; ++k) {
      const uint32_t y = (mt_[k] & 0x80000000u) | (mt_[(k + 1) % N] & 0x7fffffffu);
      mt_[k] = mt_[(k + M) % N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    idx_ = 0;
  }
  uint32_t mt_[N];
  int idx_ = N;
};

inline int bit_length(uint32_t n) { return n ? 32 - __builtin_clz(n) : 0; }

// CPython Random._randbelow_with_getrandbits
inline uint32_t randbelow(Mt19937 &g, uint32_t n) {
  const int shift = 32 - bit_length(n);
  uint32_t r = g.next() >> shift;
  while (r >= n) r = g.next() >> shift;
  return r;
}

// NumPy legacy RandomState.randint(high) for high-1 <= 0xFFFFFFFF (masked rejection on 32-bit draws)
inline uint32_t legacy_randint(Mt19937 &g, uint32_t high) {
  const uint32_t rng = high - 1u;
  if (rng == 0) return 0;
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  for (;;) {
    const uint32_t v = g.next() & mask;
    if (v <= rng) return v;
  }
}

}  // namespace

struct bprx_sampler {
  std::vector<int64_t> indptr;
  std::vector<int32_t> items;
  int32_t U, I;
};

extern "C" int bprx_sampler_create(const int64_t *indptr, const int32_t *items, int32_t num_users, int32_t num_items,
                                   bprx_sampler **out) {
  if (!indptr || !out || num_users < 0 || num_items <= 0) return BPRX_E_INVALID;
  if (indptr[0] != 0) return BPRX_E_INVALID;
  for (int32_t u = 0; u < num_users; ++u)
    if (indptr[u + 1] < indptr[u]) return BPRX_E_INVALID;
  const int64_t n = indptr[num_users];
  if (n > 0 && !items) return BPRX_E_INVALID;
  for (int64_t p = 0; p < n; ++p)
    if (items[p] < 0 || items[p] >= num_items) return BPRX_E_RANGE;
  bprx_sampler *s = new (std::nothrow) bprx_sampler();
  if (!s) return BPRX_E_NOMEM;
  s->indptr.assign(indptr, indptr + num_users + 1);
  s->items.assign(items, items + n);
  s->U = num_users;
  s->I = num_items;
  *out = s;
  return BPRX_OK;
}

extern "C" int bprx_sampler_destroy(bprx_sampler *s) {
  delete s;
  return BPRX_OK;
}

extern "C" int64_t bprx_sampler_count(const bprx_sampler *s, int32_t batch_size, int32_t epochs) {
  if (!s || batch_size <= 0 || epochs < 0) return BPRX_E_INVALID;
  const int64_t n = s->indptr[s->U];
  const int64_t actual = (n / batch_size) * batch_size * static_cast<int64_t>(epochs);   // dataset.py:89-91
  return actual > 0 ? actual : n * static_cast<int64_t>(epochs);  // the ==actual early return never fires at 0 (:109)
}

extern "C" int64_t bprx_sampler_ref_stream(bprx_sampler *s, int32_t batch_size, int32_t epochs, uint32_t py_seed,
                                           uint32_t np_seed, int32_t *user, int32_t *pos, int32_t *neg, int64_t cap) {
  if (!s || batch_size <= 0 || epochs < 0 || !user || !pos || !neg) return BPRX_E_INVALID;
  const int64_t want = bprx_sampler_count(s, batch_size, epochs);
  if (cap < want) return BPRX_E_INVALID;
  // a user whose every item is a positive can never draw a negative: the reference would spin forever
  for (int32_t u = 0; u < s->U; ++u)
    if (s->indptr[u + 1] - s->indptr[u] >= s->I) {
      std::vector<char> seen(s->I, 0);
      int64_t distinct = 0;
      for (int64_t p = s->indptr[u]; p < s->indptr[u + 1]; ++p)
        if (!seen[s->items[p]]) { seen[s->items[p]] = 1; ++distinct; }
      if (distinct >= s->I) return BPRX_E_INVALID;
    }
  Mt19937 py, np;
  py.seed_by_array(std::vector<uint32_t>{py_seed});   // random.seed(0)     BPRMF.py:15
  np.seed_genrand(np_seed);                           // np.random.seed(0)  BPRMF.py:16
  const int64_t n_pos = s->indptr[s->U];
  const int64_t actual = (n_pos / batch_size) * batch_size * static_cast<int64_t>(epochs);
  std::vector<int32_t> order(static_cast<size_t>(s->U));
  int64_t n = 0, counter = 1;
  for (int32_t ep = 0; ep < epochs; ++ep) {
    for (int32_t a = 0; a < s->U; ++a) order[a] = a;
    for (int32_t a = s->U - 1; a >= 1; --a) {          // random.shuffle
      const uint32_t b = randbelow(py, static_cast<uint32_t>(a) + 1u);
      const int32_t t = order[a]; order[a] = order[b]; order[b] = t;
    }
    for (int32_t a = 0; a < s->U; ++a) {
      const int32_t u = order[a];
      const int32_t *lst = s->items.data() + s->indptr[u];
      const int64_t len = s->indptr[u + 1] - s->indptr[u];
      for (int64_t p = 0; p < len; ++p) {
        int32_t j;
        bool clash;
        do {
          j = static_cast<int32_t>(legacy_randint(np, static_cast<uint32_t>(s->I)));
          clash = false;
          for (int64_t q = 0; q < len; ++q)
            if (lst[q] == j) { clash = true; break; }
        } while (clash);
        user[n] = u; pos[n] = lst[p]; neg[n] = j; ++n;
        if (counter == actual) return n;                 // dataset.py:109-110
        ++counter;
      }
    }
  }
  return n;
}
