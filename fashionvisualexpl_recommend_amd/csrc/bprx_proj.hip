// bprx_proj.hip -- gfx950 kernels for VBPR's visual projection, item-centric:
//   forward   P[t, 0:d] = f_t . E ,  P[t, d] = f_t . Bp          (VBPR.py:83-84, one row per item, not per triplet)
//   backward  dE = sum_t f_t^T W[t, 0:d] , dBp = sum_t f_t W[t,d] (the dense gradients of VBPR.py:141)
// with W[t] = sum over the batch of +-g_b*[theta_u | 1] (bprx_sparse.hip).  Both products read every
// feature row exactly once per step; they are HBM-bound (8 KB bf16 row vs 655 KFLOP), so the layout goal is
// full-width coalesced row streaming with enough bytes in flight, and MFMA only has to keep up:
//   bf16 path: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  Bp rides along as column d of [E|Bp|0] (N padded
//              to a multiple of 16).
//   fp32 path: exact reference-precision path for small configs (fp64 accumulate on the vector ALU).
#include "bprx_internal.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ uint16_t f2bf(float x) {   // round-to-nearest-even; inputs are finite
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// OCP e4m3fn code of x: round-to-nearest-even, saturating at +-448 (written out instead of v_cvt_pk_fp8_f32 so that
// the CPU oracle's restatement is the same arithmetic bit for bit).  x is finite.
__device__ __forceinline__ uint32_t f2e4m3(float x) {
  const uint32_t u = __float_as_uint(x), sign = (u >> 24) & 0x80u, a = u & 0x7fffffffu;
  const float ax = __uint_as_float(a);
  if (ax < 0.015625f) return sign | (uint32_t)__builtin_rintf(ax * 512.0f);        // below 2^-6: multiples of 2^-9
  const uint32_t rr = a + 0x7ffffu + ((a >> 20) & 1u);                             // keep 3 mantissa bits, RNE
  const int e = (int)(rr >> 23) - 127;
  const uint32_t m = (rr >> 20) & 7u;
  if (e > 8 || (e == 8 && m == 7u)) return sign | 0x7eu;                           // 448
  return sign | ((uint32_t)(e + 7) << 3) | m;
}

typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) long i64x2_t;
// A bf16 feature table is streamed once per projection (C2: 410 MB per pass, more than the 256-MiB Infinity Cache holds):
// its loads carry the `nt` bit (streaming, no-retain), so that a pass does not push the step's RE-USED tables (P, the
// factor tables, W: ~100 MB at C2) out of the Infinity Cache -- C2 0.2596 -> 0.2438 ms/step (proj_fwd 83.0 -> 76.8,
// proj_bwd 72.6 -> 69.4, triplet_grad 38.1 -> 35.2 us), c4 shard 0.3561 -> 0.3503.  NOT for fp8 tables: C2's and C5's fp8
// table (205 MB) fits the cache and the backward pass re-reads what the forward pass left there (c2fp8 with nt: 0.2091 ->
// 0.2152 ms/step), nor in the v8 / f8s forward kernels (measured slower with it: c4 shard 118.8 -> 121.6, c5 74.2 -> 79.2 us).
template <bool NT>
__device__ __forceinline__ i32x4_t ld_stream16(const void *p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const i32x4_t *>(p));
  else return *reinterpret_cast<const i32x4_t *>(p);
}
// one 16-byte operand fragment per lane: 8 bf16 (one 16x16x32 MFMA) or 16 fp8 (two 16x16x32 fp8 MFMAs over the low and
// the high 8 bytes; A and B fragments are cut the same way, so every k is paired with itself exactly once)
template <bool F8>
__device__ __forceinline__ f32x4 mfma_frag(i32x4_t a, i32x4_t b, f32x4 c) {
  if constexpr (F8) {
    const i64x2_t av = __builtin_bit_cast(i64x2_t, a), bv = __builtin_bit_cast(i64x2_t, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av.x, bv.x, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av.y, bv.y, c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

// max |x| over E and Bp as the bit pattern of a non-negative float (monotonic as uint32): qs[0]
__global__ __launch_bounds__(256) void k_absmax(const float *__restrict__ E, size_t nE, const float *__restrict__ Bp, size_t nB,
                                                uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nE + nB; e += (size_t)gridDim.x * 256) {
    const float v = e < nE ? E[e] : Bp[e - nE];
    const uint32_t a = __float_as_uint(v) & 0x7fffffffu;
    m = a > m ? a : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
  __shared__ uint32_t wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {                               // one same-address atomic per workgroup (~15 ns each)
    uint32_t b = wm[0];
    for (int w = 1; w < 4; ++w) b = wm[w] > b ? wm[w] : b;
    if (b) atomicMax(out, b);
  }
}

// fp8 images of [E|Bp|0]^T: e4m3fn(x * sE), sE = 448 / max|E,Bp|.  One block per 16 k-rows: the codes of a (column n,
// 16 consecutive k) run are one 16-byte piece in each of the three images, written with one store each:
//   Et   chunk-major with 256-wide chunks (256-byte rows: the same BYTE layout as the bf16 image with 128-wide chunks, so
//        the forward kernels are shared):      ((k/256)*PS + n)*256 + k%256
//   EtF  fragment-major for k_proj_fwd_rows:   ((((k/256)*4 + (k%256)/64) * PS/16 + n/16) * 64 + ((k%64)/16)*16 + n%16) * 16
//   EtS  (optional, wide projections) the order of k_proj_fwd_f8s:
//                                              ((((k/128) * PS/16 + n/16) * 2 + (k%32)/16) * 64 + ((k%128)/32)*16 + n%16) * 16
// Block 0 also publishes qs[1] = 1 / (feat_scale * sE), the factor that turns the fp8 products back into P, and clears
// the absmax slot of the NEXT cast.
__global__ __launch_bounds__(256) void k_cast_Et8(const float *__restrict__ E, const float *__restrict__ Bp,
                                                  uint8_t *__restrict__ Et, uint8_t *__restrict__ EtF, uint8_t *__restrict__ EtS,
                                                  int D, int d, int PS, float *__restrict__ qs, float feat_scale, int slot) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[288][16];        // [n][k within the 16-row group]
  const float amax = qs[2 + slot];                         // k_absmax / k_dense_update; the other slot is cleared here for
  const float sE = amax > 0.f ? 448.0f / amax : 1.0f;      // the next round (no memset launch per step)
  if (blockIdx.x == 0 && threadIdx.x == 0) { qs[1] = 1.0f / (feat_scale * sE); qs[2 + (slot ^ 1)] = 0.f; }
  const int k0 = blockIdx.x * 16, NT = PS >> 4;
  for (int idx = threadIdx.x; idx < 16 * PS; idx += 256) {
    const int kr = idx / PS, n = idx - kr * PS, kk = k0 + kr;
    float v = 0.f;
    if (kk < D) v = n < d ? E[(size_t)kk * d + n] : (n == d ? Bp[kk] : 0.f);
    tile[n][kr] = (uint8_t)f2e4m3(v * sE);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < PS; n += 256) {
    const uint4 v = *reinterpret_cast<const uint4 *>(&tile[n][0]);
    const int b = k0 & 255, bs = k0 & 127;
    *reinterpret_cast<uint4 *>(Et + ((size_t)(k0 >> 8) * PS + n) * 256 + b) = v;
    *reinterpret_cast<uint4 *>(EtF + (((((size_t)(k0 >> 8) * 4 + (b >> 6)) * NT + (n >> 4)) * 64) + ((b >> 4) & 3) * 16 + (n & 15)) * 16) = v;
    if (EtS)
      *reinterpret_cast<uint4 *>(EtS + (((((size_t)(k0 >> 7) * NT + (n >> 4)) * 2 + ((bs >> 4) & 1)) * 64) + (bs >> 5) * 16 + (n & 15)) * 16) = v;
  }
}

// [E|Bp|0]^T in bf16, stored CHUNK-MAJOR: element (n, k) at ((k/128)*PS + n)*128 + k%128, i.e. each 128-wide k-chunk
// of all PS rows is one contiguous PS*256-byte block.  (A plain [PS][D] image has 8-KB rows: the rows of one chunk
// then sit at a power-of-two stride and every workgroup's chunk load lands on the same few L2 channels.)
__device__ __forceinline__ size_t et_idx(int n, int k, int PS) { return ((size_t)(k >> 7) * PS + n) * 128 + (k & 127); }

// TILED copy of the frozen feature table (made once at bprx_bind_tables; the projections read only this copy).
// Units: 16-bit elements (a bf16 row has D of them, an fp8 row D/2 -- "Deq").  The matrix is cut into blocks of
// 32 items x 128 units = 8 KB, each block contiguous, blocks ordered item-block-major:
//     unit (t, c)  ->  ((t >> 5) * (Deq / 128) + (c >> 7)) * 4096 + (t & 31) * 128 + (c & 127)
// Why: both projections consume F in column slices (forward: 16 items x 256 B per k-chunk and wave; backward: 32 items x
// 256-512 B per tile).  In the row-major table those slices are 64-512-B pieces 8 KB apart -- the stand-alone probe
// (scripts/probe/ldpat.hip) streams that pattern at 4.5-5.3 TB/s, the same fragment loads inside contiguous 4-KB
// blocks at 5.4-5.9 TB/s.  Items past the end of the table are zero rows of the last block (no bounds masks needed).
__device__ __forceinline__ size_t ft_row(int t, int Deq) { return ((size_t)(t >> 5) * (Deq >> 7)) * 4096 + (size_t)(t & 31) * 128; }
__device__ __forceinline__ size_t ft_col(int c) { return (size_t)(c >> 7) * 4096 + (c & 127); }

__global__ __launch_bounds__(256) void k_tile_F(const uint16_t *__restrict__ F, uint16_t *__restrict__ Ft, int I, int Deq) {
  // one workgroup per 8-KB block: 512 pieces of 16 B; piece p = row (p >> 4), 16-B column (p & 15)
  const int cb = blockIdx.x, tb = blockIdx.y;
  uint4 *dst = reinterpret_cast<uint4 *>(Ft + ((size_t)tb * (Deq >> 7) + cb) * 4096);
  for (int p = threadIdx.x; p < 512; p += 256) {
    const int t = tb * 32 + (p >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t < I) v = *reinterpret_cast<const uint4 *>(F + (size_t)t * Deq + cb * 128 + (p & 15) * 8);
    dst[p] = v;
  }
}

// Et(n, k) = E[k][n] for n < d ; Bp[k] for n == d ; 0 above.  One block per (64 k-rows x 16 columns) tile, transposed
// through LDS so that every Et row segment is written as 128 contiguous bytes.
// EtF: the same values FRAGMENT-MAJOR for k_proj_fwd_rows -- element (n, k) at
//   ((((k/128)*4 + (k%128)/32) * (PS/16) + n/16) * 64 + ((k%32)/8)*16 + n%16) * 8 + k%8 :
// the 64 lanes x 8 elements of one (chunk, k-step, column tile) MFMA B fragment are 1 KB of contiguous bytes.
__global__ __launch_bounds__(256) void k_cast_Et(const float *__restrict__ E, const float *__restrict__ Bp,
                                                 uint16_t *__restrict__ Et, uint16_t *__restrict__ EtF, int D, int d, int PS) {
  __shared__ float tile[64][17];
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 16;
  for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
    const int kr = idx >> 4, nc = idx & 15, kk = k0 + kr, n = n0 + nc;
    float v = 0.f;
    if (kk < D) v = n < d ? E[(size_t)kk * d + n] : (n == d ? Bp[kk] : 0.f);
    tile[kr][nc] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
    const int nc = idx >> 6, kr = idx & 63, kk = k0 + kr, n = n0 + nc;
    if (n < PS && kk < D) {
      const uint16_t v = f2bf(tile[kr][nc]);
      Et[et_idx(n, kk, PS)] = v;
      const int e = kk & 127;
      EtF[(((((size_t)(kk >> 7) * 4 + (e >> 5)) * (PS >> 4) + (n >> 4)) * 64) + ((e >> 3) & 3) * 16 + (n & 15)) * 8 + (e & 7)] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// forward, bf16.  Workgroup = 4 waves; wave w owns MT tiles of 16 rows; all NT column tiles stay in registers.
// A (feature rows) goes HBM -> VGPR directly in MFMA operand order (lane: row l&15, 16 B at k = 8*(l>>4));
// B ([E|Bp]^T chunk, shared by the 4 waves and re-read from L2 by every workgroup) is staged through LDS.
// ------------------------------------------------------------------------------------------------------------
constexpr int KC = 128;          // k-chunk staged per barrier pair
constexpr int BS_STRIDE = KC + 8;  // bf16 elements; 272-B rows keep ds_read_b128 nearly conflict-free (v1)

// F8: the operands are fp8 (e4m3fn) bytes; F / Et are then addressed as if they were bf16 matrices of half the width
// (D = feat_dim / 2: identical byte layout), and P is rescaled by *pscale.
template <int NT, int MT, bool F8>
__global__ __launch_bounds__(256) void k_proj_fwd_bf16(const uint16_t *__restrict__ F, const int32_t *__restrict__ rows,
                                                       int nrows, int nitems, int D, const uint16_t *__restrict__ Et,
                                                       float *__restrict__ P, int PS, int32_t *errflag, int stagger,
                                                       const float *__restrict__ pscale) {
  __shared__ __attribute__((aligned(16))) uint16_t Bs[NT * 16 * BS_STRIDE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = (blockIdx.x * 4 + w) * MT * 16;
  // Every workgroup walks the same [E|Bp]^T chunks; started together they would all hit the same few L2 lines
  // at the same moment.  `stagger` rotates the chunk order per workgroup (only the fp32 summation order changes).
  const int nchunks = D / KC;
  const int cshift = (stagger & 1) ? (int)((blockIdx.x >> 3) % (unsigned)nchunks) : 0;
  const bool dbg_skip_b = stagger & 2, dbg_skip_a = stagger & 4;   // timing-only ablations (wrong results)
  const uint16_t *arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int t = row0 + mt * 16 + r;
    if (t >= nrows) t = nrows - 1;                      // padding lanes re-read the last row; never stored
    int item = rows ? rows[t] : t;
    if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
    arow[mt] = F + ft_row(item, D) + q * 8;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int cc = 0; cc < nchunks; ++cc) {
    int ce = cc + cshift;
    if (ce >= nchunks) ce -= nchunks;
    const int k0 = ce * KC;
    __syncthreads();
    if (!(dbg_skip_b && cc > 0))
    for (int idx = threadIdx.x; idx < NT * 16 * (KC / 8); idx += 256) {
      const int n = idx / (KC / 8), kk = (idx % (KC / 8)) * 8;
      *reinterpret_cast<uint4 *>(&Bs[n * BS_STRIDE + kk]) = *reinterpret_cast<const uint4 *>(&Et[et_idx(n, k0 + kk, NT * 16)]);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KC; ks += 32) {
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const bf16x8 *>(arow[mt] + ft_col(((dbg_skip_a && cc > 0) ? 0 : k0) + ks));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Bs[(nt * 16 + r) * BS_STRIDE + ks + q * 8]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt] = mfma_frag<F8>(__builtin_bit_cast(i32x4_t, a[mt]), __builtin_bit_cast(i32x4_t, b), acc[mt][nt]);
      }
    }
  }
  const float ps = F8 ? *pscale : 1.0f;
  // C/D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4)*4 + reg
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = row0 + mt * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + nt * 16 + r] = acc[mt][nt][reg] * ps;
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward over a ROW LIST (sparse batches: the batch's distinct items; bprx_score_pairs: one row per pair).  Few rows,
// so the parallelism has to come from K: one NW-wave workgroup per MT*16 listed rows and NTW column tiles, the waves
// split the k-chunks round-robin (wave w: chunks w, w+NW, ...).  Every wave loads its A fragments (feature rows of the
// tiled F, gathered through the list) and its B fragments straight into MFMA operand order -- no two waves need the
// same bytes, so nothing is staged through LDS and the loop has no barrier -- and the partial accumulators meet in LDS
// in a fixed tree order (bit-reproducible).
//   B comes from the FRAGMENT-MAJOR image EtF of [E|Bp]^T (k_cast_Et): the 64 x 16 B of one (chunk, k-step, column
//   tile) fragment are contiguous in lane order, so a wave load is 1 KB of whole cache lines.  (From the chunk-major
//   image the same fragment is 16 rows x 64 B: measured ~18 B/clk per CU through the vector L1 -- a workgroup that
//   streams the whole 0.65-MB image took ~17 us whatever the row count.)
//   NTW < NT (tiny launches: fewer row tiles than CUs): the column tiles are split over blockIdx.y, each workgroup then
//   streams only its NTW/NT share of the image and the row tile's A bytes are re-read from L2 by its siblings.
//   All feature fragments of a wave's chunk group are requested before the first MFMA (independent 16-B loads: the rows
//   are random 8-KB rows of a table far larger than any cache, their round trips must overlap), the B fragments run
//   one k-step ahead; scheduling fences keep hipcc from sinking the loads next to their MFMAs.
// nrows_dev: the list length is only known on the device (k_row_count builds the list): the grid is sized for the
// host-side bound `nrows`, surplus workgroups leave at once.  scatter: result row t goes to P[rows[t]].
// ------------------------------------------------------------------------------------------------------------
template <int NTW, int MT, int NW, bool F8>
__global__ __launch_bounds__(NW * 64) void k_proj_fwd_rows(const uint16_t *__restrict__ F, const int32_t *__restrict__ rows,
                                                       int nrows, const int32_t *__restrict__ nrows_dev, int nitems, int D,
                                                       const uint16_t *__restrict__ EtF, float *__restrict__ P, int PS,
                                                       int32_t *errflag, const float *__restrict__ pscale, int scatter) {
  extern __shared__ __attribute__((aligned(16))) float red_rows[];   // [NW/2 waves][NTW*4 registers][64 lanes]
  // chunks per group: up to 16 A fragments in flight per lane where the register budget allows (128 per lane at 16 waves,
  // 256 at 8; wide projections hold NTW*4 accumulators and 2*NTW*4 B fragments)
  constexpr int CG = NW == 16 ? 2 : (MT == 1 ? (NTW <= 9 ? 4 : (NTW <= 13 ? 2 : 1)) : (MT == 2 ? 2 : 1));
  constexpr int KS = KC / 32;
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  const int row0 = blockIdx.x * (MT * 16);
  if (row0 >= nrows) return;                              // workgroup-uniform
  const int NT = PS >> 4, nt0 = blockIdx.y * NTW;         // this workgroup's column tiles [nt0, nt0 + NTW) (clamped below)
  // (readfirstlane: the wave index is uniform, which hipcc cannot prove from threadIdx -- chunk indices and the B offsets
  //  derived from it then live in SGPRs instead of per-lane 64-bit address pairs)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const uint16_t *arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int t = row0 + mt * 16 + r;
    if (t >= nrows) t = nrows - 1;                        // padding lanes re-read the last row; never stored
    int item = rows ? rows[t] : t;
    if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
    arow[mt] = F + ft_row(item, D) + q * 8;
  }
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = D / KC;
  // fragment (c, ks, nt) of EtF: (((c*KS + ks)*NT + nt)*64 + lane)*8 units; column tiles past the end repeat the last
  // one.  Byte offsets in 32 bits (the image is PS*D*2 <= 2.3 MB): uniform base + per-lane 32-bit offset loads.
  const unsigned char *eb = reinterpret_cast<const unsigned char *>(EtF);
  uint32_t boff[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) { const int n = nt0 + nt < NT ? nt0 + nt : NT - 1; boff[nt] = (uint32_t)(n * 64 + lane) * 16u; }
  const uint32_t kstep = (uint32_t)NT * 1024u;            // bytes per (c, ks)
  // wave w owns chunks w, w + NW, ...; a group = CG of them (clamped re-reads past the end, masked out of the sums)
  for (int c0 = w; c0 < nch; c0 += NW * CG) {
    i32x4_t a[CG][KS][MT];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
      int c = c0 + cg * NW;
      c = c < nch ? c : c0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[cg][ks][mt] = *reinterpret_cast<const i32x4_t *>(arow[mt] + ((size_t)c << 12) + ks * 32);
    }
    i32x4_t b[2][NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) b[0][nt] = *reinterpret_cast<const i32x4_t *>(eb + ((uint32_t)(c0 * KS) * kstep + boff[nt]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < CG * KS; ++st) {
      const int cg = st / KS, ks = st % KS;
      if (st + 1 < CG * KS) {                               // next k-step's B fragments before this step's MFMAs
        const int cg1 = (st + 1) / KS, ks1 = (st + 1) % KS;
        int c1 = c0 + cg1 * NW;
        c1 = c1 < nch ? c1 : c0;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          b[(st + 1) & 1][nt] = *reinterpret_cast<const i32x4_t *>(eb + ((uint32_t)(c1 * KS + ks1) * kstep + boff[nt]));
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c0 + cg * NW < nch) {                             // wave-uniform
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_frag<F8>(a[cg][ks][mt], b[st & 1][nt], acc[mt][nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float ps = F8 ? *pscale : 1.0f;
  // pairwise tree over the waves (w += w + half, half = NW/2 .. 1): the same order every run
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
      if (w >= half && w < 2 * half) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) red_rows[(((w - half) * NTW + nt) * 4 + reg) * 64 + lane] = acc[mt][nt][reg];
      }
      __syncthreads();
      if (w < half) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[mt][nt][reg] += red_rows[((w * NTW + nt) * 4 + reg) * 64 + lane];
      }
      __syncthreads();
    }
    if (w == 0) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {                 // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4)*4 + reg
        const int t = row0 + mt * 16 + q * 4 + reg;
        if (t < nrows) {
          int o = t;
          if (scatter) { o = rows[t]; if ((unsigned)o >= (unsigned)nitems) o = 0; }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            if (nt0 + nt < NT) P[(size_t)o * PS + (nt0 + nt) * 16 + r] = acc[mt][nt][reg] * ps;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, bf16.  out[m, n] = sum_t F[t, m] * W[t, n]: both operands are stored t-major but the MFMA wants
// the reduction index contiguous per lane, so 32-item tiles of F (128 columns) and W go through LDS and are
// read back with the hardware transpose read ds_read_b64_tr_b16.  grid = (D/128 column ranges, SK item splits);
// every workgroup writes its fp32 partial [128, PS] slab with plain stores; k_reduce_parts sums the SK slabs in a
// fixed order (bit-reproducible, no float atomics).
// ------------------------------------------------------------------------------------------------------------
constexpr int BT = 32;                 // items per tile = one MFMA k-step
constexpr int FS_STRIDE = 128 + 16;    // 288-B rows: 8 rows x 32 B cover the 64 banks exactly once
template <int NT>
struct WsStride {                      // bf16 elements; (bytes/4) % 64 == 8
  static constexpr int raw = NT * 16;
  static constexpr int value = ((raw * 2 + 255 - 32) / 256) * 128 + 16;
};

__device__ __forceinline__ bf16x4 lds_tr16(const uint16_t *p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
  return __builtin_bit_cast(bf16x4, v);
}

template <int NT>
__global__ __launch_bounds__(256) void k_proj_bwd_bf16(const uint16_t *__restrict__ F, int nrows, int D,
                                                       const float *__restrict__ W, int PS, float *__restrict__ part,
                                                       int rows_per_split) {
  constexpr int WS = WsStride<NT>::value;
  __shared__ __attribute__((aligned(16))) uint16_t Fs[BT * FS_STRIDE];
  __shared__ __attribute__((aligned(16))) uint16_t Ws[BT * WS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
  const int m0 = blockIdx.x * 128;
  const int tbeg = blockIdx.y * rows_per_split;
  int tend = tbeg + rows_per_split;
  if (tend > nrows) tend = nrows;
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int t0 = tbeg; t0 < tend; t0 += BT) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < BT * 16; idx += 256) {       // F tile: 32 rows x 16 chunks of 16 B
      const int tr = idx >> 4, ch = idx & 15, t = t0 + tr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t < tend) v = *reinterpret_cast<const uint4 *>(F + ft_row(t, D) + ft_col(m0 + ch * 8));
      *reinterpret_cast<uint4 *>(&Fs[tr * FS_STRIDE + ch * 8]) = v;
    }
    for (int idx = threadIdx.x; idx < BT * NT * 4; idx += 256) {   // W tile: fp32 -> bf16
      const int tr = idx / (NT * 4), c4 = idx % (NT * 4), t = t0 + tr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < tend) v = *reinterpret_cast<const float4 *>(W + (size_t)t * PS + c4 * 4);
      uint2 pk;
      pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
      pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
      *reinterpret_cast<uint2 *>(&Ws[tr * WS + c4 * 4]) = pk;
    }
    __syncthreads();
    // lane 16g + 4qq + p supplies row (8g + qq [+4]) , columns 4p..4p+3 of a 16-column block and receives
    // column i16 of rows 8g..8g+3 [+4]: exactly the 16x16x32 operand order  X[row/col i16][k = 8g + j].
    bf16x8 a[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int col = (w * 2 + mt) * 16 + 4 * p;
      bf16x4 lo = lds_tr16(&Fs[(8 * g + qq) * FS_STRIDE + col]);
      bf16x4 hi = lds_tr16(&Fs[(8 * g + qq + 4) * FS_STRIDE + col]);
      a[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + 4 * p;
      bf16x4 lo = lds_tr16(&Ws[(8 * g + qq) * WS + col]);
      bf16x4 hi = lds_tr16(&Ws[(8 * g + qq + 4) * WS + col]);
      const bf16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[mt][nt], 0, 0, 0);
    }
  }
  float *slab = part + ((size_t)blockIdx.y * D + m0) * PS;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = (w * 2 + mt) * 16 + g * 4 + reg;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) slab[(size_t)m * PS + nt * 16 + i16] = acc[mt][nt][reg];
    }
}


// Wb (bf16 [I][PS]) = W (fp32), and W is re-zeroed for the next step in the same pass.
__global__ __launch_bounds__(256) void k_cast_W(float *__restrict__ W, uint16_t *__restrict__ Wb, size_t n4) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<float4 *>(W)[e];
    uint2 pk;
    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    reinterpret_cast<uint2 *>(Wb)[e] = pk;
    reinterpret_cast<float4 *>(W)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward v3: as v2 with (a) W pre-cast to bf16 (k_cast_W) so the tile loads are plain 16-B copies, and (b) an
// LDS image without bank conflicts for ds_read_b64_tr_b16.  One 32-lane half of a transpose read touches tile rows
// {8g+qq} for two values of g, i.e. rows r and r+8 together: with 288-B rows (8 dwords mod 64) those alias, so rows
// with bit 3 set are displaced by 128 B -- XOR inside the 256-B F row, an added offset inside the padded W row.
// ------------------------------------------------------------------------------------------------------------
template <int NT>
struct WsStride3 {                     // bytes; >= NT*32 + 128 and == 32 (mod 256)
  static constexpr int bytes = ((NT * 32 + 128 - 32 + 255) / 256) * 256 + 32;
};

// NW waves per workgroup, 32 feature columns per wave: a workgroup owns NW*32 columns, so every W tile it loads from L2
// serves NW*32 columns (NW = 8: half the W re-read traffic of NW = 4; W tiles are 5/8 of an F tile at NW = 4, NT = 5).
// 16 fp8 (e4m3fn) -> 16 bf16, exact (hardware decode + truncation of an exactly representable value)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ void fp8x4_to_bf16x4(uint32_t w, uint32_t &o0, uint32_t &o1) {
  const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  o0 = (__float_as_uint(a.x) >> 16) | (__float_as_uint(a.y) & 0xffff0000u);
  o1 = (__float_as_uint(b.x) >> 16) | (__float_as_uint(b.y) & 0xffff0000u);
}

// F8: F holds fp8 codes (1 byte per element): the tile loads move half the bytes and the codes are widened to bf16 on
// the way into LDS (W stays bf16); the slabs then hold (F*feat_scale)^T W and are rescaled where they are summed.
// ROWS: the sum runs over the LISTED items only (sparse batches): tile row p is item rows[p], its feature pieces are
// gathered from the tiled F (256 contiguous bytes per item and 128-column block) and its W row from the fp32 table Wf
// (rows indexed by item id, as k_triplet_grad accumulated them; rounded to bf16 on the way into LDS -- no conversion
// pass), the list length comes from the device (*nrows_dev; `nrows` is the host-side bound) and the item splits are cut
// from it here.
// DB = 2: two LDS images used by alternate tiles -- the barrier that protects an image from being overwritten while other
// waves still read it disappears (one barrier per tile instead of two) and a wave's commit of tile t+1 (for fp8 tables:
// the widening to bf16) overlaps the other waves' MFMAs of tile t.  For the MFMA-paced shapes (fp8 tables, wide projections).
template <int NT, int BTV, int NW, int PD, bool F8, bool ROWS = false, int DB = 1>
__global__ __launch_bounds__(NW * 64) void k_proj_bwd_bf16_v3(const uint16_t *__restrict__ F, int nrows, int D,
                                                          const uint16_t *__restrict__ Wb, int PS, float *__restrict__ part,
                                                          int rows_per_split, int descend, int xcd_map,
                                                          const int32_t *__restrict__ rows = nullptr,
                                                          const int32_t *__restrict__ nrows_dev = nullptr,
                                                          const float *__restrict__ Wf = nullptr) {
  constexpr int NTH = NW * 64, MC = NW * 32;         // threads, feature columns per workgroup
  constexpr int ESZ = F8 ? 1 : 2;                    // bytes per feature element in HBM
  constexpr int FCH = MC * ESZ / 16;                 // 16-B pieces per F tile row (HBM side)
  constexpr int FSB = MC * 2 + 32;                   // F tile row stride, bytes (== 32 mod 256)
  constexpr int WSB = WsStride3<NT>::bytes;
  constexpr int FPT = BTV * FCH / NTH;               // 16-B F pieces per thread and tile
  constexpr int CBK = MC * ESZ / 256;                // 8-KB blocks of the tiled F per 32 tile rows
  static_assert(CBK >= 1 && BTV % 32 == 0 && (BTV * FCH) % NTH == 0, "tile must be whole 8-KB blocks");
  constexpr int WCH = NT * 2;                        // 16-B pieces per W row
  constexpr int WPT = (BTV * WCH + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_bwd3[];   // DB x (F image | W image)
  constexpr int IMG = BTV * FSB + BTV * WSB;
  unsigned char *const Fs0 = lds_bwd3, *const Ws0 = lds_bwd3 + BTV * FSB;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, i16 = lane & 15, qq = i16 >> 2, p = i16 & 3;
  // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (own L2 each) in dispatch order.  The
  // column ranges of one item split read the same W rows, so they are placed on ONE XCD (W re-fetched 8x otherwise:
  // +65 MB of HBM/MALL fetch per launch measured on C2).
  int bx = blockIdx.x, by = blockIdx.y;
  if (xcd_map && gridDim.y % 8 == 0) {
    const int flat = blockIdx.x + gridDim.x * blockIdx.y, xcd = flat & 7, slot = flat >> 3;
    by = (slot / (int)gridDim.x) * 8 + xcd;
    bx = slot % (int)gridDim.x;
  }
  const int m0 = bx * MC;
  const int Deq = D * ESZ / 2, m0q = m0 * ESZ / 2;   // row width / first column in 16-bit units (tiled F addressing)
  if constexpr (ROWS) {
    const int n = *nrows_dev;
    nrows = n < nrows ? n : nrows;
    rows_per_split = ((nrows + (int)gridDim.y - 1) / (int)gridDim.y + BTV - 1) / BTV * BTV;
  }
  const int tbeg = by * rows_per_split;
  int tend = tbeg + rows_per_split;
  if (tend > nrows) tend = nrows;
  const int ntiles = tend > tbeg ? (tend - tbeg + BTV - 1) / BTV : 0;
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // PD tiles are in flight per workgroup (registers: PD * (FPT + WPT) * 4 VGPRs): the loop is latency-bound on the
  // global loads, so bytes in flight per CU set the delivered bandwidth
  uint4 freg[PD][FPT], wreg[PD][WPT];
  uint4 wreg2[ROWS ? PD : 1][WPT];                   // ROWS: a 16-B bf16 piece of W is 32 B of the fp32 row
  // Loads are unconditional (rows / tiles past the end are clamped to the last valid one and zeroed at commit):
  // with a load inside a divergent branch the compiler waits for vmcnt(0) at every commit and the pipeline collapses.
#define BWD3_ISSUE(ST, TILE)                                                                                             \
  {                                                                                                                      \
    int tile_ = (TILE);                                                                                                  \
    tile_ = tile_ < ntiles ? tile_ : ntiles - 1;                                                                         \
    const int t0 = tbeg + (descend ? (ntiles - 1 - tile_) : tile_) * BTV;                                                \
    /* tiled F: the tile is (BTV/32) x CBK contiguous 8-KB blocks; piece p lies in block p >> 9 at 16-B slot p & 511 */ \
    _Pragma("unroll") for (int x = 0; x < FPT; ++x) {                                                                    \
      const int pp = threadIdx.x + x * NTH, blk = pp >> 9;                                                               \
      if constexpr (ROWS) {                                                                                              \
        int tp = t0 + (blk / CBK) * 32 + ((pp & 511) >> 4);                                                              \
        tp = tp < tend ? tp : tend - 1;                                                                                  \
        const int item = rows[tp];                                                                                       \
        freg[ST][x] = *reinterpret_cast<const uint4 *>(F + ft_row(item, Deq) + (size_t)((m0q >> 7) + blk % CBK) * 4096 + \
                                                       (size_t)(pp & 15) * 8);                                           \
      } else {                                                                                                           \
        const size_t bidx = (size_t)((t0 >> 5) + blk / CBK) * (Deq >> 7) + (m0q >> 7) + blk % CBK;                       \
        { const i32x4_t v_ = ld_stream16<!F8>(F + bidx * 4096 + (size_t)(pp & 511) * 8);                                 \
          freg[ST][x] = make_uint4((unsigned)v_.x, (unsigned)v_.y, (unsigned)v_.z, (unsigned)v_.w); }                      \
      }                                                                                                                  \
    }                                                                                                                    \
    _Pragma("unroll") for (int x = 0; x < WPT; ++x) {                                                                    \
      int idx = threadIdx.x + x * NTH;                                                                                   \
      idx = idx < BTV * WCH ? idx : BTV * WCH - 1;                                                                       \
      const int tr = idx / WCH, ch = idx % WCH;                                                                          \
      int t = t0 + tr;                                                                                                   \
      t = t < tend ? t : tend - 1;                                                                                       \
      if constexpr (ROWS) {                                                                                              \
        const float *wr = Wf + (size_t)rows[t] * PS + ch * 8;                                                            \
        wreg[ST][x] = *reinterpret_cast<const uint4 *>(wr);                                                              \
        wreg2[ST][x] = *reinterpret_cast<const uint4 *>(wr + 4);                                                         \
      } else {                                                                                                           \
        wreg[ST][x] = *reinterpret_cast<const uint4 *>(Wb + (size_t)t * PS + ch * 8);                                    \
      }                                                                                                                  \
    }                                                                                                                    \
  }
#define BWD3_COMMIT(ST, TILE)                                                                                            \
  {                                                                                                                      \
    const int tile_ = (TILE);                                                                                            \
    unsigned char *const Fs = Fs0 + (DB == 2 ? (tile_ & 1) * IMG : 0), *const Ws = Ws0 + (DB == 2 ? (tile_ & 1) * IMG : 0); \
    const int t0 = tile_ < ntiles ? tbeg + (descend ? (ntiles - 1 - tile_) : tile_) * BTV : tend;                        \
    _Pragma("unroll") for (int x = 0; x < FPT; ++x) {                                                                    \
      const int pp = threadIdx.x + x * NTH, blk = pp >> 9;                                                               \
      const int tr = (blk / CBK) * 32 + ((pp & 511) >> 4), ch = (blk % CBK) * 16 + (pp & 15);                            \
      const uint32_t mk = (t0 + tr < tend) ? 0xffffffffu : 0u;                                                           \
      uint4 v = freg[ST][x];                                                                                             \
      v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;                                                                        \
      if constexpr (F8) {                                                                                                \
        uint4 lo, hi;                                                                                                    \
        fp8x4_to_bf16x4(v.x, lo.x, lo.y); fp8x4_to_bf16x4(v.y, lo.z, lo.w);                                              \
        fp8x4_to_bf16x4(v.z, hi.x, hi.y); fp8x4_to_bf16x4(v.w, hi.z, hi.w);                                              \
        *reinterpret_cast<uint4 *>(&Fs[tr * FSB + ((ch * 32) ^ ((tr & 8) << 4))]) = lo;                                  \
        *reinterpret_cast<uint4 *>(&Fs[tr * FSB + ((ch * 32 + 16) ^ ((tr & 8) << 4))]) = hi;                             \
      } else {                                                                                                           \
        *reinterpret_cast<uint4 *>(&Fs[tr * FSB + ((ch * 16) ^ ((tr & 8) << 4))]) = v;                                   \
      }                                                                                                                  \
    }                                                                                                                    \
    _Pragma("unroll") for (int x = 0; x < WPT; ++x) {                                                                    \
      const int idx = threadIdx.x + x * NTH, tr = idx / WCH, ch = idx % WCH;                                             \
      const uint32_t mk = (t0 + tr < tend) ? 0xffffffffu : 0u;                                                           \
      uint4 v = wreg[ST][x];                                                                                             \
      if constexpr (ROWS) {                                                                                              \
        const uint4 u = wreg2[ST][x];                                                                                    \
        v.x = (uint32_t)f2bf(__uint_as_float(v.x)) | ((uint32_t)f2bf(__uint_as_float(v.y)) << 16);                       \
        v.y = (uint32_t)f2bf(__uint_as_float(v.z)) | ((uint32_t)f2bf(__uint_as_float(v.w)) << 16);                       \
        v.z = (uint32_t)f2bf(__uint_as_float(u.x)) | ((uint32_t)f2bf(__uint_as_float(u.y)) << 16);                       \
        v.w = (uint32_t)f2bf(__uint_as_float(u.z)) | ((uint32_t)f2bf(__uint_as_float(u.w)) << 16);                       \
      }                                                                                                                  \
      v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;                                                                        \
      if (idx < BTV * WCH) *reinterpret_cast<uint4 *>(&Ws[tr * WSB + ((tr & 8) << 4) + ch * 16]) = v;                    \
    }                                                                                                                    \
  }
  if (ntiles == 0) {                                  // empty split: zero slab
    float *slab0 = part + ((size_t)by * D + m0) * PS;
    for (int e = threadIdx.x; e < MC * PS; e += NTH) slab0[e] = 0.f;
    return;
  }
#pragma unroll
  for (int st = 0; st < PD; ++st) BWD3_ISSUE(st, st)
  for (int tile0 = 0; tile0 < ntiles; tile0 += PD) {
#pragma unroll
   for (int st = 0; st < PD; ++st) {                 // tiles past the end are all-zero: computed, harmless
    if (DB == 1) __syncthreads();                     // (DB == 2: the image of tile t+1 was last read for tile t-1, and every
    BWD3_COMMIT(st, tile0 + st)                       //  wave has passed tile t's barrier since)
    __syncthreads();
    BWD3_ISSUE(st, tile0 + st + PD)
    const unsigned char *const Fs = Fs0 + (DB == 2 ? ((tile0 + st) & 1) * IMG : 0), *const Ws = Ws0 + (DB == 2 ? ((tile0 + st) & 1) * IMG : 0);
#pragma unroll
    for (int kk = 0; kk < BTV / 32; ++kk) {
      const int rlo = kk * 32 + 8 * g + qq, rhi = rlo + 4;       // (rlo & 8) == (rhi & 8) == 8*(g & 1)
      const int disp = (g & 1) << 7;
      bf16x8 a[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int cb = (((w * 2 + mt) * 16 + 4 * p) * 2) ^ disp;
        bf16x4 lo = lds_tr16(reinterpret_cast<const uint16_t *>(&Fs[rlo * FSB + cb]));
        bf16x4 hi = lds_tr16(reinterpret_cast<const uint16_t *>(&Fs[rhi * FSB + cb]));
        a[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int cb = (nt * 16 + 4 * p) * 2 + disp;
        bf16x4 lo = lds_tr16(reinterpret_cast<const uint16_t *>(&Ws[rlo * WSB + cb]));
        bf16x4 hi = lds_tr16(reinterpret_cast<const uint16_t *>(&Ws[rhi * WSB + cb]));
        const bf16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[mt][nt], 0, 0, 0);
      }
    }
   }
  }
#undef BWD3_ISSUE
#undef BWD3_COMMIT
  float *slab = part + ((size_t)by * D + m0) * PS;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = (w * 2 + mt) * 16 + g * 4 + reg;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) slab[(size_t)m * PS + nt * 16 + i16] = acc[mt][nt][reg];
    }
}


// ------------------------------------------------------------------------------------------------------------
// forward v6: explicit ping-pong software pipeline (two named register sets, loop unrolled by two) whose global loads
// and waits are written as inline asm.  hipcc kept re-timing every C++ formulation of the same pipeline (it sank the
// [E|Bp]^T loads next to their LDS store and waited on same-iteration loads inside the MFMA block; see
// profiles/r01_sweeps.md), so every chunk still paid a full memory round trip.  Here every loop load is an asm
// `global_load_dwordx4`; the compiler sees no VMEM event, inserts no vmcnt of its own, and the two counted waits per
// half-iteration are placed by hand (vmcnt retires in issue order: B pieces first, then the A fragments):
//   before parking chunk c+1's B pieces:  vmcnt(KS*MT)        -> only the A fragments of c+1 stay in flight
//   before the MFMAs of chunk c:          vmcnt(NT + KS*MT)   -> everything of c+1 stays in flight
// Each register is then passed through an empty asm ("+v") so no use can be scheduled above its wait.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void asm_gload(i32x4 &dst, const void *p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void asm_tie(i32x4 &r) { asm volatile("" : "+v"(r)); }
// LDS read the compiler can neither sink nor serialize (it turns  read-all-then-MFMA  back into  read, wait, MFMA  per
// fragment to save registers); completion is waited for with counted lgkmcnt (LDS operations return in order).
__device__ __forceinline__ void asm_dsread(i32x4 &dst, const void *lds_ptr) {
  const uint32_t a = (uint32_t)(size_t)(const __attribute__((address_space(3))) void *)lds_ptr;
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a) : "memory");
}
__device__ __forceinline__ void asm_lgkmcnt(int n) {   // n is a constant after unrolling: one case survives
  switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
    default: asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory"); break;
  }
}

// ABL: compile-time timing-only ablations (wrong results): 1 no global loads after the first issue, 2 no MFMA,
// 4 no LDS reads, 8 no LDS writes
template <int NT, int MT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k_proj_fwd_bf16_v6(const uint16_t *__restrict__ F, const int32_t *__restrict__ rows,
                                                          int nrows, int nitems, int D, const uint16_t *__restrict__ Et,
                                                          float *__restrict__ P, int PS, int32_t *errflag, int stagger,
                                                          const float *__restrict__ /*pscale: bf16 only*/) {
  constexpr int BSS = KC + 16;
  constexpr int KS = KC / 32;
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2][NT * 16 * BSS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = (blockIdx.x * 4 + w) * MT * 16;
  const uint16_t *arow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int t = row0 + mt * 16 + r;
    if (t >= nrows) t = nrows - 1;
    int item = rows ? rows[t] : t;
    if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
    arow[mt] = F + ft_row(item, D) + q * 8;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the compiler's own prologue loads are done
  const int bn = threadIdx.x >> 4, bk = (threadIdx.x & 15) * 8;
  const int nch = D / KC;
  const int cshift = (stagger & 1) ? (int)((blockIdx.x >> 3) % (unsigned)nch) : 0;
  const bool nobar = stagger & 2;                       // timing-only ablations (wrong results): no barriers /
  const int amask = (stagger & 4) ? 0 : -1;             // every A load re-reads chunk 0 (cache hits)
  const int bmask = (stagger & 8) ? 0 : -1;             // every B load re-reads chunk 0
  auto kof = [&](int c) { int ce = c + cshift; if (ce >= nch) ce -= nch; return ce * KC; };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4 bX[NT], bY[NT], aX[KS][MT], aY[KS][MT];
#define V6_ISSUE(c_, BR, AR)                                                                                          \
  {                                                                                                                   \
    const int k1 = kof(c_);                                                                                           \
    if (!(ABL & 1) || c_ == 0) {                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) asm_gload(BR[t], &Et[et_idx(t * 16 + bn, (k1 & bmask) + bk, NT * 16)]);      \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)               \
        asm_gload(AR[ks][mt], arow[mt] + ((size_t)((k1 & amask) >> 7) << 12) + ks * 32);                                                               \
    }                                                                                                                 \
  }
#define V6_PARK(buf_, BR, NWAIT)                                                                                      \
  {                                                                                                                   \
    if (!(ABL & 1)) asm_vmcnt<NWAIT>();                                                                               \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) asm_tie(BR[t]);                                                    \
    if (!(ABL & 8))                                                                                                   \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                    \
        *reinterpret_cast<i32x4 *>(&Bs[buf_][(t * 16 + bn) * BSS + bk]) = BR[t];                                      \
  }
#define V6_COMPUTE(buf_, AR, NWAIT)                                                                                   \
  {                                                                                                                   \
    if (!(ABL & 1)) asm_vmcnt<NWAIT>();                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)               \
        asm_tie(AR[ks][mt]);                                                                                          \
    /* B fragments stream from LDS through a window of LWIN reads in flight; fragment f = ks*NT + nt */               \
    i32x4 bfr[KS * NT];                                                                                               \
    if (!(ABL & 4))                                                                                                   \
    _Pragma("unroll") for (int f = 0; f < LWIN && f < KS * NT; ++f)                                                   \
        asm_dsread(bfr[f], &Bs[buf_][((f % NT) * 16 + r) * BSS + (f / NT) * 32 + q * 8]);                             \
    _Pragma("unroll") for (int f = 0; f < KS * NT; ++f) {                                                             \
      const int left = KS * NT - 1 - f;                                                                               \
      if (!(ABL & 4)) asm_lgkmcnt(left < LWIN - 1 ? left : LWIN - 1);                                                 \
      asm_tie(bfr[f]);                                                                                                \
      if (!(ABL & 2))                                                                                                 \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                               \
          acc[mt][f % NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, AR[f / NT][mt]),       \
                                                                    __builtin_bit_cast(bf16x8, bfr[f]), acc[mt][f % NT], 0, 0, 0); \
      if (!(ABL & 4) && f + LWIN < KS * NT)                                                                           \
        asm_dsread(bfr[f + LWIN], &Bs[buf_][(((f + LWIN) % NT) * 16 + r) * BSS + ((f + LWIN) / NT) * 32 + q * 8]);    \
    }                                                                                                                 \
  }
  constexpr int LWIN = 10;   // LDS reads in flight per wave (lgkmcnt counts to 15)
  constexpr int NA = KS * MT, NALL = NT + KS * MT;
  V6_ISSUE(0, bX, aX)
  V6_PARK(0, bX, NA)
  __syncthreads();
  // No branch may separate an asm load from its wait: at a control-flow join the compiler is free to copy what it
  // believes are finished values (observed: v_mov of in-flight registers in an else-branch -> garbage).  The loop
  // body is therefore straight-line; the last prefetch re-reads chunk nch-1 instead of being skipped.
  for (int c = 0; c < nch; c += 2) {            // nch is even (D % 256 == 0 is required by the launcher)
    V6_ISSUE(c + 1, bY, aY)
    V6_COMPUTE(0, aX, NALL)
    V6_PARK(1, bY, NA)
    if (!nobar) __syncthreads();
    const int cn = c + 2 < nch ? c + 2 : nch - 1;
    V6_ISSUE(cn, bX, aX)
    V6_COMPUTE(1, aY, NALL)
    V6_PARK(0, bX, NA)
    if (!nobar) __syncthreads();
  }
  asm_vmcnt<0>();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm_tie(aX[ks][mt]);
#undef V6_ISSUE
#undef V6_PARK
#undef V6_COMPUTE
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = row0 + mt * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + nt * 16 + r] = acc[mt][nt][reg];
      }
    }
}


// ------------------------------------------------------------------------------------------------------------
// forward v8: the v6 pipeline on ONE 8-wave workgroup per CU with a balanced share of the 16-row tiles.
// Measured (scripts/fwd_scale.py, scripts/probe/ldpat.hip, profiles/r01_sweeps.md): v6 streams F at the ~4.9 TB/s this
// access pattern reaches on the chip ONLY when every CU holds the same number of workgroups (32768 or 65536 items);
// co-resident workgroups do not overlap, so 50000 items = 391 workgroups of 128 items (two on 135 CUs, one on the
// rest) take as long as 65536 items.  Here every CU gets T/G tiles (+-1) and reads the [E|Bp]^T chunks once.
//   grid G (= #CUs while T <= 16 G); workgroup g owns tiles [g*T/G, (g+1)*T/G), at most 16, and has as many waves
//   (5..8) as give every wave TWO tiles (one when the launch is small): the waves walk the chunks in lockstep, so a
//   wave with one tile among waves with two buys nothing (12 tiles on 8 waves: 95.6 us; on 6-7 waves: see sweeps).
//   The pipeline body is instantiated for 2 and for 1 row tile and picked per wave (same barrier sequence in both) --
//   a repeated dummy tile would cost its full load-issue time.
// ------------------------------------------------------------------------------------------------------------
// NWMIN: smallest workgroup (waves) the launcher may pick for this instantiation (fixes the B pieces per thread)
template <int NT, int MT, int NWMIN, bool F8>
__device__ __forceinline__ void v8_body(const uint16_t *const (&arow)[2], const uint16_t *__restrict__ Et, uint16_t (*Bs)[NT * 16 * (KC + 16)],
                                        f32x4 (&acc)[2][NT], int D, int cshift, int r, int q, int et_chunk) {
  constexpr int BSS = KC + 16;
  constexpr int KS = KC / 32;
  constexpr int NPIECE = NT * 16 * (KC / 8);              // 16-B pieces of one [E|Bp]^T chunk (contiguous in Et)
  constexpr int NBP = (NPIECE + NWMIN * 64 - 1) / (NWMIN * 64);   // pieces per thread for the smallest workgroup;
  // piece x of this thread: tid + x*blockDim, surplus slots clamp onto piece NPIECE-1 (one address per wave); the
  // offsets are recomputed at every use (two VALU ops) instead of living in 2*NBP registers
  const int tid = threadIdx.x, bdim = (int)blockDim.x;
  auto piece = [&](int x) { const int pc = tid + x * bdim; return pc < NPIECE ? pc : NPIECE - 1; };
  const int nch = D / KC;
  auto kof = [&](int c) { int ce = c + cshift; if (ce >= nch) ce -= nch; return ce * KC; };
  i32x4 bX[NBP], bY[NBP], aX[KS][MT], aY[KS][MT];
#define V8_ISSUE(c_, BR, AR)                                                                                          \
  {                                                                                                                   \
    const int k1 = kof(c_);                                                                                           \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) asm_gload(BR[x], &Et[(size_t)(k1 >> 7) * et_chunk + piece(x) * 8]); \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)               \
        asm_gload(AR[ks][mt], arow[mt] + ((size_t)(k1 >> 7) << 12) + ks * 32);   /* k1 % 128 == 0: ft_col(k1) + ks*32 */                                                               \
  }
#define V8_PARK(buf_, BR, NWAIT)                                                                                      \
  {                                                                                                                   \
    asm_vmcnt<NWAIT>();                                                                                               \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) asm_tie(BR[x]);                                                   \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) {                                                                 \
      const int pc = piece(x);                                                                                        \
      *reinterpret_cast<i32x4 *>(&Bs[buf_][(pc / (KC / 8)) * BSS + (pc % (KC / 8)) * 8]) = BR[x];                     \
    }                                                                                                                 \
  }
#define V8_COMPUTE(buf_, AR, NWAIT)                                                                                   \
  {                                                                                                                   \
    asm_vmcnt<NWAIT>();                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)               \
        asm_tie(AR[ks][mt]);                                                                                          \
    i32x4 bfr[KS * NT];                                                                                               \
    _Pragma("unroll") for (int f = 0; f < LWIN8 && f < KS * NT; ++f)                                                  \
        asm_dsread(bfr[f], &Bs[buf_][((f % NT) * 16 + r) * BSS + (f / NT) * 32 + q * 8]);                             \
    _Pragma("unroll") for (int f = 0; f < KS * NT; ++f) {                                                             \
      const int left = KS * NT - 1 - f;                                                                               \
      asm_lgkmcnt(left < LWIN8 - 1 ? left : LWIN8 - 1);                                                               \
      asm_tie(bfr[f]);                                                                                                \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                               \
          acc[mt][f % NT] = mfma_frag<F8>(AR[f / NT][mt], bfr[f], acc[mt][f % NT]);                                   \
      if (f + LWIN8 < KS * NT)                                                                                        \
        asm_dsread(bfr[f + LWIN8], &Bs[buf_][(((f + LWIN8) % NT) * 16 + r) * BSS + ((f + LWIN8) / NT) * 32 + q * 8]); \
    }                                                                                                                 \
  }
  constexpr int LWIN8 = NT <= 6 ? 8 : (F8 ? 3 : 4);   // LDS fragment reads in flight (register budget of the wide instantiations)
  constexpr int NA = KS * MT, NALL = NBP + KS * MT;
  V8_ISSUE(0, bX, aX)
  V8_PARK(0, bX, NA)
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {            // straight-line body (see v6); nch is even
    V8_ISSUE(c + 1, bY, aY)
    V8_COMPUTE(0, aX, NALL)
    V8_PARK(1, bY, NA)
    __syncthreads();
    const int cn = c + 2 < nch ? c + 2 : nch - 1;
    V8_ISSUE(cn, bX, aX)
    V8_COMPUTE(1, aY, NALL)
    V8_PARK(0, bX, NA)
    __syncthreads();
  }
  asm_vmcnt<0>();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) asm_tie(aX[ks][mt]);
#undef V8_ISSUE
#undef V8_PARK
#undef V8_COMPUTE
}

// NWMAX = 8 (two waves per SIMD, 256 registers each) up to NT = 9; NWMAX = 4 (one wave per SIMD, the unified 512-register
// file: accumulators in AGPRs) for the wide projections (d >= 112).
template <int NT, int NWMAX, bool F8>
__global__ __launch_bounds__(NWMAX * 64, 1) void k_proj_fwd_bf16_v8(const uint16_t *__restrict__ F, const int32_t *__restrict__ rows,
                                                          int nrows, int nitems, int D, const uint16_t *__restrict__ Et,
                                                          float *__restrict__ P, int PS, int32_t *errflag, int stagger,
                                                          const float *__restrict__ pscale, int tiles_per_wave, int n0) {
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2][NT * 16 * (KC + 16)];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int T = (nrows + 15) >> 4;
  const int t0 = (int)(((long long)blockIdx.x * T) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * T) / gridDim.x);
  int tile[2];
  const uint16_t *arow[2];
  const int first = t0 + w * tiles_per_wave;                               // tiles_per_wave is 1 or 2
  const int nlive = (first < t1 ? 1 : 0) + ((tiles_per_wave == 2 && first + 1 < t1) ? 1 : 0);   // wave-uniform
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    tile[mt] = first + mt;
    if (tile[mt] >= t1) tile[mt] = t1 > t0 ? t1 - 1 : 0;                  // only read by a wave without work
    int t = tile[mt] * 16 + r;
    if (t >= nrows) t = nrows - 1;
    int item = rows ? rows[t] : t;
    if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
    arow[mt] = F + ft_row(item, D) + q * 8;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the compiler's own prologue loads are done
  const int nch = D / KC;
  const int cshift = (stagger & 1) ? (int)(blockIdx.x % (unsigned)nch) : 0;
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NWMIN = NWMAX == 8 ? (NT <= 7 ? 5 : 8) : NWMAX;   // wide tiles: always 8 waves (fewer B pieces per thread)
  // this launch covers the NT column tiles [n0, n0 + NT*16) of the PS-wide projection (wide projections are split
  // over two launches): the chunk images are PS*128 elements apart, the rows of a chunk contiguous from n0*128
  const uint16_t *Et0 = Et + (size_t)n0 * 128;
  if (nlive == 2) v8_body<NT, 2, NWMIN, F8>(arow, Et0, Bs, acc, D, cshift, r, q, PS * 128);
  else v8_body<NT, 1, NWMIN, F8>(arow, Et0, Bs, acc, D, cshift, r, q, PS * 128);   // nlive == 0: a spare wave repeats a tile
  const float ps = F8 ? *pscale : 1.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (mt >= nlive) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = tile[mt] * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + n0 + nt * 16 + r] = acc[mt][nt][reg] * ps;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// forward v9: as v8 (one balanced workgroup per CU), but the A operand (feature rows) is loaded like the backward
// kernel loads its tiles -- every thread moves contiguous 16-B pieces of the tiled F (a 16-row tile x 128-column chunk
// is one contiguous 4-KB run), registers -> LDS, and the MFMA fragments are read from LDS -- instead of 16-row x 64-B
// fragment loads straight into VGPRs.  Plain C++ (the compiler keeps counted vmcnt for unconditional loads, see the
// backward kernel); PD9 chunks in flight in registers, one LDS image, two barriers per chunk.
// ------------------------------------------------------------------------------------------------------------
template <int NT, bool F8>
__global__ __launch_bounds__(512, 1) void k_proj_fwd_bf16_v9(const uint16_t *__restrict__ F, const int32_t *__restrict__ rows,
                                                          int nrows, int nitems, int D, const uint16_t *__restrict__ Et,
                                                          float *__restrict__ P, int PS, int32_t *errflag, int stagger,
                                                          const float *__restrict__ pscale, int tiles_per_wave, int n0) {
  constexpr int RB = 288;                                  // LDS row stride, bytes (256 B of a chunk row + 32)
  constexpr int APT = 8;                                   // A pieces per thread and chunk (16 tiles x 256 pieces / 512)
  constexpr int NPIECE = NT * 16 * 16;                     // 16-B pieces of one [E|Bp]^T chunk
  constexpr int NBP = (NPIECE + 5 * 64 - 1) / (5 * 64);
  constexpr int PD9 = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds9[];
  unsigned char *As = lds9;                                // [256 rows][RB]
  unsigned char *Bs = lds9 + 256 * RB;                     // [NT*16 rows][RB]
  const int tid = threadIdx.x, bdim = (int)blockDim.x;
  const int lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int T = (nrows + 15) >> 4;
  const int t0 = (int)(((long long)blockIdx.x * T) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * T) / gridDim.x);
  const int ntile = t1 - t0, napiece = ntile * 256;        // 16 rows x 16 pieces per tile
  const int first = w * tiles_per_wave;                    // this wave's tiles, workgroup-local numbering
  const int nlive = (first < ntile ? 1 : 0) + ((tiles_per_wave == 2 && first + 1 < ntile) ? 1 : 0);
  // per A piece of this thread: where its row starts in the tiled F (pieces past the end repeat the last one)
  const uint16_t *abase[APT];
  int alds[APT];
#pragma unroll
  for (int x = 0; x < APT; ++x) {
    int pc = tid + x * bdim;
    pc = pc < napiece ? pc : (napiece > 0 ? napiece - 1 : 0);
    int t = t0 * 16 + (pc >> 4);
    if (t >= nrows) t = nrows - 1;
    int item = rows ? rows[t] : t;
    if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
    abase[x] = F + ft_row(item, D) + (pc & 15) * 8;
    alds[x] = (pc >> 4) * RB + (pc & 15) * 16;
  }
#define V9_BPIECE(x) ((tid + (x) * bdim) < NPIECE ? (tid + (x) * bdim) : NPIECE - 1)   /* no lambda: a by-reference capture
                                                                                          puts the arrays in scratch */
  const uint16_t *Et0 = Et + (size_t)n0 * 128;
  const int nch = D / KC;
  const int cshift = (stagger & 1) ? (int)(blockIdx.x % (unsigned)nch) : 0;
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x4_t areg0[APT], areg1[APT], areg2[APT], breg0[NBP], breg1[NBP], breg2[NBP];   // first-class vectors: a uint4 struct copy becomes a memcpy
                                                            // through a private array the compiler then keeps in scratch
#define V9_ISSUE(ST, C)                                                                                              \
  {                                                                                                                  \
    int cc_ = (C);                                                                                                   \
    cc_ = cc_ < nch ? cc_ : nch - 1;                                                                                 \
    int ce_ = cc_ + cshift;                                                                                          \
    if (ce_ >= nch) ce_ -= nch;                                                                                      \
    const size_t kb_ = (size_t)ce_ << 12;                     /* chunk ce_ of a tiled row: ce_ * 4096 units */       \
    _Pragma("unroll") for (int x = 0; x < APT; ++x) areg##ST[x] = *reinterpret_cast<const i32x4_t *>(abase[x] + kb_);  \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x)                                                                  \
        breg##ST[x] = *reinterpret_cast<const i32x4_t *>(Et0 + (size_t)ce_ * (PS * 128) + V9_BPIECE(x) * 8);           \
  }
#define V9_COMMIT(ST)                                                                                                \
  {                                                                                                                  \
    _Pragma("unroll") for (int x = 0; x < APT; ++x) *reinterpret_cast<i32x4_t *>(As + alds[x]) = areg##ST[x];         \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) {                                                                \
      const int pc = V9_BPIECE(x);                                                                                   \
      *reinterpret_cast<i32x4_t *>(Bs + (pc >> 4) * RB + (pc & 15) * 16) = breg##ST[x];                                \
    }                                                                                                                \
  }
#define V9_COMPUTE()                                                                                                 \
  if (nlive > 0) {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < KC / 32; ++ks) {                                                         \
      const i32x4_t a0 = *reinterpret_cast<const i32x4_t *>(As + (first * 16 + r) * RB + ks * 64 + q * 16);          \
      const i32x4_t a1 = *reinterpret_cast<const i32x4_t *>(As + ((nlive == 2 ? first + 1 : first) * 16 + r) * RB + ks * 64 + q * 16); \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                            \
        const i32x4_t b = *reinterpret_cast<const i32x4_t *>(Bs + (nt * 16 + r) * RB + ks * 64 + q * 16);            \
        acc[0][nt] = mfma_frag<F8>(a0, b, acc[0][nt]);                                                               \
        acc[1][nt] = mfma_frag<F8>(a1, b, acc[1][nt]);   /* a wave with one tile computes it twice, stores once */   \
      }                                                                                                              \
    }                                                                                                                \
  }
  // three chunks in flight (stages written out so that the staging registers keep static names); chunks past the end are
  // clamped re-reads of the last chunk and are never committed
  V9_ISSUE(0, 0)
  V9_ISSUE(1, 1)
  V9_ISSUE(2, 2)
  for (int c0 = 0; c0 < nch; c0 += 3) {
    __syncthreads();
    V9_COMMIT(0)
    __syncthreads();
    V9_ISSUE(0, c0 + 3)
    V9_COMPUTE()
    if (c0 + 1 < nch) {                                       // workgroup-uniform
      __syncthreads();
      V9_COMMIT(1)
      __syncthreads();
      V9_ISSUE(1, c0 + 4)
      V9_COMPUTE()
    }
    if (c0 + 2 < nch) {
      __syncthreads();
      V9_COMMIT(2)
      __syncthreads();
      V9_ISSUE(2, c0 + 5)
      V9_COMPUTE()
    }
  }
#undef V9_COMPUTE
#undef V9_ISSUE
#undef V9_COMMIT
#undef V9_BPIECE
  const float ps = F8 ? *pscale : 1.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (mt >= nlive) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = (t0 + first + mt) * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + n0 + nt * 16 + r] = acc[mt][nt][reg] * ps;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// forward v10: v8's balanced one-workgroup-per-CU shape, but the A operand arrives as CONTIGUOUS pieces of the tiled F --
// a 16-row tile x 128-column chunk is one contiguous 4-KB run, lane l moves 16 B at l*16 (+1 KB per load), exactly the load
// shape of the backward kernel (which streams F at the device's ceiling) -- into a WAVE-PRIVATE LDS area, from which the
// MFMA fragments are read back with ds_read_b128.  v8's fragment loads take 16 rows x 64 B per instruction: half of every
// 128-B line per instruction, every line requested by two instructions.  Nothing of A is shared between waves, so the
// detour costs no barrier (v9 tried this load shape with a workgroup-shared image and two barriers per chunk and lost).
// Plain C++ with scheduling fences; B as in v8 (chunk through a double-buffered shared LDS image, one barrier per chunk).
// ------------------------------------------------------------------------------------------------------------
template <int NT, int MT, bool F8>
__device__ __forceinline__ void v10_body(const uint16_t *const (&asrc)[2], const uint16_t *__restrict__ Et, unsigned char *lds,
                                         unsigned char *myA, f32x4 (&acc)[2][NT], int D, int cshift, int lane, int et_chunk) {
  constexpr int BSB = (KC + 16) * 2;                    // B row stride in bytes (288)
  constexpr int BBUF = NT * 16 * BSB;                   // one B buffer
  constexpr int KS = KC / 32;
  constexpr int NPIECE = NT * 16 * (KC / 8);
  const int tid = threadIdx.x, bdim = (int)blockDim.x;
  const int r = lane & 15, q = lane >> 4;
  const int nch = D / KC;
  i32x4_t aX[MT][4], aY[MT][4];
  constexpr int NBR = (NPIECE + 319) / 320;             // B pieces per thread for the smallest workgroup (5 waves)
  i32x4_t bst[NBR];
#define V10_ISSUE(c_, AR)                                                                                             \
  {                                                                                                                   \
    int ce_ = (c_) + cshift;                                                                                          \
    if (ce_ >= nch) ce_ -= nch;                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int x = 0; x < 4; ++x)                   \
        AR[mt][x] = ld_stream16<!F8>(asrc[mt] + ((size_t)ce_ << 12) + x * 512 + lane * 8);                          \
    const uint16_t *bc_ = Et + (size_t)ce_ * et_chunk;                                                                \
    _Pragma("unroll") for (int x = 0; x < NBR; ++x) {                                                                 \
      int pc = tid + x * bdim;                                                                                        \
      pc = pc < NPIECE ? pc : NPIECE - 1;                                                                             \
      bst[x] = *reinterpret_cast<const i32x4_t *>(bc_ + pc * 8);                                                      \
    }                                                                                                                 \
  }
#define V10_PARK(buf_)                                                                                                \
  {                                                                                                                   \
    _Pragma("unroll") for (int x = 0; x < NBR; ++x) {                                                                 \
      int pc = tid + x * bdim;                                                                                        \
      pc = pc < NPIECE ? pc : NPIECE - 1;                                                                             \
      *reinterpret_cast<i32x4_t *>(lds + (buf_) * BBUF + (pc / (KC / 8)) * BSB + (pc % (KC / 8)) * 16) = bst[x];      \
    }                                                                                                                 \
  }
#define V10_COMPUTE(buf_, AR)                                                                                         \
  {                                                                                                                   \
    /* A pieces -> wave-private image (row = x*4 + lane/16, 16-B column lane%16), then the fragments back */         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int x = 0; x < 4; ++x)                   \
        *reinterpret_cast<i32x4_t *>(myA + (mt * 16 + x * 4 + (lane >> 4)) * BSB + (lane & 15) * 16) = AR[mt][x];      \
    i32x4_t af[KS][MT];                                                                                               \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)               \
        af[ks][mt] = *reinterpret_cast<const i32x4_t *>(myA + (mt * 16 + r) * BSB + ks * 64 + q * 16);                \
    const unsigned char *bl_ = lds + (buf_) * BBUF + r * BSB + q * 16;                                                \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                               \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                             \
        const i32x4_t b = *reinterpret_cast<const i32x4_t *>(bl_ + nt * 16 * BSB + ks * 64);                          \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_frag<F8>(af[ks][mt], b, acc[mt][nt]);    \
      }                                                                                                               \
    }                                                                                                                 \
  }
  V10_ISSUE(0, aX)
  V10_PARK(0)
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {                    // nch is even; the last prefetch re-reads chunk nch-1
    V10_ISSUE(c + 1, aY)
    __builtin_amdgcn_sched_barrier(0);
    V10_COMPUTE(0, aX)
    __builtin_amdgcn_sched_barrier(0);
    V10_PARK(1)
    __syncthreads();
    const int cn = c + 2 < nch ? c + 2 : nch - 1;
    V10_ISSUE(cn, aX)
    __builtin_amdgcn_sched_barrier(0);
    V10_COMPUTE(1, aY)
    __builtin_amdgcn_sched_barrier(0);
    V10_PARK(0)
    __syncthreads();
  }
  // (A two chunks ahead -- three register sets in rotation, 128 KB per CU in flight -- was measured SLOWER: 85.0 vs 80.2 us
  //  on C2, 129.5 vs 115.8 us (v8) on the c4 shard; removed)
#undef V10_ISSUE
#undef V10_PARK
#undef V10_COMPUTE
}

template <int NT, bool F8>
__global__ __launch_bounds__(512) void k_proj_fwd_bf16_v10(const uint16_t *__restrict__ F, int nrows, int D,
                                                           const uint16_t *__restrict__ Et, float *__restrict__ P, int PS,
                                                           const float *__restrict__ pscale, int stagger, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_v10[];   // Bs[2] | As[waves][2 tiles][16 rows][288 B]
  constexpr int BSB = (KC + 16) * 2;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int T = (nrows + 15) >> 4;
  const int t0 = (int)(((long long)blockIdx.x * T) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * T) / gridDim.x);
  const int first = t0 + w * tiles_per_wave;
  const int nlive = (first < t1 ? 1 : 0) + ((tiles_per_wave == 2 && first + 1 < t1) ? 1 : 0);   // wave-uniform
  const uint16_t *asrc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int tile = first + mt;
    if (tile >= t1) tile = t1 > t0 ? t1 - 1 : 0;         // only read by a wave without work
    asrc[mt] = F + ft_row(tile * 16, D);                 // the tile's 16 rows x 128 columns: 4 KB contiguous (rows past the
  }                                                      // end of the table are zero rows of the tiled copy)
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nch = D / KC;
  const int cshift = (stagger & 1) ? (int)(blockIdx.x % (unsigned)nch) : 0;
  unsigned char *myA = lds_v10 + 2 * NT * 16 * BSB + w * (2 * 16 * BSB);
  if (nlive == 2) v10_body<NT, 2, F8>(asrc, Et, lds_v10, myA, acc, D, cshift, lane, PS * 128);
  else v10_body<NT, 1, F8>(asrc, Et, lds_v10, myA, acc, D, cshift, lane, PS * 128);
  const float ps = F8 ? *pscale : 1.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (mt >= nlive) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = (first + mt) * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + nt * 16 + r] = acc[mt][nt][reg] * ps;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// forward, fp8 features, WIDE projections (NT >= 10 column tiles: BASELINE.json configs[4], k = d = 256 -> PS = 272) in ONE
// pass over F on the block-scaled fp8 MFMA.  The v8 kernel keeps two waves per SIMD (256 registers each) and therefore
// covers at most nine column tiles per launch: d = 256 took three column-range launches, i.e. three reads of F, on the
// non-scaled fp8 MFMA that runs at the bf16 rate.  Here:
//   * an 8-wave workgroup per CU, a wave owns up to two 16-row tiles and ALL NT column tiles -- 2*NT*4 accumulator
//     registers (136 at NT = 17), which with one-step-ahead fragment reads fits the 256-register budget of two waves per
//     SIMD -- so F is read exactly once.  (One wave per SIMD with four tiles, 272 accumulators in the 512-register file,
//     was tried first: hipcc keeps MFMA accumulators in AGPRs only, spilled the 16 beyond 256 and renamed them with
//     hundreds of v_accvgpr moves per chunk.)
//   * v_mfma_scale_f32_16x16x128_f8f6f4 with unit (E8M0 = 127) scales: K = 128 per instruction at twice the bf16 rate
//     per clock (MI355X_MICROARCH.md, Matrix cores).  Operands are 32 B per lane: lane (i = l & 15, g = l >> 4) holds
//     k = 32 g .. 32 g + 31 of row / column i; A and B use the same k split, so any k order inside it is consistent;
//   * the [E|Bp]^T chunk (NT*2 KB per 128 k) is staged through LDS, shared by the four waves, from the image EtS that
//     k_cast_Et8 writes in exactly the LDS order ((chunk, column tile, half, lane) x 16 B): the global -> LDS copy is a
//     straight contiguous copy and every fragment read is a conflict-free 1-KB ds_read_b128;
//   * two-stage pipeline, one barrier per chunk: chunk c+1's feature fragments and B pieces are requested before chunk
//     c's MFMAs (scheduling fences keep hipcc from sinking them), the B pieces are parked in the other LDS buffer after.
// Balanced share of the row tiles per workgroup as in v8; the body is instantiated for 1 and 2 row tiles and picked per wave.
//   * (round 3) WHOLE-LINE feature loads: a 16-row tile x 128-k chunk is 16 aligned 128-B lines of the tiled F (rows 256 B
//     apart); a load instruction takes 8 of them whole (lane l: row 8x + l/8, 16-B piece l%8) instead of half of all 16 (the
//     operand-order fragment loads of round 2 asked for every line twice, 64 B each time).  The pieces pass through a
//     WAVE-PRIVATE 2-KB LDS image per tile (piece p of row i at i*128 + ((p ^ (i&7) ^ (i>>3)) * 16): conflict-free for the
//     8-lane store groups and the 16-lane read groups alike) and come back in MFMA operand order into the same registers:
//     two ds_write_b128 + two ds_read_b128 per tile and chunk, no barrier.  NTL: the loads carry `nt` -- for tables larger
//     than the Infinity Cache (configs[4] at I = 500 K: 2 GB), which a pass must stream without evicting P and the factors;
//     a table that fits the cache (I = 50 K: 205 MB) keeps default-policy loads, the backward pass re-reads it from there.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

template <int NT, int MT, bool NTL>
__device__ __forceinline__ void f8s_body(const unsigned char *const (&arow)[2][2], const unsigned char *__restrict__ EtS,
                                         unsigned char *lds, unsigned char *myA, int nch, int cshift, int lane,
                                         float *__restrict__ P, int PS, int nrows, int first, int nstore, float ps,
                                         const int32_t *__restrict__ rows, int scatter, int nitems) {
  constexpr int BCH = NT * 2048;                        // bytes of one [E|Bp]^T chunk
  constexpr int NPIECE = NT * 128;                      // its 16-B pieces
  constexpr int NBP = (NPIECE + 511) / 512;             // pieces per thread (the last round clamps onto the last piece)
  const int tid = threadIdx.x;
  // accumulators live and die inside this instantiation (no merge of the instantiations' accumulators after the switch)
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8_t aX[MT], aY[MT];                               // 32 B per lane and row tile: first the loaded pieces, then the operand
  i32x4_t bst[NBP];
  // wave-private image: store position of (my loaded) piece l%8 of rows l/8 and 8 + l/8; read position of operand pieces 2g, 2g+1
  const int i_w = lane >> 3, p_w = lane & 7, i_r = lane & 15, g_r = lane >> 4;
  const int wo0 = i_w * 128 + ((p_w ^ (i_w & 7)) << 4);                               // row i_w       (i >> 3 == 0)
  const int wo1 = (8 + i_w) * 128 + ((p_w ^ (i_w & 7) ^ 1) << 4);                     // row 8 + i_w   (i >> 3 == 1)
  const int sw_r = (i_r & 7) ^ (i_r >> 3);
  const int ro0 = i_r * 128 + (((2 * g_r) ^ sw_r) << 4), ro1 = i_r * 128 + (((2 * g_r + 1) ^ sw_r) << 4);
#define F8S_ISSUE(c_, AR)                                                                                             \
  {                                                                                                                   \
    int ce_ = (c_) + cshift;                                                                                          \
    if (ce_ >= nch) ce_ -= nch;                                                                                       \
    const size_t ao_ = (size_t)(ce_ >> 1) * 8192 + (size_t)(ce_ & 1) * 128;                                           \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                               \
      AR[mt].lo = ld_stream16<NTL>(arow[mt][0] + ao_);                                                                \
      AR[mt].hi = ld_stream16<NTL>(arow[mt][1] + ao_);                                                                \
    }                                                                                                                 \
    const unsigned char *bc_ = EtS + (size_t)ce_ * BCH;                                                               \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) {                                                                 \
      int pc = tid + x * 512;                                                                                         \
      pc = pc < NPIECE ? pc : NPIECE - 1;                                                                             \
      bst[x] = *reinterpret_cast<const i32x4_t *>(bc_ + pc * 16);                                                     \
    }                                                                                                                 \
  }
#define F8S_PARK(buf_)                                                                                                \
  {                                                                                                                   \
    _Pragma("unroll") for (int x = 0; x < NBP; ++x) {                                                                 \
      int pc = tid + x * 512;                                                                                         \
      pc = pc < NPIECE ? pc : NPIECE - 1;                                                                             \
      *reinterpret_cast<i32x4_t *>(lds + (buf_) * BCH + pc * 16) = bst[x];                                            \
    }                                                                                                                 \
  }
#define F8S_COMPUTE(buf_, AR)                                                                                         \
  {                                                                                                                   \
    /* loaded pieces -> wave-private image -> operand order, in the same registers (LDS operations of a wave complete \
       in order: the reads see the stores; the previous chunk's reads have long been consumed) */                      \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                               \
      *reinterpret_cast<i32x4_t *>(myA + mt * 2048 + wo0) = AR[mt].lo;                                                \
      *reinterpret_cast<i32x4_t *>(myA + mt * 2048 + wo1) = AR[mt].hi;                                                \
    }                                                                                                                 \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                               \
      AR[mt].lo = *reinterpret_cast<const i32x4_t *>(myA + mt * 2048 + ro0);                                          \
      AR[mt].hi = *reinterpret_cast<const i32x4_t *>(myA + mt * 2048 + ro1);                                          \
    }                                                                                                                 \
    /* fragment nt+1 is read from LDS before the MFMAs of fragment nt; the fences keep hipcc from hoisting ALL NT     \
       fragment reads (8 registers each) above the first MFMA */                                                      \
    const unsigned char *bl_ = lds + (buf_) * BCH + lane * 16;                                                        \
    i32x8_t bq[2];                                                                                                    \
    bq[0].lo = *reinterpret_cast<const i32x4_t *>(bl_);                                                               \
    bq[0].hi = *reinterpret_cast<const i32x4_t *>(bl_ + 1024);                                                        \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                               \
      if (nt + 1 < NT) {                                                                                              \
        bq[(nt + 1) & 1].lo = *reinterpret_cast<const i32x4_t *>(bl_ + (nt + 1) * 2048);                              \
        bq[(nt + 1) & 1].hi = *reinterpret_cast<const i32x4_t *>(bl_ + (nt + 1) * 2048 + 1024);                       \
      }                                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                               \
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(AR[mt], bq[nt & 1], acc[mt][nt], 0, 0, 0,      \
                                                                       0x7f7f7f7f, 0, 0x7f7f7f7f);                    \
      __builtin_amdgcn_sched_barrier(0);                                                                              \
    }                                                                                                                 \
  }
  F8S_ISSUE(0, aX)
  F8S_PARK(0)
  __syncthreads();
  for (int c = 0; c < nch; c += 2) {                    // nch is even (D % 256 == 0); the last prefetch re-reads chunk nch-1
    F8S_ISSUE(c + 1, aY)
    __builtin_amdgcn_sched_barrier(0);
    F8S_COMPUTE(0, aX)
    F8S_PARK(1)
    __syncthreads();
    const int cn = c + 2 < nch ? c + 2 : nch - 1;
    F8S_ISSUE(cn, aX)
    __builtin_amdgcn_sched_barrier(0);
    F8S_COMPUTE(1, aY)
    F8S_PARK(0)
    __syncthreads();
  }
#undef F8S_ISSUE
#undef F8S_PARK
#undef F8S_COMPUTE
  const int r = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (mt >= nstore) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = (first + mt) * 16 + g * 4 + reg;     // C/D layout: col = lane & 15, row = (lane >> 4)*4 + reg
      if (t < nrows) {
        int orow = t;                                     // row list: row t of the result is item rows[t] (scattered to P[item])
        if (scatter) { orow = rows[t]; orow = (unsigned)orow < (unsigned)nitems ? orow : 0; }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)orow * PS + nt * 16 + r] = acc[mt][nt][reg] * ps;
      }
    }
  }
}

template <int NT, bool NTL>
__global__ __launch_bounds__(512) void k_proj_fwd_f8s(const unsigned char *__restrict__ F, int nrows, int D,
                                                      const unsigned char *__restrict__ EtS, float *__restrict__ P, int PS,
                                                      const float *__restrict__ pscale, int stagger,
                                                      const int32_t *__restrict__ rows, const int32_t *__restrict__ nrows_dev,
                                                      int scatter, int nitems, int32_t *__restrict__ errflag) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_f8s[];   // 2 x NT*2048 ([E|Bp]^T chunks) | 8 waves x 2 tiles x 2 KB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // rows != nullptr: the projection of the LISTED items (list mode of large batches: a listed item's 256-B pieces are whole
  // line pairs of the tiled F, so the gather streams like the table does); the list length comes from the device
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  if (nrows <= 0) return;                                // (workgroup-uniform)
  const int T = (nrows + 15) >> 4;
  const int t0 = (int)(((long long)blockIdx.x * T) / gridDim.x), t1 = (int)(((long long)(blockIdx.x + 1) * T) / gridDim.x);
  const int ntile = t1 - t0;                             // <= 16: the launcher sizes the grid for it
  if (ntile == 0) return;                                // (a row list shorter than its host-side bound; workgroup-uniform)
  const int base = ntile >> 3, rem = ntile & 7;
  const int nlive = base + (w < rem ? 1 : 0);            // this wave's row tiles (wave-uniform, 0..2)
  const int first = t0 + w * base + (w < rem ? w : rem);
  const unsigned char *arow[2][2];                       // [tile][rows 0-7 / rows 8-15]: my 16-B piece of my row's 128-B line
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int tile = first + mt;
    if (mt >= nlive) tile = nlive ? first + nlive - 1 : (T ? T - 1 : 0);     // never stored
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      int t = tile * 16 + 8 * x + (lane >> 3);
      if (t >= nrows) t = nrows - 1;
      if (rows) {
        t = rows[t];
        if ((unsigned)t >= (unsigned)nitems) { *errflag = 2; t = 0; }
      }
      arow[mt][x] = F + ((size_t)(t >> 5) * (size_t)(D >> 8)) * 8192 + (size_t)(t & 31) * 256 + (lane & 7) * 16;
    }
  }
  const int nch = D >> 7;
  const int cshift = (stagger & 1) ? (int)(blockIdx.x % (unsigned)nch) : 0;
  const float ps = *pscale;
  unsigned char *myA = lds_f8s + 2 * NT * 2048 + w * 4096;
  if (nlive == 2) f8s_body<NT, 2, NTL>(arow, EtS, lds_f8s, myA, nch, cshift, lane, P, PS, nrows, first, 2, ps, rows, scatter, nitems);
  else f8s_body<NT, 1, NTL>(arow, EtS, lds_f8s, myA, nch, cshift, lane, P, PS, nrows, first, nlive, ps, rows, scatter, nitems);   // 0: a spare wave
}

// dEp[k*d + n] = sum_s part[s][k][n] (n < d) ; dEp[D*d + k] = sum_s part[s][k][d]
__global__ __launch_bounds__(256) void k_reduce_parts(const float *__restrict__ part, int SK, int D, int d, int PS,
                                                      float *__restrict__ dEp, float gscale) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)D * PS) return;
  const int kk = (int)(e / PS), n = (int)(e % PS);
  if (n > d) return;
  float s = 0.f;
  for (int sidx = 0; sidx < SK; ++sidx) s += part[(size_t)sidx * D * PS + e];
  s *= gscale;
  if (n < d) dEp[(size_t)kk * d + n] = s;
  else dEp[(size_t)D * d + kk] = s;
}

// ------------------------------------------------------------------------------------------------------------
// fp32 feature path (small, reference-precision configs): fp64 accumulate on the vector ALU.
// ------------------------------------------------------------------------------------------------------------
// one wave per row; lane n owns output columns n, n+64, ...; F[t][k] is a wave-wide broadcast, E rows are coalesced.
// (nrows_dev / scatter: as k_proj_fwd_rows)
__global__ __launch_bounds__(256) void k_proj_fwd_f32(const float *__restrict__ F, const int32_t *__restrict__ rows,
                                                      int nrows, const int32_t *__restrict__ nrows_dev, int scatter, int nitems,
                                                      int D, const float *__restrict__ E, const float *__restrict__ Bp, int d,
                                                      float *__restrict__ P, int PS, int32_t *errflag) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  if (t >= nrows) return;
  int item = rows ? rows[t] : t;
  if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
  const float *f = F + (size_t)item * D;
  const size_t o = scatter ? (size_t)item : (size_t)t;
  for (int n = lane; n <= d; n += 64) {
    double acc = 0.0;
    if (n < d) for (int kk = 0; kk < D; ++kk) acc += (double)f[kk] * (double)E[(size_t)kk * d + n];
    else for (int kk = 0; kk < D; ++kk) acc += (double)f[kk] * (double)Bp[kk];
    P[o * PS + n] = (float)acc;
  }
}

// one thread per output (k, n), n <= d; W rows are coalesced over n, F[t][k] is a broadcast.
// rows != nullptr: the sum runs over the listed items only (list mode; W rows are indexed by item id in both forms)
__global__ __launch_bounds__(256) void k_proj_bwd_f32(const float *__restrict__ F, int nrows, int D,
                                                      const float *__restrict__ W, int d, int PS, float *__restrict__ dEp,
                                                      const int32_t *__restrict__ rows, const int32_t *__restrict__ nrows_dev) {
  const int kk = blockIdx.x;
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  for (int n = threadIdx.x; n <= d; n += 256) {
    double acc = 0.0;
    if (rows) {
      for (int t = 0; t < nrows; ++t) {
        const int item = rows[t];
        acc += (double)F[(size_t)item * D + kk] * (double)W[(size_t)item * PS + n];
      }
    } else {
      for (int t = 0; t < nrows; ++t) acc += (double)F[(size_t)t * D + kk] * (double)W[(size_t)t * PS + n];
    }
    if (n < d) dEp[(size_t)kk * d + n] = (float)acc;
    else dEp[(size_t)D * d + kk] = (float)acc;
  }
}

// The same two products, parallelised for what the reference's CLI defaults produce (train_rec.py:23,33-35: batch 256, fp32
// features, k = 128, d = 20 -- a few hundred listed rows, 21 output columns): the kernels above give a wave 21 busy lanes
// and a 4096-step serial loop (975 us + 158 us per step at I = 10 000, D = 4096).  Same arithmetic (every product and every
// sum in fp64), other summation order.
//   forward: a block = FT rows x 32 columns; its 8 lane groups of 32 take every 8th 4-element piece of k (F row pieces are
//   16-B broadcasts, E rows are coalesced over the columns); the 8 partial sums meet in LDS, fixed order.
constexpr int F32_RT = 2, F32_NSL = 32;                    // rows per block, k slices per block (1024 threads)
__global__ __launch_bounds__(1024) void k_proj_fwd_f32_tile(const float *__restrict__ F, const int32_t *__restrict__ rows,
                                                            int nrows, const int32_t *__restrict__ nrows_dev, int scatter,
                                                            int nitems, int D, const float *__restrict__ E,
                                                            const float *__restrict__ Bp, int d, float *__restrict__ P, int PS,
                                                            int32_t *errflag) {
  __shared__ double red[F32_NSL][F32_RT][32];
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  const int t0 = blockIdx.x * F32_RT;
  if (t0 >= nrows) return;                                  // (block-uniform)
  const int ks = threadIdx.x >> 5, n = threadIdx.x & 31, col = blockIdx.y * 32 + n;
  const float *f[F32_RT];
  int item[F32_RT];
#pragma unroll
  for (int r = 0; r < F32_RT; ++r) {
    const int t = t0 + r < nrows ? t0 + r : nrows - 1;      // (rows past the end: recomputed, not stored)
    int it = rows ? rows[t] : t;
    if ((unsigned)it >= (unsigned)nitems) { *errflag = 2; it = 0; }
    item[r] = it;
    f[r] = F + (size_t)it * D;
  }
  double acc[F32_RT];
#pragma unroll
  for (int r = 0; r < F32_RT; ++r) acc[r] = 0.0;
  const bool isE = col < d, isB = col == d;
  // a slice's k pieces: ks*4, ks*4 + 128, ...; PT pieces per trip, all loads of a trip before its arithmetic (the loop is a
  // chain of memory round trips: 8 of them at D = 4096)
  constexpr int STEP = F32_NSL * 4, PT = 4;
  for (int kb = ks * 4; kb < D; kb += PT * STEP) {          // D % 4 == 0 (launcher)
    int kq[PT];
    double wq[PT];
#pragma unroll
    for (int x = 0; x < PT; ++x) {                          // (pieces past the end: the first one again, weight 0)
      const bool in = kb + x * STEP < D;
      kq[x] = in ? kb + x * STEP : kb;
      wq[x] = in ? 1.0 : 0.0;
    }
    float e[PT][4];
    float4 v[F32_RT][PT];
#pragma unroll
    for (int x = 0; x < PT; ++x)
#pragma unroll
      for (int q = 0; q < 4; ++q) e[x][q] = isE ? E[(size_t)(kq[x] + q) * d + col] : (isB ? Bp[kq[x] + q] : 0.f);
#pragma unroll
    for (int r = 0; r < F32_RT; ++r)
#pragma unroll
      for (int x = 0; x < PT; ++x) v[r][x] = *reinterpret_cast<const float4 *>(f[r] + kq[x]);
#pragma unroll
    for (int r = 0; r < F32_RT; ++r)
#pragma unroll
      for (int x = 0; x < PT; ++x) {
        acc[r] += wq[x] * ((double)v[r][x].x * (double)e[x][0]);
        acc[r] += wq[x] * ((double)v[r][x].y * (double)e[x][1]);
        acc[r] += wq[x] * ((double)v[r][x].z * (double)e[x][2]);
        acc[r] += wq[x] * ((double)v[r][x].w * (double)e[x][3]);
      }
  }
#pragma unroll
  for (int r = 0; r < F32_RT; ++r) red[ks][r][n] = acc[r];
  __syncthreads();
  if (ks < F32_RT && (isE || isB)) {                        // lane group r sums row r
    const int r = ks;
    if (t0 + r < nrows) {
      double sum = red[0][r][n];
      for (int q = 1; q < F32_NSL; ++q) sum += red[q][r][n];
      const size_t o = scatter ? (size_t)item[r] : (size_t)(t0 + r);
      P[o * PS + col] = (float)sum;
    }
  }
}

//   backward over a row list: a block = 8 values of k x 32 columns; its 32 lane groups take every 32nd listed row (the W row
//   is coalesced over the columns, the 8 feature values are two 16-B broadcasts); partial sums meet in LDS, fixed order.
constexpr int F32_NRS = 32, F32_KV = 8;
__global__ __launch_bounds__(1024) void k_proj_bwd_f32_tile(const float *__restrict__ F, int nrows, int D,
                                                            const float *__restrict__ W, int d, int PS, float *__restrict__ dEp,
                                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ nrows_dev) {
  __shared__ double red[F32_NRS][F32_KV][32];
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  const int rs = threadIdx.x >> 5, n = threadIdx.x & 31, col = blockIdx.y * 32 + n, k0 = blockIdx.x * F32_KV;
  double acc[F32_KV];
#pragma unroll
  for (int q = 0; q < F32_KV; ++q) acc[q] = 0.0;
  const bool on = col <= d;
  // four listed rows per trip, all loads of a trip before its arithmetic (rows past the end: the last one again, weight 0)
  constexpr int RPT = 4;
  for (int t = rs; t < nrows; t += RPT * F32_NRS) {
    int item[RPT];
    double w[RPT];
    float4 v[RPT][2];
#pragma unroll
    for (int x = 0; x < RPT; ++x) {
      const int tx = t + x * F32_NRS;
      item[x] = rows ? rows[tx < nrows ? tx : t] : (tx < nrows ? tx : t);
    }
#pragma unroll
    for (int x = 0; x < RPT; ++x) {
      const float wv = on ? W[(size_t)item[x] * PS + col] : 0.f;
      w[x] = t + x * F32_NRS < nrows ? (double)wv : 0.0;
      const float *fr = F + (size_t)item[x] * D + k0;
      v[x][0] = *reinterpret_cast<const float4 *>(fr); v[x][1] = *reinterpret_cast<const float4 *>(fr + 4);
    }
#pragma unroll
    for (int x = 0; x < RPT; ++x) {
      acc[0] += (double)v[x][0].x * w[x]; acc[1] += (double)v[x][0].y * w[x]; acc[2] += (double)v[x][0].z * w[x]; acc[3] += (double)v[x][0].w * w[x];
      acc[4] += (double)v[x][1].x * w[x]; acc[5] += (double)v[x][1].y * w[x]; acc[6] += (double)v[x][1].z * w[x]; acc[7] += (double)v[x][1].w * w[x];
    }
  }
#pragma unroll
  for (int q = 0; q < F32_KV; ++q) red[rs][q][n] = acc[q];
  __syncthreads();
  if (on && rs < F32_KV) {                                  // lane group q sums k value q
    const int q = rs;
    double sum = red[0][q][n];
    for (int x = 1; x < F32_NRS; ++x) sum += red[x][q][n];
    const int kk = k0 + q;
    if (col < d) dEp[(size_t)kk * d + col] = (float)sum;
    else dEp[(size_t)D * d + kk] = (float)sum;
  }
}

// The same two products on the fp64 matrix instruction (v_mfma_f64_16x16x4_f64: products and sums in fp64, as above; what the
// vector-ALU forms spend on 8-cycle v_fma_f64 and on converting both operands of every product goes to the matrix pipe and to
// two conversions per four products).  Operand maps (cdna_hip_programming.md, fragment layout): lane l holds A[row l&15][k = l>>4]
// and B[k = l>>4][col l&15]; result register r of lane l is C[row (l>>4) + 4r][col l&15].  The instruction's four k slots are
// filled with the actual k = kb + 4*(l>>4) + s in step s (any assignment is fine as long as A and B use the same one), so a
// lane's A values of four steps are ONE 16-byte load.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
//   forward: block = 16 rows x 16 columns, its 16 waves take every 16th 16-element piece of k; partial tiles meet in LDS.
__global__ __launch_bounds__(1024) void k_proj_fwd_f32_mfma(const float *__restrict__ F, const int32_t *__restrict__ rows,
                                                            int nrows, const int32_t *__restrict__ nrows_dev, int scatter,
                                                            int nitems, int D, const float *__restrict__ E,
                                                            const float *__restrict__ Bp, int d, float *__restrict__ P, int PS,
                                                            int32_t *errflag) {
  __shared__ double red[16][4][64];
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  const int t0 = blockIdx.x * 16;
  if (t0 >= nrows) return;                                  // (block-uniform)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int t = t0 + i < nrows ? t0 + i : nrows - 1;        // (rows past the end: recomputed, not stored)
  int item = rows ? rows[t] : t;
  if ((unsigned)item >= (unsigned)nitems) { *errflag = 2; item = 0; }
  const float *fa = F + (size_t)item * D + 4 * kq;
  const int col = blockIdx.y * 16 + i;
  const bool isE = col < d, isB = col == d;
  f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int nchunk = D >> 4;                                // D % 16 == 0 (launcher)
  constexpr int PT = 4;                                     // pieces per trip: all loads of a trip before its arithmetic
  for (int c = w; c < nchunk; c += 16 * PT) {
    int kbx[PT];
    double wx[PT];
    float4 ax[PT];
    float bx[PT][4];
#pragma unroll
    for (int x = 0; x < PT; ++x) {                          // (pieces past the end: the first one again, weight 0)
      const bool in = c + 16 * x < nchunk;
      kbx[x] = (in ? c + 16 * x : c) << 4;
      wx[x] = in ? 1.0 : 0.0;
    }
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      ax[x] = *reinterpret_cast<const float4 *>(fa + kbx[x]);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int kk = kbx[x] + 4 * kq + s4;
        bx[x][s4] = isE ? E[(size_t)kk * d + col] : (isB ? Bp[kk] : 0.f);
      }
    }
#pragma unroll
    for (int x = 0; x < PT; ++x) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ax[x].x * wx[x], (double)bx[x][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ax[x].y * wx[x], (double)bx[x][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ax[x].z * wx[x], (double)bx[x][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ax[x].w * wx[x], (double)bx[x][3], acc, 0, 0, 0);
    }
  }
  red[w][0][lane] = acc.x; red[w][1][lane] = acc.y; red[w][2][lane] = acc.z; red[w][3][lane] = acc.w;
  __syncthreads();
  if (w < 4) {                                              // wave r sums result register r of the 16 partial tiles
    double sum = red[0][w][lane];
    for (int x = 1; x < 16; ++x) sum += red[x][w][lane];
    const int row = (lane >> 4) + 4 * w, oc = blockIdx.y * 16 + (lane & 15);
    if (t0 + row < nrows && oc <= d) {
      const int trow = t0 + row;
      int it = rows ? rows[trow] : trow;
      if ((unsigned)it >= (unsigned)nitems) it = 0;
      const size_t o = scatter ? (size_t)it : (size_t)trow;
      P[o * PS + oc] = (float)sum;
    }
  }
}

//   backward over a row list: block = 16 values of k x 16 columns, its 4 waves take every 4th 16-row piece of the list.
__global__ __launch_bounds__(256) void k_proj_bwd_f32_mfma(const float *__restrict__ F, int nrows, int D,
                                                           const float *__restrict__ W, int d, int PS, float *__restrict__ dEp,
                                                           const int32_t *__restrict__ rows, const int32_t *__restrict__ nrows_dev) {
  __shared__ double red[4][4][64];
  if (nrows_dev) { const int n = *nrows_dev; nrows = n < nrows ? n : nrows; }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int k0 = blockIdx.x * 16, col = blockIdx.y * 16 + i;
  const bool on = col <= d;
  f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int nchunk = (nrows + 15) >> 4;
  for (int c = w; c < nchunk; c += 4) {
    const int tb = (c << 4) + 4 * kq;
    int it[4];
    float wv[4], fv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int t = tb + s4;
      it[s4] = rows ? rows[t < nrows ? t : nrows - 1] : (t < nrows ? t : nrows - 1);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      fv[s4] = F[(size_t)it[s4] * D + k0 + i];
      wv[s4] = (on && tb + s4 < nrows) ? W[(size_t)it[s4] * PS + col] : 0.f;     // (rows past the end: weight 0)
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)fv[s4], (double)wv[s4], acc, 0, 0, 0);
  }
  red[w][0][lane] = acc.x; red[w][1][lane] = acc.y; red[w][2][lane] = acc.z; red[w][3][lane] = acc.w;
  __syncthreads();
  {                                                         // wave r sums result register r of the 4 partial tiles
    const double sum = (red[0][w][lane] + red[1][w][lane]) + (red[2][w][lane] + red[3][w][lane]);
    const int kk = k0 + (lane >> 4) + 4 * w, oc = blockIdx.y * 16 + (lane & 15);
    if (oc < d) dEp[(size_t)kk * d + oc] = (float)sum;
    else if (oc == d) dEp[(size_t)D * d + kk] = (float)sum;
  }
}

extern "C" int bprx_variant_safe(int ver, int nt, int mt, int rem);   // generated at build time (build.py)

// Deq: row width in bf16-sized units (fp8 rows are addressed as bf16 rows of half the width)
#define FWD_ARGS (const uint16_t *)h->Ft, rows, (int)nrows, h->cfg.num_items, Deq, (const uint16_t *)h->Et, Pout, h->PS, h->errflag
template <int NT>
void launch_v9(bprx_handle *h, const int32_t *rows, int64_t nrows, float *Pout, hipStream_t s, int Deq, bool f8,
               const float *pscale, int stagger, int n0) {
  const int64_t T = (nrows + 15) / 16;
  const int ncu = h->num_cu > 0 ? h->num_cu : 256;
  int64_t G = (T + 15) / 16;
  if (G < ncu) G = T < ncu ? T : ncu;
  else G = (G + ncu - 1) / ncu * ncu;
  const int tpw_max = (int)((T + G - 1) / G);
  const int per_wave = tpw_max > 8 ? 2 : 1;
  int nw = (tpw_max + per_wave - 1) / per_wave;
  if (nw < 5) nw = 5;
  const size_t lds = (size_t)(256 + NT * 16) * 288;
  if (f8) {
    auto kfn = k_proj_fwd_bf16_v9<NT, true>;
    (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(nw * 64), lds, s, FWD_ARGS, stagger, pscale, per_wave, n0);
  } else {
    auto kfn = k_proj_fwd_bf16_v9<NT, false>;
    (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(nw * 64), lds, s, FWD_ARGS, stagger, pscale, per_wave, n0);
  }
}

template <int NT>
void launch_v8(bprx_handle *h, const int32_t *rows, int64_t nrows, float *Pout, hipStream_t s, int Deq, bool f8,
               const float *pscale, int stagger, int n0) {
  constexpr int NWMAX = 8, NWMIN = NT <= 7 ? 5 : 8;   // must match the kernel's NWMIN
  const int64_t T = (nrows + 15) / 16;
  const int ncu = h->num_cu > 0 ? h->num_cu : 256;
  int64_t G = (T + 2 * NWMAX - 1) / (2 * NWMAX);
  if (G < ncu) G = T < ncu ? T : ncu;
  else G = (G + ncu - 1) / ncu * ncu;
  const int tpw_max = (int)((T + G - 1) / G);                 // most tiles any workgroup owns (<= 2 NWMAX)
  const int per_wave = tpw_max > NWMAX ? 2 : 1;
  int nw = (tpw_max + per_wave - 1) / per_wave;
  if (nw < NWMIN) nw = NWMIN;
  if (f8) hipLaunchKernelGGL((k_proj_fwd_bf16_v8<NT, NWMAX, true>), dim3((unsigned)G), dim3(nw * 64), 0, s, FWD_ARGS, stagger, pscale, per_wave, n0);
  else hipLaunchKernelGGL((k_proj_fwd_bf16_v8<NT, NWMAX, false>), dim3((unsigned)G), dim3(nw * 64), 0, s, FWD_ARGS, stagger, pscale, per_wave, n0);
}

void launch_v8_rt(int nt, bprx_handle *h, const int32_t *rows, int64_t nrows, float *Pout, hipStream_t s, int Deq, bool f8,
                  const float *pscale, int stagger, int n0) {
  switch (nt) {
    case 1: launch_v8<1>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 2: launch_v8<2>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 3: launch_v8<3>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 4: launch_v8<4>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 5: launch_v8<5>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 6: launch_v8<6>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 7: launch_v8<7>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    case 8: launch_v8<8>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
    default: launch_v8<9>(h, rows, nrows, Pout, s, Deq, f8, pscale, stagger, n0); break;
  }
}

// row-list forward (k_proj_fwd_rows): column split / row tiles per workgroup by launch size and register budget
template <int NT>
void launch_fwd_rows(bprx_handle *h, const int32_t *rows, int64_t nrows, const int32_t *nrows_dev, int scatter, float *Pout,
                     hipStream_t s) {
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const int Deq = f8 ? h->cfg.feat_dim / 2 : h->cfg.feat_dim;
  const float *pscale = h->qs + 1;
  const int64_t tiles = (nrows + 15) / 16;
  const int ncu = h->num_cu > 0 ? h->num_cu : 256;
  int mt = 1;
  if (NT <= 5 && tiles > 8 * (int64_t)ncu) mt = 4;
  else if (NT <= 9 && tiles > 2 * (int64_t)ncu) mt = 2;
  if (const char *e = getenv("BPRX_ROWS_MT")) { const int v = atoi(e); if (v == 1 || (v == 2 && NT <= 9) || (v == 4 && NT <= 5)) mt = v; }
  // tiny launches (row tiles x column tiles fit the chip twice): one column tile per workgroup, 16 waves split K
  bool split = NT > 1 && tiles * NT <= 2 * (int64_t)ncu;
  if (const char *e = getenv("BPRX_ROWS_SPLIT")) split = NT > 1 && atoi(e) != 0;
#define ROWS_LAUNCH(NTW_, MT_, NW_, F8_, GY_)                                                                            \
  do {                                                                                                                   \
    auto kfn = k_proj_fwd_rows<NTW_, MT_, NW_, F8_>;                                                                     \
    const size_t lds = (size_t)NTW_ * 2048 * NW_ / 4;                                                                    \
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, dim3((unsigned)((tiles + MT_ - 1) / MT_), (unsigned)(GY_)), dim3(NW_ * 64), lds, s,          \
                       (const uint16_t *)h->Ft, rows, (int)nrows, nrows_dev, h->cfg.num_items, Deq,                      \
                       (const uint16_t *)h->EtF, Pout, h->PS, h->errflag, pscale, scatter);                              \
  } while (0)
  if (split) { if (f8) ROWS_LAUNCH(1, 1, 16, true, NT); else ROWS_LAUNCH(1, 1, 16, false, NT); return; }
  if constexpr (NT <= 5) {
    if (mt == 4) { if (f8) ROWS_LAUNCH(NT, 4, 8, true, 1); else ROWS_LAUNCH(NT, 4, 8, false, 1); return; }
  }
  if constexpr (NT <= 9) {
    if (mt == 2) { if (f8) ROWS_LAUNCH(NT, 2, 8, true, 1); else ROWS_LAUNCH(NT, 2, 8, false, 1); return; }
  }
  if (f8) ROWS_LAUNCH(NT, 1, 8, true, 1); else ROWS_LAUNCH(NT, 1, 8, false, 1);
#undef ROWS_LAUNCH
}

// fp8 features, wide projection (NT >= 10): ONE pass on the block-scaled fp8 MFMA (k_proj_fwd_f8s), over the whole table or a
// row list (nrows = the host-side bound of the list, the length is read on the device)
template <int NT>
void launch_f8s(bprx_handle *h, const int32_t *rows, int64_t nrows, const int32_t *nrows_dev, int scatter, float *Pout, hipStream_t s,
                int stagger) {
  const float *pscale = h->qs + 1;
  const int64_t T = (nrows + 15) / 16;
  const int ncu = h->num_cu > 0 ? h->num_cu : 256;
  int64_t G = (T + 15) / 16;                              // at most 16 row tiles per workgroup (8 waves x 2)
  if (G < ncu) G = T < ncu ? T : ncu;
  else G = (G + ncu - 1) / ncu * ncu;
  const size_t lds = (size_t)2 * NT * 2048 + 8 * 4096;
  // streaming (`nt`) loads when the table cannot stay in the 256-MiB Infinity Cache between the two passes of a step
  const bool ntl = (size_t)h->cfg.num_items * h->cfg.feat_dim > ((size_t)192 << 20);
#define F8S_LAUNCH(NTL_)                                                                                                  \
  do {                                                                                                                    \
    auto kfn = k_proj_fwd_f8s<NT, NTL_>;                                                                                  \
    (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(512), lds, s, (const unsigned char *)h->Ft, (int)nrows, h->cfg.feat_dim,  \
                       (const unsigned char *)h->EtS, Pout, h->PS, pscale, stagger, rows, nrows_dev, scatter, h->cfg.num_items, \
                       h->errflag);                                                                                        \
  } while (0)
  if (ntl) F8S_LAUNCH(true); else F8S_LAUNCH(false);
#undef F8S_LAUNCH
}

template <int NT>
int launch_fwd_nt(bprx_handle *h, const int32_t *rows, int64_t nrows, float *Pout, hipStream_t s) {
  constexpr int MTD = NT <= 9 ? 2 : 1;
  // fwd_variant & 7: 0 = v1 (2 barriers per chunk, nothing overlapped), 1 = v1 with one row tile per wave,
  //              2 = v6 (asm-pinned ping-pong pipeline, 4-wave workgroups of 128 items), 3 = v6 with one row tile per wave,
  //              4 = v8 (default: one balanced workgroup per CU), 5 = v9 (A operand through LDS; experiment);
  //              +8 = staggered chunk order; +16 / +32 / +64 = timing-only ablations of v1 / v6 (wrong results)
  const int v = h->fwd_variant & 7, stg = (h->fwd_variant >> 3);
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const int Deq = f8 ? h->cfg.feat_dim / 2 : h->cfg.feat_dim;
  const float *pscale = h->qs + 1;
  const int MT = (v == 1 || v == 3) ? 1 : MTD;
  dim3 grid((unsigned)((nrows + 4 * MT * 16 - 1) / (4 * MT * 16)));
  // the pipelined kernel only where the build verified it spill-free (build.py), and D must hold an even chunk count
  const bool pipe = !f8 && (v == 2 || v == 3 || v == 4) && Deq % 256 == 0 && bprx_variant_safe(6, NT, MT, 0);
  if constexpr (NT == 5) {
    static const int abl = getenv("BPRX_FWD_ABL") ? atoi(getenv("BPRX_FWD_ABL")) : 0;
    if (pipe && MT != 1 && abl) {
      switch (abl) {
        case 1: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 1>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 2: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 2>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 3: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 3>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 4: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 4>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 6: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 6>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 7: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 7>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 14: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 14>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        case 15: hipLaunchKernelGGL((k_proj_fwd_bf16_v6<5, 2, 15>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale); return 0;
        default: break;
      }
    }
  }
  if constexpr (NT <= 9) {
    if (v == 5 && Deq % 256 == 0) {                      // v9: A operand through LDS, contiguous loads of the tiled F
      launch_v9<NT>(h, rows, nrows, Pout, s, Deq, f8, pscale, stg & 1, 0);
      return 0;
    }
  }
  if constexpr (NT >= 10) {
    // fp8 features, wide projection: ONE pass on the block-scaled fp8 MFMA (k_proj_fwd_f8s); BPRX_F8S=0 keeps the
    // column-range passes of v8 for A/B measurements
    const int f8s_on = getenv("BPRX_F8S") ? atoi(getenv("BPRX_F8S")) : 1;
    if (f8 && v == 4 && f8s_on && h->EtS && h->cfg.feat_dim % 256 == 0) {
      launch_f8s<NT>(h, nullptr, nrows, nullptr, 0, Pout, s, stg & 1);
      return 0;
    }
  }
  if constexpr (NT <= 9) {
    // v10 (A through a wave-private LDS image, contiguous loads, `nt`): the default for bf16 tables up to nine column tiles
    // -- C2: 76.7-80.2 us against v8's 83.4-87.2 us on the same boxes; with streaming (`nt`) loads 76.8-77.2 us and the
    // whole step 0.2596 -> 0.2436 ms; c4 shard (nine tiles): 116.5 us against v8's 118.8-125.7 us, step 0.3587 -> 0.3511.
    // Not for fp8 tables (c2fp8: 45.0 vs 41.7 us).  BPRX_FWD_LDS=0 keeps v8; variant 6 / 14 forces v10 (+ staggered chunks).
    const bool lds_default = v == 4 && !f8 && NT <= 9 && !(getenv("BPRX_FWD_LDS") && atoi(getenv("BPRX_FWD_LDS")) == 0);
    const int stg10 = v == 6 ? (stg & 1) : 0;
    if ((v == 6 || lds_default) && Deq % 256 == 0 && !rows) {
      constexpr int NWMAX = 8, NWMIN = 5;
      const int64_t T = (nrows + 15) / 16;
      const int ncu = h->num_cu > 0 ? h->num_cu : 256;
      int64_t G = (T + 2 * NWMAX - 1) / (2 * NWMAX);
      if (G < ncu) G = T < ncu ? T : ncu;
      else G = (G + ncu - 1) / ncu * ncu;
      const int tpw_max = (int)((T + G - 1) / G);
      const int per_wave = tpw_max > NWMAX ? 2 : 1;
      int nw = (tpw_max + per_wave - 1) / per_wave;
      if (nw < NWMIN) nw = NWMIN;
      const size_t lds = (size_t)2 * NT * 16 * 288 + (size_t)nw * 2 * 16 * 288;
      if (f8) {
        auto kfn = k_proj_fwd_bf16_v10<NT, true>;
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(nw * 64), lds, s, (const uint16_t *)h->Ft, (int)nrows, Deq,
                           (const uint16_t *)h->Et, Pout, h->PS, pscale, stg10, per_wave);
      } else {
        auto kfn = k_proj_fwd_bf16_v10<NT, false>;
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(nw * 64), lds, s, (const uint16_t *)h->Ft, (int)nrows, Deq,
                           (const uint16_t *)h->Et, Pout, h->PS, pscale, stg10, per_wave);
      }
      return 0;
    }
  }
  if (v == 4 && Deq % 256 == 0) {
    // v8 (one balanced workgroup per CU).  Projections wider than 9 column tiles (d > 143) are covered by two launches
    // over column ranges (F is read twice: still less time than one pass of the plain kernel).
    // Column ranges of the widest spill-free width (<= 9 tiles; normally the whole projection in one launch), the last
    // one right-aligned: ranges may overlap by some tiles, which are then computed and stored twice, identically.
    int wmax = NT < 9 ? NT : 9;
    while (wmax >= 1 && !bprx_variant_safe(8, wmax, 8, f8 ? 1 : 0)) --wmax;
    if (wmax >= (NT < 5 ? NT : 5)) {
      for (int c0 = 0; c0 < NT; c0 += wmax) {
        const int start = c0 + wmax <= NT ? c0 : NT - wmax;
        launch_v8_rt(wmax, h, rows, nrows, Pout, s, Deq, f8, pscale, stg & 1, start * 16);
      }
      return 0;
    }
  }
  if (pipe && MT == 1) hipLaunchKernelGGL((k_proj_fwd_bf16_v6<NT, 1>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale);
  else if (pipe) hipLaunchKernelGGL((k_proj_fwd_bf16_v6<NT, MTD>), grid, dim3(256), 0, s, FWD_ARGS, stg & 15, pscale);
  else if (f8) hipLaunchKernelGGL((k_proj_fwd_bf16<NT, MTD, true>), grid, dim3(256), 0, s, FWD_ARGS, stg & 1, pscale);
  else if (MT == 1) hipLaunchKernelGGL((k_proj_fwd_bf16<NT, 1, false>), grid, dim3(256), 0, s, FWD_ARGS, stg, pscale);
  else hipLaunchKernelGGL((k_proj_fwd_bf16<NT, MTD, false>), grid, dim3(256), 0, s, FWD_ARGS, stg, pscale);
  return 0;
}

// one launch of k_proj_bwd_bf16_v3 with its dynamic LDS size (DB images of F tile + W tile)
template <int NT, int BTV, int NW, int PD, bool F8, bool ROWS, int DB>
void launch_bwd3(dim3 grid, hipStream_t s, const uint16_t *Ft, int nrows, int D, const uint16_t *Wb, int PS, float *part, int rps,
                 int desc, int xmap, const int32_t *rows, const int32_t *nrows_dev, const float *Wf) {
  constexpr size_t lds = (size_t)DB * (BTV * (NW * 32 * 2 + 32) + BTV * WsStride3<NT>::bytes);
  auto kfn = k_proj_bwd_bf16_v3<NT, BTV, NW, PD, F8, ROWS, DB>;
  if (lds > 48 * 1024) {
    static bool attr_set = false;                        // per instantiation
    if (!attr_set) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; }
  }
  hipLaunchKernelGGL(kfn, grid, dim3(NW * 64), lds, s, Ft, nrows, D, Wb, PS, part, rps, desc, xmap, rows, nrows_dev, Wf);
}

// backward over the touched-item list (list mode): v3 kernel in ROWS form, 2 tiles in flight; `bound` = host-side bound of
// the list length
template <int NT>
int launch_bwd_rows(bprx_handle *h, int64_t bound, hipStream_t s) {
  const int D = h->cfg.feat_dim;
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const bool w8 = D % 256 == 0 || f8;
  dim3 g3(D / (w8 ? 256 : 128), h->SK_step);
#define BWDR_LAUNCH(NW_, F8_)                                                                                            \
  launch_bwd3<NT, 32, NW_, 2, F8_, true, 1>(g3, s, (const uint16_t *)h->Ft, (int)bound, D, (const uint16_t *)h->Wb, h->PS,     \
                                            h->part, 0, 0, 1, (const int32_t *)h->ilist, (const int32_t *)h->list_cur,        \
                                            (const float *)h->W)
  if (f8) BWDR_LAUNCH(8, true);
  else if (w8) BWDR_LAUNCH(8, false);
  else BWDR_LAUNCH(4, false);
#undef BWDR_LAUNCH
  return 0;
}

template <int NT>
int launch_bwd_nt(bprx_handle *h, hipStream_t s) {
  const int D = h->cfg.feat_dim, I = h->cfg.num_items;
  // bwd_variant: 0 = v1 (fp32 W, no prefetch); 8 = v3 (bf16 W, conflict-free LDS image, next tile prefetched),
  //              9 = v3 with 64-item tiles; +2 = 8 waves / 256 columns per workgroup; +4 = descending item order
  const int desc = (h->bwd_variant >> 2) & 1;
  if (h->bwd_variant >= 8) {
    const int bt3 = (h->bwd_variant & 1) ? 64 : 32;
    int rps3 = (I + h->SK - 1) / h->SK;
    rps3 = (rps3 + bt3 - 1) / bt3 * bt3;
    const size_t n4 = (size_t)I * h->PS / 4;
    if (!h->item_mode)   // k_item_seg has already written Wb (bf16) itself
      hipLaunchKernelGGL(k_cast_W, dim3(1024), dim3(256), 0, s, h->W, (uint16_t *)h->Wb, n4);
    // bwd_variant & 2: 8 waves / 256 columns per workgroup (needs D % 256 == 0); bits 4-5: tiles in flight - 1
    const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
    const bool w8 = ((h->bwd_variant & 2) && D % 256 == 0) || f8;   // fp8: a 256-column tile row is one 256-B block row
    const int pd = ((h->bwd_variant >> 4) & 3) + 1;
    const int xmap = (h->bwd_variant & 64) ? 0 : 1;   // +64: plain blockIdx mapping (A/B)
    dim3 g3(D / (w8 ? 256 : 128), h->SK);
    // double-buffered LDS image (one barrier per tile) for the MFMA-paced shapes: fp8 tables and projections wider than nine
    // column tiles, two tiles in flight; BPRX_BWD_DB = 0 / 1 forces it off / on wherever it is instantiated
    bool db = (f8 || NT > 9) && pd == 2 && bt3 == 32 && w8;
    if (const char *e = getenv("BPRX_BWD_DB")) db = atoi(e) != 0 && pd == 2 && bt3 == 32 && w8;
#define BWD3_ARGS (const uint16_t *)h->Ft, I, D, (const uint16_t *)h->Wb, h->PS, h->part, rps3, desc, xmap, (const int32_t *)nullptr, \
                  (const int32_t *)nullptr, (const float *)nullptr
#define BWD3_LAUNCH(BTV_, NW_, PD_)                                                                                      \
  do {                                                                                                                   \
    if constexpr (NW_ == 8 && PD_ == 2 && BTV_ == 32) {                                                                  \
      if (db) {                                                                                                          \
        if (f8) launch_bwd3<NT, 32, 8, 2, true, false, 2>(g3, s, BWD3_ARGS);                                             \
        else launch_bwd3<NT, 32, 8, 2, false, false, 2>(g3, s, BWD3_ARGS);                                               \
        break;                                                                                                           \
      }                                                                                                                  \
    }                                                                                                                    \
    if constexpr (NW_ == 8) {                                                                                            \
      if (f8) {                                                                                                          \
        launch_bwd3<NT, BTV_, 8, PD_, true, false, 1>(g3, s, BWD3_ARGS);                                                 \
        break;                                                                                                           \
      }                                                                                                                  \
    }                                                                                                                    \
    launch_bwd3<NT, BTV_, NW_, PD_, false, false, 1>(g3, s, BWD3_ARGS);                                                  \
  } while (0)
#define BWD3_PD(BTV_, NW_)                                        \
  switch (pd) {                                                   \
    case 1: BWD3_LAUNCH(BTV_, NW_, 1); break;                     \
    case 2: BWD3_LAUNCH(BTV_, NW_, 2); break;                     \
    case 3: BWD3_LAUNCH(BTV_, NW_, 3); break;                     \
    default: BWD3_LAUNCH(BTV_, NW_, 4); break;                    \
  }
    if (bt3 == 32 && w8) { BWD3_PD(32, 8) }
    else if (bt3 == 32) { BWD3_PD(32, 4) }
    else if (w8) { BWD3_LAUNCH(64, 8, 1); }
    else { BWD3_LAUNCH(64, 4, 1); }
#undef BWD3_PD
#undef BWD3_LAUNCH
#undef BWD3_ARGS
    return 0;
  }
  int rps = (I + h->SK - 1) / h->SK;
  rps = (rps + BT - 1) / BT * BT;
  dim3 grid(D / 128, h->SK);
  hipLaunchKernelGGL((k_proj_bwd_bf16<NT>), grid, dim3(256), 0, s, (const uint16_t *)h->Ft, I, D, h->W, h->PS, h->part, rps);
  return 0;
}

#define NT_SWITCH(NT, CALL)            \
  switch (NT) {                        \
    case 1: CALL(1); break;            \
    case 2: CALL(2); break;            \
    case 3: CALL(3); break;            \
    case 4: CALL(4); break;            \
    case 5: CALL(5); break;            \
    case 6: CALL(6); break;            \
    case 7: CALL(7); break;            \
    case 8: CALL(8); break;            \
    case 9: CALL(9); break;            \
    case 10: CALL(10); break;          \
    case 11: CALL(11); break;          \
    case 12: CALL(12); break;          \
    case 13: CALL(13); break;          \
    case 14: CALL(14); break;          \
    case 15: CALL(15); break;          \
    case 16: CALL(16); break;          \
    default: CALL(17); break;          \
  }

}  // namespace

// builds the tiled copy of F (see k_tile_F); called from bprx_bind_tables, synchronous
int bprx_launch_tile_F(bprx_handle *h) {
  if (h->cfg.model != BPRX_MODEL_VBPR || h->cfg.feat_dtype == BPRX_F_FP32) return BPRX_OK;
  const int I = h->cfg.num_items, Deq = h->cfg.feat_dtype == BPRX_F_FP8 ? h->cfg.feat_dim / 2 : h->cfg.feat_dim;
  dim3 grid((unsigned)(Deq / 128), (unsigned)((I + 31) / 32));
  hipLaunchKernelGGL(k_tile_F, grid, dim3(256), 0, 0, (const uint16_t *)h->t.F, (uint16_t *)h->Ft, I, Deq);
  BPRX_LAUNCH_CHECK(h, "k_tile_F");
  BPRX_HIP(h, hipStreamSynchronize(0));
  return BPRX_OK;
}

int bprx_launch_cast_Et(bprx_handle *h, hipStream_t s) {
  if (h->cfg.feat_dtype == BPRX_F_FP32) return BPRX_OK;
  if (h->et_valid) return BPRX_OK;                      // E / Bp unchanged since the image was made
  h->et_valid = true;
  const int D = h->cfg.feat_dim;
  BprxProfScope ps(h, BPRX_PHASE_CAST_ET, s);
  dim3 grid((D + 63) / 64, h->PS / 16);
  if (h->cfg.feat_dtype == BPRX_F_FP8) {
    const int slot = h->qs_slot;                            // max|E,Bp| sits / accumulates in qs[2 + slot]
    h->qs_slot ^= 1;
    if (!h->absmax_valid) {                                 // not left there by the last k_dense_update (first step, outside write)
      BPRX_HIP(h, hipMemsetAsync((uint32_t *)h->qs + 2 + slot, 0, sizeof(uint32_t), s));
      hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, s, h->t.E, (size_t)D * h->cfg.embed_d, h->t.Bp, (size_t)D,
                         (uint32_t *)h->qs + 2 + slot);
    }
    h->absmax_valid = false;
    hipLaunchKernelGGL(k_cast_Et8, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, s, h->t.E, h->t.Bp, (uint8_t *)h->Et,
                       (uint8_t *)h->EtF, (uint8_t *)h->EtS, D, h->cfg.embed_d, h->PS, h->qs, h->cfg.feat_scale, slot);
    BPRX_LAUNCH_CHECK(h, "k_cast_Et8");
    return BPRX_OK;
  }
  hipLaunchKernelGGL(k_cast_Et, grid, dim3(256), 0, s, h->t.E, h->t.Bp, (uint16_t *)h->Et, (uint16_t *)h->EtF, D,
                     h->cfg.embed_d, h->PS);
  BPRX_LAUNCH_CHECK(h, "k_cast_Et");
  return BPRX_OK;
}

int bprx_launch_proj_fwd(bprx_handle *h, const int32_t *rows, int64_t nrows, const int32_t *nrows_dev, int scatter, float *Pout,
                         hipStream_t s) {
  if (nrows <= 0) return BPRX_OK;
  BprxProfScope ps(h, BPRX_PHASE_PROJ_FWD, s);
  if (h->cfg.feat_dtype != BPRX_F_FP32) {
    const int NT = h->PS / 16;
    if (rows && NT >= 10 && h->cfg.feat_dtype == BPRX_F_FP8 && h->EtS && h->cfg.feat_dim % 256 == 0 && nrows >= 4096 &&
        !(getenv("BPRX_F8S") && atoi(getenv("BPRX_F8S")) == 0)) {
      // a LARGE row list of a wide fp8 projection (list mode at configs[4] scale: ~123 K distinct items per batch of 65 536):
      // the one-pass streaming kernel gathers the listed rows (k_proj_fwd_rows is built for lists of a few hundred rows)
      switch (NT) {
        case 10: launch_f8s<10>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 11: launch_f8s<11>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 12: launch_f8s<12>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 13: launch_f8s<13>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 14: launch_f8s<14>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 15: launch_f8s<15>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        case 16: launch_f8s<16>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
        default: launch_f8s<17>(h, rows, nrows, nrows_dev, scatter, Pout, s, 0); break;
      }
      BPRX_LAUNCH_CHECK(h, "k_proj_fwd_f8s<rows>");
      return BPRX_OK;
    }
    if (rows) {                                          // row list: K split over the waves of one workgroup per 16-64 rows
#define CALL(N) launch_fwd_rows<N>(h, rows, nrows, nrows_dev, scatter, Pout, s)
      NT_SWITCH(NT, CALL)
#undef CALL
      BPRX_LAUNCH_CHECK(h, "k_proj_fwd_rows");
      return BPRX_OK;
    }
#define CALL(N) launch_fwd_nt<N>(h, rows, nrows, Pout, s)
    NT_SWITCH(NT, CALL)
#undef CALL
    BPRX_LAUNCH_CHECK(h, "k_proj_fwd_bf16");
  } else {
    static const int f32_tile = getenv("BPRX_F32_TILE") ? atoi(getenv("BPRX_F32_TILE")) : 2;
    if (f32_tile >= 2 && h->cfg.feat_dim % 16 == 0) {     // (2, the default: fp64 matrix instruction; 1: vector-ALU tiles; 0: first form)
      dim3 grid((unsigned)((nrows + 15) / 16), (unsigned)((h->cfg.embed_d + 1 + 15) / 16));
      hipLaunchKernelGGL(k_proj_fwd_f32_mfma, grid, dim3(1024), 0, s, (const float *)h->t.F, rows, (int)nrows, nrows_dev, scatter,
                         h->cfg.num_items, h->cfg.feat_dim, h->t.E, h->t.Bp, h->cfg.embed_d, Pout, h->PS, h->errflag);
    } else if (f32_tile && h->cfg.feat_dim % 4 == 0) {
      dim3 grid((unsigned)((nrows + F32_RT - 1) / F32_RT), (unsigned)((h->cfg.embed_d + 1 + 31) / 32));
      hipLaunchKernelGGL(k_proj_fwd_f32_tile, grid, dim3(F32_NSL * 32), 0, s, (const float *)h->t.F, rows, (int)nrows, nrows_dev, scatter,
                         h->cfg.num_items, h->cfg.feat_dim, h->t.E, h->t.Bp, h->cfg.embed_d, Pout, h->PS, h->errflag);
    } else {
      dim3 grid((unsigned)((nrows + 3) / 4));
      hipLaunchKernelGGL(k_proj_fwd_f32, grid, dim3(256), 0, s, (const float *)h->t.F, rows, (int)nrows, nrows_dev, scatter,
                         h->cfg.num_items, h->cfg.feat_dim, h->t.E, h->t.Bp, h->cfg.embed_d, Pout, h->PS, h->errflag);
    }
    BPRX_LAUNCH_CHECK(h, "k_proj_fwd_f32");
  }
  return BPRX_OK;
}

// B: batch size of the step (bounds the touched-item list in list mode)
int bprx_launch_proj_bwd(bprx_handle *h, int64_t B, hipStream_t s) {
  const int D = h->cfg.feat_dim, d = h->cfg.embed_d, I = h->cfg.num_items;
  const int64_t bound = 2 * B < (int64_t)I ? 2 * B : (int64_t)I;     // list mode: at most 2B distinct items
  h->SK_step = h->SK;
  if (h->cfg.feat_dtype != BPRX_F_FP32) {
    const int NT = h->PS / 16;
    if (h->list_mode) {
      // few rows: fewer item splits (each split writes a D x PS fp32 slab that the dense update reads back)
      int sk = (int)((bound + 127) / 128);
      h->SK_step = sk < 1 ? 1 : (sk > h->SK ? h->SK : sk);
      {
        BprxProfScope ps(h, BPRX_PHASE_PROJ_BWD, s);
#define CALL(N) launch_bwd_rows<N>(h, bound, s)
        NT_SWITCH(NT, CALL)
#undef CALL
      }
      BPRX_LAUNCH_CHECK(h, "k_proj_bwd_bf16_v3<rows>");
    } else {
      BprxProfScope ps(h, BPRX_PHASE_PROJ_BWD, s);
#define CALL(N) launch_bwd_nt<N>(h, s)
      NT_SWITCH(NT, CALL)
#undef CALL
      BPRX_LAUNCH_CHECK(h, "k_proj_bwd_bf16");
    }
    if (h->fused_reduce) return BPRX_OK;              // k_dense_update sums the slabs (bprx_step, bf16 path)
    const size_t n = (size_t)D * h->PS;
    BprxProfScope ps(h, BPRX_PHASE_REDUCE, s);
    hipLaunchKernelGGL(k_reduce_parts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->part, h->SK_step, D, d, h->PS, h->dEp,
                       h->cfg.feat_dtype == BPRX_F_FP8 ? 1.0f / h->cfg.feat_scale : 1.0f);
    BPRX_LAUNCH_CHECK(h, "k_reduce_parts");
  } else {
    {
      BprxProfScope ps(h, BPRX_PHASE_PROJ_BWD, s);
      static const int f32_tile = getenv("BPRX_F32_TILE") ? atoi(getenv("BPRX_F32_TILE")) : 2;
      // (the tiled forms split the listed rows over the lane groups / waves of a block: for a row LIST; the whole-table sum
      //  keeps one block per k)
      // (the fp64-matrix form of the backward sum is no faster than the vector-ALU tiles here -- 19 vs 17 us: BPRX_F32_TILE=3)
      if (f32_tile >= 3 && h->list_mode && D % 16 == 0)
        hipLaunchKernelGGL(k_proj_bwd_f32_mfma, dim3(D / 16, (unsigned)((d + 1 + 15) / 16)), dim3(256), 0, s, (const float *)h->t.F,
                           (int)bound, D, h->W, d, h->PS, h->dEp, (const int32_t *)h->ilist, (const int32_t *)h->list_cur);
      else if (f32_tile && h->list_mode && D % F32_KV == 0)
        hipLaunchKernelGGL(k_proj_bwd_f32_tile, dim3(D / F32_KV, (unsigned)((d + 1 + 31) / 32)), dim3(F32_NRS * 32), 0, s, (const float *)h->t.F,
                           (int)bound, D, h->W, d, h->PS, h->dEp, (const int32_t *)h->ilist, (const int32_t *)h->list_cur);
      else
      hipLaunchKernelGGL(k_proj_bwd_f32, dim3(D), dim3(256), 0, s, (const float *)h->t.F, h->list_mode ? (int)bound : I, D, h->W, d,
                         h->PS, h->dEp, h->list_mode ? (const int32_t *)h->ilist : (const int32_t *)nullptr,
                         h->list_mode ? (const int32_t *)h->list_cur : (const int32_t *)nullptr);
    }
    BPRX_LAUNCH_CHECK(h, "k_proj_bwd_f32");
  }
  return BPRX_OK;
}
