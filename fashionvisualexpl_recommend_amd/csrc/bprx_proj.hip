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
  constexpr bool dbg_skip_b = false, dbg_skip_a = false;
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
// the reduction index contiguous per lane, so 32-item tiles of F and W go through LDS and are read back with the
// hardware transpose read ds_read_b64_tr_b16 (k_proj_bwd_bf16_v3 below).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x4 lds_tr16(const uint16_t *p) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
  return __builtin_bit_cast(bf16x4, v);
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
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void fp8x4_to_bf16x4(uint32_t w, uint32_t &o0, uint32_t &o1) {
  // v_cvt_scalef32_pk_bf16_fp8 (gfx950): two codes -> two packed bf16 per instruction (scale 1.0: exact), instead of
  // v_cvt_pk_f32_fp8 + a shift and an and-or per pair -- 2 VALU instructions per four codes instead of 6
  o0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
  o1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true));
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
// NS (wave grid): the NW waves form (NW / NS) x NS; a wave owns 2*NS of the workgroup's feature-column tiles and ceil(NT / NS)
// of the projection's column tiles.  Per 32 items a wave reads (2*NS + NT/NS) KB of fragments from LDS for 2*NT MFMAs; NS = 1
// at NT = 17 is 19 KB for 34 -- the transpose reads, 64 B/clk per CU, take 2.2x the MFMAs' time and pace the kernel (c5: 1 024 us
// measured, 990 us by that count); NS = 2 is 13 KB for 36.
template <int NT, int BTV, int NW, int PD, bool F8, bool ROWS = false, int DB = 1, int NS = 1>
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
  static_assert(NW % NS == 0, "wave grid");
  constexpr int MTW = 2 * NS, NTW = (NT + NS - 1) / NS;
  const int wm = w / NS, wn = w % NS;
  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    const bool part_ = t0 + BTV > tend;      /* (wave-uniform: only the last tile of a split masks its rows) */           \
    _Pragma("unroll") for (int x = 0; x < FPT; ++x) {                                                                    \
      const int pp = threadIdx.x + x * NTH, blk = pp >> 9;                                                               \
      const int tr = (blk / CBK) * 32 + ((pp & 511) >> 4), ch = (blk % CBK) * 16 + (pp & 15);                            \
      uint4 v = freg[ST][x];                                                                                             \
      if (part_) {                                                                                                       \
        const uint32_t mk = (t0 + tr < tend) ? 0xffffffffu : 0u;                                                         \
        v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;                                                                      \
      }                                                                                                                  \
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
      uint4 v = wreg[ST][x];                                                                                             \
      if constexpr (ROWS) {                                                                                              \
        const uint4 u = wreg2[ST][x];                                                                                    \
        v.x = (uint32_t)f2bf(__uint_as_float(v.x)) | ((uint32_t)f2bf(__uint_as_float(v.y)) << 16);                       \
        v.y = (uint32_t)f2bf(__uint_as_float(v.z)) | ((uint32_t)f2bf(__uint_as_float(v.w)) << 16);                       \
        v.z = (uint32_t)f2bf(__uint_as_float(u.x)) | ((uint32_t)f2bf(__uint_as_float(u.y)) << 16);                       \
        v.w = (uint32_t)f2bf(__uint_as_float(u.z)) | ((uint32_t)f2bf(__uint_as_float(u.w)) << 16);                       \
      }                                                                                                                  \
      if (part_) {                                                                                                       \
        const uint32_t mk = (t0 + tr < tend) ? 0xffffffffu : 0u;                                                         \
        v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;                                                                      \
      }                                                                                                                  \
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
    BWD3_COMMIT(st, tile0 + st)                       //  wave has passed tile t's barrier since.  Committing tile t+1 BEHIND tile
    __syncthreads();                                  //  t's barrier, in one block with tile t's MFMAs, measured 2-4 % slower.)
    BWD3_ISSUE(st, tile0 + st + PD)
    const unsigned char *const Fs = Fs0 + (DB == 2 ? ((tile0 + st) & 1) * IMG : 0), *const Ws = Ws0 + (DB == 2 ? ((tile0 + st) & 1) * IMG : 0);
#pragma unroll
    for (int kk = 0; kk < BTV / 32; ++kk) {
      const int rlo = kk * 32 + 8 * g + qq, rhi = rlo + 4;       // (rlo & 8) == (rhi & 8) == 8*(g & 1)
      const int disp = (g & 1) << 7;
      bf16x8 a[MTW];
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int cb = (((wm * MTW + mt) * 16 + 4 * p) * 2) ^ disp;
        bf16x4 lo = lds_tr16(reinterpret_cast<const uint16_t *>(&Fs[rlo * FSB + cb]));
        bf16x4 hi = lds_tr16(reinterpret_cast<const uint16_t *>(&Fs[rhi * FSB + cb]));
        a[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if (NS > 1 && wn * NTW + nt >= NT) break;                   // (wave-uniform: the last wave column has fewer tiles)
        const int cb = ((wn * NTW + nt) * 16 + 4 * p) * 2 + disp;
        bf16x4 lo = lds_tr16(reinterpret_cast<const uint16_t *>(&Ws[rlo * WSB + cb]));
        bf16x4 hi = lds_tr16(reinterpret_cast<const uint16_t *>(&Ws[rhi * WSB + cb]));
        const bf16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[mt][nt], 0, 0, 0);
      }
    }
   }
  }
#undef BWD3_ISSUE
#undef BWD3_COMMIT
  float *slab = part + ((size_t)by * D + m0) * PS;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = (wm * MTW + mt) * 16 + g * 4 + reg;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        if (NS == 1 || wn * NTW + nt < NT) slab[(size_t)m * PS + (wn * NTW + nt) * 16 + i16] = acc[mt][nt][reg];
    }
}


// ------------------------------------------------------------------------------------------------------------
// forward v10 (the streaming forward of every projection of up to nine column tiles, bf16 and fp8; wider bf16 projections in
// column-range passes; wide fp8 ones: k_proj_fwd_f8s).  ONE balanced workgroup per CU that owns T/CUs (+-1) of the 16-row tiles,
// with as many waves (5..8) as give every wave two tiles (round 1: co-resident smaller workgroups did not overlap and the
// launch cost (max workgroups on a CU) x (one workgroup's time): 50 000 items cost as much as 65 536); the A operand arrives as
// CONTIGUOUS pieces of the tiled F --
// a 16-row tile x 128-column chunk is one contiguous 4-KB run, lane l moves 16 B at l*16 (+1 KB per load), exactly the load
// shape of the backward kernel (which streams F at the device's ceiling) -- into a WAVE-PRIVATE LDS area, from which the
// MFMA fragments are read back with ds_read_b128.  v8's fragment loads take 16 rows x 64 B per instruction: half of every
// 128-B line per instruction, every line requested by two instructions.  Nothing of A is shared between waves, so the
// detour costs no barrier (v9 tried this load shape with a workgroup-shared image and two barriers per chunk and lost).
// Plain C++ with scheduling fences; B as in v8 (chunk through a double-buffered shared LDS image, one barrier per chunk).
// ------------------------------------------------------------------------------------------------------------
template <int NT, int MT, bool F8, bool NTL>
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
        AR[mt][x] = ld_stream16<NTL>(asrc[mt] + ((size_t)ce_ << 12) + x * 512 + lane * 8);                          \
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

// NTL: the feature loads carry `nt` (tables that cannot stay in the Infinity Cache between the two passes of a step);
// n0: first column of this launch's column range (NT tiles from there; 0 unless a wide projection is covered in passes).
template <int NT, bool F8, bool NTL>
__global__ __launch_bounds__(512) void k_proj_fwd_bf16_v10(const uint16_t *__restrict__ F, int nrows, int D,
                                                           const uint16_t *__restrict__ Et, float *__restrict__ P, int PS,
                                                           const float *__restrict__ pscale, int stagger, int tiles_per_wave,
                                                           int n0) {
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
  // (chunk-major image: chunk c holds rows [0, PS) x 128 k; this launch's columns start at row n0 of every chunk)
  if (nlive == 2) v10_body<NT, 2, F8, NTL>(asrc, Et + (size_t)n0 * 128, lds_v10, myA, acc, D, cshift, lane, PS * 128);
  else v10_body<NT, 1, F8, NTL>(asrc, Et + (size_t)n0 * 128, lds_v10, myA, acc, D, cshift, lane, PS * 128);
  const float ps = F8 ? *pscale : 1.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    if (mt >= nlive) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int t = (first + mt) * 16 + q * 4 + reg;
      if (t < nrows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) P[(size_t)t * PS + n0 + nt * 16 + r] = acc[mt][nt][reg] * ps;
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

// The same products, parallelised for what the reference's CLI defaults produce (train_rec.py:23,33-35: batch 256, fp32
// features, k = 128, d = 20 -- a few hundred listed rows, 21 output columns): the kernels above give a wave 21 busy lanes
// and a 4096-step serial loop (975 us + 158 us per step at I = 10 000, D = 4096).  Same arithmetic (every product and every
// sum in fp64), other summation order.  Forward: k_proj_fwd_f32_mfma below (fp64 matrix instruction).
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


// Deq: row width in bf16-sized units (fp8 rows are addressed as bf16 rows of half the width)
#define FWD_ARGS (const uint16_t *)h->Ft, rows, (int)nrows, h->cfg.num_items, Deq, (const uint16_t *)h->Et, Pout, h->PS, h->errflag
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
  // tiny launches (row tiles x column tiles fit the chip twice): one column tile per workgroup, 16 waves split K
  bool split = NT > 1 && tiles * NT <= 2 * (int64_t)ncu;
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

// streaming (`nt`) feature loads when the table cannot stay in the 256-MiB Infinity Cache between the two passes of a step:
// bf16 tables always (C2: 410 MB), fp8 tables from ~192 MB (C2's / c5small's 205-MB fp8 table stays: the backward pass
// re-reads what the forward pass left in the cache, c2fp8 0.2091 vs 0.2152 ms/step with nt)
inline bool fwd_nt_loads(const bprx_handle *h) {
  const size_t bytes = (size_t)h->cfg.num_items * h->cfg.feat_dim * (h->cfg.feat_dtype == BPRX_F_FP8 ? 1 : 2);
  return h->cfg.feat_dtype == BPRX_F_BF16 ? true : bytes > ((size_t)256 << 20);
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
  const bool ntl = fwd_nt_loads(h);
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

// v10 over the whole table, NT (<= 9) column tiles from column n0
template <int NT>
void launch_v10(bprx_handle *h, int64_t nrows, float *Pout, hipStream_t s, int n0) {
  constexpr int NWMAX = 8, NWMIN = 5;
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const int Deq = f8 ? h->cfg.feat_dim / 2 : h->cfg.feat_dim;
  const float *pscale = h->qs + 1;
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
  const bool ntl = fwd_nt_loads(h);
#define V10_LAUNCH(F8_, NTL_)                                                                                             \
  do {                                                                                                                    \
    auto kfn = k_proj_fwd_bf16_v10<NT, F8_, NTL_>;                                                                        \
    (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    hipLaunchKernelGGL(kfn, dim3((unsigned)G), dim3(nw * 64), lds, s, (const uint16_t *)h->Ft, (int)nrows, Deq,            \
                       (const uint16_t *)h->Et, Pout, h->PS, pscale, 0, per_wave, n0);                                     \
  } while (0)
  if (f8) { if (ntl) V10_LAUNCH(true, true); else V10_LAUNCH(true, false); }
  else { if (ntl) V10_LAUNCH(false, true); else V10_LAUNCH(false, false); }
#undef V10_LAUNCH
}

// P = F.[E|Bp] over the whole table.  BPRX_FWD_VARIANT=0: the plain kernel (k_proj_fwd_bf16: two barriers per chunk, nothing
// overlapped) -- the reference every streaming kernel is tested against, and the fallback for odd widths.
//   shape                                           kernel                              covered by
//   <= 9 column tiles (d <= 143), bf16 or fp8       k_proj_fwd_bf16_v10                 test_gpu_variants::test_lds_staged_forward...
//   >= 10 tiles, fp8 (configs[4]: d = 256)          k_proj_fwd_f8s (scaled fp8 MFMA)    test_gpu_variants::test_wide_projection...
//   >= 10 tiles, bf16 (or BPRX_F8S=0)               v10 in right-aligned passes of 9    test_gpu_variants::test_wide_projection...
//   feat_dim not a multiple of 256 (512 for fp8)    k_proj_fwd_bf16 (plain)             test_gpu_parity (D = 128 / 384 shapes)
template <int NT>
int launch_fwd_nt(bprx_handle *h, const int32_t *rows, int64_t nrows, float *Pout, hipStream_t s) {
  constexpr int MTD = NT <= 9 ? 2 : 1;
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const int Deq = f8 ? h->cfg.feat_dim / 2 : h->cfg.feat_dim;
  const float *pscale = h->qs + 1;
  const bool plain = h->fwd_variant == 0 || Deq % 256 != 0 || rows;
  if (!plain) {
    if constexpr (NT >= 10) {
      static const int f8s_on = getenv("BPRX_F8S") ? atoi(getenv("BPRX_F8S")) : 1;   // 0: column-range passes (A/B, tests)
      if (f8 && f8s_on && h->EtS && h->cfg.feat_dim % 256 == 0) {
        launch_f8s<NT>(h, nullptr, nrows, nullptr, 0, Pout, s, 1);
        return 0;
      }
      // column ranges of nine tiles, the last one right-aligned: ranges may overlap by some tiles, which are then computed and
      // stored twice, identically (F is read once per pass: two passes at d = 256)
      for (int c0 = 0; c0 < NT; c0 += 9) launch_v10<9>(h, nrows, Pout, s, (c0 + 9 <= NT ? c0 : NT - 9) * 16);
      return 0;
    } else {
      launch_v10<NT>(h, nrows, Pout, s, 0);
      return 0;
    }
  }
  dim3 grid((unsigned)((nrows + 4 * MTD * 16 - 1) / (4 * MTD * 16)));
  if (f8) hipLaunchKernelGGL((k_proj_fwd_bf16<NT, MTD, true>), grid, dim3(256), 0, s, FWD_ARGS, 0, pscale);
  else hipLaunchKernelGGL((k_proj_fwd_bf16<NT, MTD, false>), grid, dim3(256), 0, s, FWD_ARGS, 0, pscale);
  return 0;
}

// one launch of k_proj_bwd_bf16_v3 with its dynamic LDS size (DB images of F tile + W tile)
template <int NT, int BTV, int NW, int PD, bool F8, bool ROWS, int DB, int NS = 1>
void launch_bwd3(dim3 grid, hipStream_t s, const uint16_t *Ft, int nrows, int D, const uint16_t *Wb, int PS, float *part, int rps,
                 int desc, int xmap, const int32_t *rows, const int32_t *nrows_dev, const float *Wf) {
  constexpr size_t lds = (size_t)DB * (BTV * (NW * 32 * 2 + 32) + BTV * WsStride3<NT>::bytes);
  auto kfn = k_proj_bwd_bf16_v3<NT, BTV, NW, PD, F8, ROWS, DB, NS>;
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

// dE|dBp slabs = F^T W over the whole table: k_proj_bwd_bf16_v3 (W as bf16, conflict-free LDS image for the transpose reads).
//   shape                                        instantiation                                      why
//   bf16, <= 9 column tiles, D % 256 == 0        8 waves (256 columns), 3 tiles in flight           C2 / c4 shard: HBM-bound, deepest prefetch
//   fp8, or > 9 column tiles                     8 waves, 2 tiles in flight, double-buffered LDS    MFMA-paced: one barrier per tile (c5, c2fp8);
//                                                > 9 tiles: waves as a 4 x 2 grid (NS = 2)
//   D % 256 != 0 (bf16)                          4 waves (128 columns), 2 tiles in flight           small / odd feature widths
template <int NT>
int launch_bwd_nt(bprx_handle *h, hipStream_t s) {
  const int D = h->cfg.feat_dim, I = h->cfg.num_items;
  int rps3 = (I + h->SK - 1) / h->SK;
  rps3 = (rps3 + 31) / 32 * 32;
  const size_t n4 = (size_t)I * h->PS / 4;
  if (!h->item_mode)   // k_item_seg has already written Wb (bf16) itself
    hipLaunchKernelGGL(k_cast_W, dim3(1024), dim3(256), 0, s, h->W, (uint16_t *)h->Wb, n4);
  const bool f8 = h->cfg.feat_dtype == BPRX_F_FP8;
  const bool w8 = D % 256 == 0 || f8;                   // fp8: a 256-column tile row is one 256-B block row
  dim3 g3(D / (w8 ? 256 : 128), h->SK);
#define BWD3_ARGS (const uint16_t *)h->Ft, I, D, (const uint16_t *)h->Wb, h->PS, h->part, rps3, 0, 1, (const int32_t *)nullptr, \
                  (const int32_t *)nullptr, (const float *)nullptr
  if constexpr (NT > 9) {        // wide projections: 4 x 2 wave grid (2 % on c5 / c5small / c5bf16; the kernel stays paced by
    if (f8) { launch_bwd3<NT, 32, 8, 2, true, false, 2, 2>(g3, s, BWD3_ARGS); return 0; }          // issue, not by LDS volume)
    if (w8) { launch_bwd3<NT, 32, 8, 2, false, false, 2, 2>(g3, s, BWD3_ARGS); return 0; }
  }
  if constexpr (NT <= 9) {
    if (f8) { launch_bwd3<NT, 32, 8, 2, true, false, 2>(g3, s, BWD3_ARGS); return 0; }
    if (w8) { launch_bwd3<NT, 32, 8, 3, false, false, 1>(g3, s, BWD3_ARGS); return 0; }
  }
  launch_bwd3<NT, 32, 4, 2, false, false, 1>(g3, s, BWD3_ARGS);
#undef BWD3_ARGS
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
    // fp32 features (the reference-precision path, fp64 accumulation): the fp64 matrix instruction where D allows, the plain
    // wave-per-row kernel otherwise
    if (h->cfg.feat_dim % 16 == 0) {
      dim3 grid((unsigned)((nrows + 15) / 16), (unsigned)((h->cfg.embed_d + 1 + 15) / 16));
      hipLaunchKernelGGL(k_proj_fwd_f32_mfma, grid, dim3(1024), 0, s, (const float *)h->t.F, rows, (int)nrows, nrows_dev, scatter,
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
      // a row LIST is split over the lane groups of a block (vector-ALU tiles; the fp64-matrix form of this sum was no faster:
      // 19 vs 17 us, removed); the whole-table sum keeps one block per k
      if (h->list_mode && D % F32_KV == 0)
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
