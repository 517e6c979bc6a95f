// Narrow-operand variants of the implicit-GEMM convolution kernels (csrc/conv_igemm.hip): the `--amp` arithmetic of the
// reference (models/models.py:22-26, options/base_options.py:127; BASELINE.json configs[4] "fp16 MFMA path").
//
// Activations, weights and gradients stay fp32 in HBM - every other kernel of the step is unchanged - and only the two
// GEMM operands are narrowed while a tile is staged through LDS; products are accumulated in fp32 by the matrix cores
// (v_mfma_f32_32x32x16_{f16,bf16}: 16x the fp32 MFMA rate per instruction):
//
//   mode 1 "f16"     x -> half(x) (round to nearest even).  One MFMA per tile pair.  This is apex O1's contract for the
//                    contraction itself (fp16 operands, fp32 accumulate); gradients need the dynamic loss scale of
//                    csrc/amp.hip because dout is an operand of the two backward GEMMs.
//   mode 2 "bf16x3"  x -> hi + lo, hi = bf16(x), lo = bf16(x - hi); a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (three MFMAs,
//                    smallest terms first).  16 mantissa bits, fp32 range: ~1e-5 relative per product, no loss scale.
//
// LDS image: both operand tiles are stored [row][32 k] with k contiguous - 64-byte rows, no padding - and the four 16-byte
// slots of a row XOR-swizzled by the row index: element (row, k) lives at row*32 + (((k >> 3) ^ ((row >> 2) & 3)) << 3) + (k & 7).
//   * A lane's MFMA fragment (8 consecutive k of one row = one slot) is ONE ds_read_b128.  That instruction is serviced in
//     16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) over a 64-dword bank row (MI355X_MICROARCH.md, LDS): the rows of
//     a group differ in (row & 3, (row >> 2) & 3), i.e. in (quarter of the bank row, swizzled slot) - conflict-free.
//   * The A-tile of the forward kernel (gathered pixels x k) is written as 4-element vectors (ds_write_b64: the 8 k-quads of a row
//     fill its 16 dwords, consecutive rows alternate bank halves - conflict-free).
//   * Every other tile arrives with the reduction index as the slow HBM index (K-major weights; activations and gradients by
//     pixel in the weight-gradient kernel) and is transposed on the way in: a work-item owns two consecutive k of four
//     consecutive rows and writes four packed pairs, with the 16 k-pairs of a chunk on consecutive lanes.  A 32-lane store group
//     then covers the 16 dwords of two rows of equal parity: 2-way, which a 4-byte LDS store absorbs (same section).
// Both tiles are double-buffered (two LDS images, 16-48 KB per workgroup) and the global loads run two chunks ahead (two
// register stages): a K chunk of 8 (x3) MFMAs per wave is short, so it pays for ONE barrier instead of two - chunk c+1 is stored
// into the idle image while chunk c is still being multiplied by the slower waves, and chunk c+2 is already in flight.
// The k <-> (lane, element) assignment inside one MFMA is the same for A and B, so the sum over k does not depend on it; the
// C/D layout is the dtype-independent 32x32 map already used by the fp32 kernels.
//
// Entry points mirror fsv_conv_gather_fwd / fsv_conv_wgrad with one extra `mode` argument and only accept what the
// narrow kernels implement (Cin % 4 == 0); callers route everything else to the fp32 entry points.
#include "fsv_common.h"
#include <type_traits>

#define FSV_NP_BK 32
#define FSV_NP_LDK 32   // elements per LDS row (64 bytes, slots swizzled by np_sw)

typedef _Float16 np_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 np_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 np_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 np_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 np_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 np_bf16x2 __attribute__((ext_vector_type(2)));

template <int MODE> struct NpTypes;
template <> struct NpTypes<1> { typedef _Float16 H; typedef np_f16x8 H8; typedef np_f16x4 H4; typedef np_f16x2 H2; static constexpr int NP = 1; };
template <> struct NpTypes<2> { typedef __bf16 H; typedef np_bf16x8 H8; typedef np_bf16x4 H4; typedef np_bf16x2 H2; static constexpr int NP = 2; };

__device__ __forceinline__ f32x16 np_mfma(np_f16x8 a, np_f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 np_mfma(np_bf16x8 a, np_bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// hi / lo planes of one value (plane 1 is only used by mode 2)
template <int MODE>
__device__ __forceinline__ void np_split(float x, typename NpTypes<MODE>::H& hi, typename NpTypes<MODE>::H& lo) {
  typedef typename NpTypes<MODE>::H H;
  hi = (H)x;
  if constexpr (MODE == 2) lo = (H)(x - (float)hi); else lo = (H)0.f;
}

// element offset of (row, k) in a swizzled LDS tile (see the header comment); k & 7 stays inside its 16-byte slot
__device__ __forceinline__ int np_sw(int row, int k) {
  return row * FSV_NP_LDK + ((((k >> 3) ^ (row >> 2)) & 3) << 3) + (k & 7);
}

// v or zeros, component by component (a ?: between two float4 STRUCTS makes the compiler keep a zero struct in scratch memory
// and select between addresses)
__device__ __forceinline__ float4 np_keep(bool ok, const float4& v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// four values -> 4-element hi / lo vectors; two values -> a packed pair (no local arrays: they would live in scratch memory)
template <int MODE>
__device__ __forceinline__ void np_split4(const float4& v, typename NpTypes<MODE>::H4& hi, typename NpTypes<MODE>::H4& lo) {
  typename NpTypes<MODE>::H h0, l0, h1, l1, h2, l2, h3, l3;
  np_split<MODE>(v.x, h0, l0); np_split<MODE>(v.y, h1, l1); np_split<MODE>(v.z, h2, l2); np_split<MODE>(v.w, h3, l3);
  hi = typename NpTypes<MODE>::H4{h0, h1, h2, h3};
  lo = typename NpTypes<MODE>::H4{l0, l1, l2, l3};
}
template <int MODE>
__device__ __forceinline__ void np_pair(float a, float b, typename NpTypes<MODE>::H2& hi, typename NpTypes<MODE>::H2& lo) {
  typename NpTypes<MODE>::H h0, l0, h1, l1;
  np_split<MODE>(a, h0, l0); np_split<MODE>(b, h1, l1);
  hi = typename NpTypes<MODE>::H2{h0, h1};
  lo = typename NpTypes<MODE>::H2{l0, l1};
}
// the four packed pairs of two float4 (same channels, two consecutive k) into rows row0 .. row0+3 at column k of plane images
template <int MODE, int NP, int ROWLEN>
__device__ __forceinline__ void np_store_pairs(typename NpTypes<MODE>::H (*img)[ROWLEN], const float4& v0, const float4& v1,
                                               int row0, int k) {
  typedef typename NpTypes<MODE>::H2 H2;
  H2 hi, lo;
  np_pair<MODE>(v0.x, v1.x, hi, lo);
  *reinterpret_cast<H2*>(&img[0][np_sw(row0 + 0, k)]) = hi;
  if constexpr (NP == 2) *reinterpret_cast<H2*>(&img[NP - 1][np_sw(row0 + 0, k)]) = lo;
  np_pair<MODE>(v0.y, v1.y, hi, lo);
  *reinterpret_cast<H2*>(&img[0][np_sw(row0 + 1, k)]) = hi;
  if constexpr (NP == 2) *reinterpret_cast<H2*>(&img[NP - 1][np_sw(row0 + 1, k)]) = lo;
  np_pair<MODE>(v0.z, v1.z, hi, lo);
  *reinterpret_cast<H2*>(&img[0][np_sw(row0 + 2, k)]) = hi;
  if constexpr (NP == 2) *reinterpret_cast<H2*>(&img[NP - 1][np_sw(row0 + 2, k)]) = lo;
  np_pair<MODE>(v0.w, v1.w, hi, lo);
  *reinterpret_cast<H2*>(&img[0][np_sw(row0 + 3, k)]) = hi;
  if constexpr (NP == 2) *reinterpret_cast<H2*>(&img[NP - 1][np_sw(row0 + 3, k)]) = lo;
}

struct NpConvP {
  const float* in;
  const float* wt;
  const float* bias;
  const float* res;
  const float* wscale;
  float* out;
  int N, H, W, Cin;
  int OH, OW, Cout;
  int K, nchunks, ldw;
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;
  int outH, outW, osy, osx, ooy, oox, dense_out;
  long long w_bstride, b_bstride;
  int per_sample, nsplit;
  int act; float scale;
  int Mz;
};

__device__ __forceinline__ void np_tap(unsigned long long lo, unsigned long long hi, int t, int& ty, int& tx) {
  unsigned long long code = (t < 8) ? lo : hi;
  int sh = (t & 7) * 8;
  ty = (int)((code >> sh) & 15ull) - 8;
  tx = (int)((code >> (sh + 4)) & 15ull) - 8;
}

// out[z][m][co] = sum_k in[gather(m, k)] * wt[z][k][co]; same contract as fsv_conv_igemm_kernel (V = 4 gathers only)
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, 2) void fsv_np_conv_kernel(NpConvP p) {
  typedef NpTypes<MODE> T;
  typedef typename T::H H;
  typedef typename T::H8 H8;
  typedef typename T::H4 H4;
  typedef typename T::H2 H2;
  constexpr int NP = T::NP;
  constexpr int BK = FSV_NP_BK, LDK = FSV_NP_LDK;
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int KV = BK / 4;          // A float4 per pixel row and chunk
  constexpr int RPP = NT / KV;        // A rows per pass
  constexpr int NPA = BM / RPP;       // A passes
  constexpr int KP = BK / 2;          // k pairs per chunk: the fast work-item index of the transposing B store
  constexpr int QPB = NT / KP;         // B column quads per pass
  constexpr int NPB = BN / (4 * QPB);  // B passes
  static_assert(NPA >= 1 && NPB >= 1 && NPA * RPP == BM && NPB * 4 * QPB == BN, "tile / thread-count mismatch");
  // two images of each tile (double buffering, one workgroup barrier per K chunk): [image][plane]
  __shared__ __attribute__((aligned(16))) H As[2 * NP][BM * LDK];
  __shared__ __attribute__((aligned(16))) H Bs[2 * NP][BN * LDK];
  static_assert(2 * NP * (BM + BN) * LDK * 2 <= 65536, "LDS images must fit a static allocation");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = blockIdx.z / p.nsplit, zk = blockIdx.z % p.nsplit;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const float* wt = p.wt + (long long)zs * p.w_bstride;

  const int kq = tid % KV, ar0 = tid / KV;
  int a_iy0[NPA], a_ix0[NPA];
  long long a_base[NPA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int m = bm0 + ar0 + i * RPP;
    if (m < p.Mz) {
      int n, rem;
      if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
      int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.sy; a_ix0[i] = ox * p.sx;
      a_base[i] = (long long)n * p.H * p.W;
    } else {
      a_iy0[i] = -(1 << 28); a_ix0[i] = 0; a_base[i] = 0;
    }
  }
  const int bkp = tid % KP, bq0 = tid / KP;      // k pair, first column quad

  const int cps = (p.nchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.nchunks) ? (c_begin + cps) : p.nchunks;

  // two register stages: the global loads of chunk c+2 are issued before the MFMAs of chunk c (a chunk is only 8 (x3) MFMAs
  // per wave = 0.1-0.3 us of matrix work, far less than an HBM miss); chunk j waits in stage (j - c_begin) & 1
  float4 areg[2][NPA];
  float4 breg[2][NPB][2];

  auto load_chunk = [&](int kc, auto stage) {
    constexpr int S = decltype(stage)::value;
    const int k = kc * BK + kq * 4;
    const bool kok = k < p.K;
    int t = kok ? (k / p.Cin) : 0;
    int ci = kok ? (k - t * p.Cin) : 0;
    int ty, tx;
    np_tap(p.taps_lo, p.taps_hi, t, ty, tx);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
      bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      long long off = ok ? ((a_base[i] + (long long)iy * p.W + ix) * p.Cin + ci) : 0ll;
      float4 v = *reinterpret_cast<const float4*>(p.in + off);
      areg[S][i] = np_keep(ok, v);
    }
    const long long krow = (long long)(kc * BK + 2 * bkp) * p.ldw;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int bcol = bn0 + (bq0 + i * QPB) * 4;
      const bool bcol_ok = bcol < p.ldw;
      const int bcol_safe = bcol_ok ? bcol : 0;
      float4 v0 = *reinterpret_cast<const float4*>(wt + krow + bcol_safe);
      float4 v1 = *reinterpret_cast<const float4*>(wt + krow + p.ldw + bcol_safe);
      breg[S][i][0] = np_keep(bcol_ok, v0);
      breg[S][i][1] = np_keep(bcol_ok, v1);
    }
  };
  auto store_chunk = [&](int buf, auto stage) {
    constexpr int S = decltype(stage)::value;
    H (*as)[BM * LDK] = As + buf * NP;
    H (*bs)[BN * LDK] = Bs + buf * NP;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int r = ar0 + i * RPP;
      H4 hi, lo;
      np_split4<MODE>(areg[S][i], hi, lo);
      *reinterpret_cast<H4*>(&as[0][np_sw(r, kq * 4)]) = hi;
      if constexpr (NP == 2) *reinterpret_cast<H4*>(&as[NP - 1][np_sw(r, kq * 4)]) = lo;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int row0 = (bq0 + i * QPB) * 4;
      np_store_pairs<MODE, NP, BN * LDK>(bs, breg[S][i][0], breg[S][i][1], row0, 2 * bkp);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lk = lane >> 5;
  const int a_row = wm * (TM * 32) + lrow, b_row = wn * (TN * 32) + lrow;
  auto compute = [&](int cur) {
    const H (*as)[BM * LDK] = As + cur * NP;
    const H (*bs)[BN * LDK] = Bs + cur * NP;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      H8 a[NP][TM], b[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const H8*>(&as[q][np_sw(a_row + i * 32, ks * 16 + lk * 8)]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const H8*>(&bs[q][np_sw(b_row + j * 32, ks * 16 + lk * 8)]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (NP == 2) {
            acc[i][j] = np_mfma(a[NP - 1][i], b[0][j], acc[i][j]);
            acc[i][j] = np_mfma(a[0][i], b[NP - 1][j], acc[i][j]);
          }
          acc[i][j] = np_mfma(a[0][i], b[0][j], acc[i][j]);
        }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (c_begin < c_end) {
    // Chunk j is stored into LDS image (j - c_begin) & 1 one iteration before it is multiplied; the image being filled was
    // last read before the previous barrier.  Loads and stores past the last chunk repeat it (valid addresses, an image that
    // nobody reads any more), so the loop body has no branch around memory operations.
    const int c_last = c_end - 1;
    load_chunk(c_begin, S0());
    store_chunk(0, S0());
    load_chunk(c_begin + 1 < c_end ? c_begin + 1 : c_last, S1());
    __syncthreads();
#pragma unroll 1
    for (int kc = c_begin; kc < c_end; kc += 2) {
      load_chunk(kc + 2 < c_end ? kc + 2 : c_last, S0());
      compute(0);
      store_chunk(1, S1());
      __syncthreads();
      if (kc + 1 >= c_end) break;                    // uniform
      load_chunk(kc + 3 < c_end ? kc + 3 : c_last, S1());
      compute(1);
      store_chunk(0, S0());
      __syncthreads();
    }
  }

  // epilogue: identical to the fp32 kernel (D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
  const float* bias = p.bias ? (p.bias + (long long)zs * p.b_bstride) : nullptr;
  const float ws = p.wscale ? p.wscale[0] : 1.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = bn0 + wn * (TN * 32) + j * 32 + lrow;
    if (co >= p.Cout) continue;
    const float bv = (bias && p.nsplit == 1) ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = bm0 + wm * (TM * 32) + i * 32 + row;
        if (m >= p.Mz) continue;
        long long opix;
        if (p.dense_out) {
          opix = (long long)zs * (p.per_sample ? p.Mz : 0) + m;
        } else {
          int n, rem;
          if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
          int oy = rem / p.OW, ox = rem - oy * p.OW;
          opix = ((long long)n * p.outH + (oy * p.osy + p.ooy)) * p.outW + (ox * p.osx + p.oox);
        }
        float* dst = p.out + opix * p.Cout + co;
        float v = acc[i][j][r] * ws;
        if (p.nsplit > 1) {
          atomicAdd(dst, v);
        } else {
          v = (v + bv) * p.scale;
          v = fsv_act(v, p.act);
          if (p.res) v += p.res[opix * p.Cout + co];
          *dst = v;
        }
      }
    }
  }
}

// finishing pass of split-K launches: out = act((out + bias) * scale) + res
__global__ __launch_bounds__(256) void fsv_np_finish_kernel(float* out, const float* bias, const float* res,
                                                            long long total, int C, long long pix_per_sample,
                                                            long long b_bstride, int act, float scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    int c = (int)(i % C);
    long long pix = i / C;
    float v = out[i];
    if (bias) {
      long long n = b_bstride ? pix / pix_per_sample : 0;
      v += bias[n * b_bstride + c];
    }
    v = fsv_act(v * scale, act);
    if (res) v += res[i];
    out[i] = v;
  }
}

// ---- weight gradient: dwt[z][t*Cin+ci][co] (+)= sum_pixels in[gather(pixel, t, ci)] * dout[pixel][co] --------------------
struct NpWgradP {
  const float* in;
  const float* dout;
  float* dwt;
  int N, H, W, Cin;
  int OH, OW, Cout;
  int K, ldw;
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;
  long long w_bstride;
  int per_sample, nsplit;
  int Mz;
  int pchunks;
};

// Both operands arrive with the reduction index (the pixel) as the slow HBM index, so both tiles are transposed on the way
// into LDS; a work-item owns two consecutive pixels of four channels and writes packed pairs.
template <int BMK, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN, (BMK * BN >= 128 * 128) ? 1 : 2) void fsv_np_wgrad_kernel(NpWgradP p) {
  typedef NpTypes<MODE> T;
  typedef typename T::H H;
  typedef typename T::H8 H8;
  typedef typename T::H4 H4;
  typedef typename T::H2 H2;
  constexpr int NP = T::NP;
  constexpr int BK = FSV_NP_BK, LDK = FSV_NP_LDK;
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BMK / (WM * 32), TN = BN / (WN * 32);
  constexpr int PP = BK / 2;                         // pixel pairs per chunk: the fast work-item index of both stores
  constexpr int QPP = NT / PP;                       // column quads per pass
  constexpr int NPA = BMK / (4 * QPP), NPB = BN / (4 * QPP);
  static_assert(TM >= 1 && TN >= 1, "tile");
  static_assert(NPA >= 1 && NPB >= 1 && NPA * 4 * QPP == BMK && NPB * 4 * QPP == BN, "tile / thread-count mismatch");
  __shared__ __attribute__((aligned(16))) H As[2 * NP][BMK * LDK];       // [image][plane], double-buffered like the forward kernel
  __shared__ __attribute__((aligned(16))) H Bs[2 * NP][BN * LDK];
  static_assert(2 * NP * (BMK + BN) * LDK * 2 <= 65536, "LDS images must fit a static allocation");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = blockIdx.z / p.nsplit, zk = blockIdx.z % p.nsplit;
  const int bi0 = blockIdx.x * BMK, bn0 = blockIdx.y * BN;
  float* dwt = p.dwt + (long long)zs * p.w_bstride;
  const int ohw = p.OH * p.OW;

  const int pp = tid % PP, q0 = tid / PP;            // pixel pair of the chunk, first column quad
  // A: the (tap, channel) quads handled by this work-item are fixed for the whole reduction
  bool kok[NPA];
  int a_ci[NPA], a_ty[NPA], a_tx[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int kcol = bi0 + (q0 + i * QPP) * 4;
    kok[i] = kcol < p.K;
    const int t = kok[i] ? kcol / p.Cin : 0;
    a_ci[i] = kok[i] ? (kcol - t * p.Cin) : 0;
    np_tap(p.taps_lo, p.taps_hi, t, a_ty[i], a_tx[i]);
  }

  const int cps = (p.pchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.pchunks) ? (c_begin + cps) : p.pchunks;

  float4 areg[2][NPA][2];      // two register stages, as in the forward kernel
  float4 breg[2][NPB][2];
  const bool cout4 = (p.Cout & 3) == 0;
  auto load_chunk = [&](int pc, auto stage) {
    constexpr int S = decltype(stage)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // the two pixels of this work-item's pair: one decomposition each, shared by all its column quads
      const int m = pc * BK + 2 * pp + h;
      const bool mok = m < p.Mz;
      const int mm = mok ? m : 0;
      int n, rem;
      if (p.per_sample) { n = zs; rem = mm; } else { n = mm / ohw; rem = mm - n * ohw; }
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int iy = oy * p.sy + a_ty[i], ix = ox * p.sx + a_tx[i];
        const bool ok = mok && kok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const long long off = ok ? ((((long long)n * p.H + iy) * p.W + ix) * p.Cin + a_ci[i]) : 0ll;
        float4 v = *reinterpret_cast<const float4*>(p.in + off);
        areg[S][i][h] = np_keep(ok, v);
      }
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bcol = bn0 + (q0 + i * QPP) * 4;
        int m = pc * BK + 2 * pp + h;
        bool rok = m < p.Mz;
        long long pix = (long long)zs * (p.per_sample ? p.Mz : 0) + (rok ? m : 0);
        if (cout4) {
          bool ok = rok && bcol < p.Cout;
          float4 v = *reinterpret_cast<const float4*>(p.dout + (ok ? (pix * p.Cout + bcol) : 0ll));
          breg[S][i][h] = np_keep(ok, v);
        } else {
          const float* src = p.dout + pix * p.Cout;
          bool o0 = rok && bcol + 0 < p.Cout, o1 = rok && bcol + 1 < p.Cout, o2 = rok && bcol + 2 < p.Cout, o3 = rok && bcol + 3 < p.Cout;
          float t0 = src[o0 ? bcol + 0 : 0], t1 = src[o1 ? bcol + 1 : 0], t2 = src[o2 ? bcol + 2 : 0], t3 = src[o3 ? bcol + 3 : 0];
          breg[S][i][h] = make_float4(o0 ? t0 : 0.f, o1 ? t1 : 0.f, o2 ? t2 : 0.f, o3 ? t3 : 0.f);
        }
      }
  };
  auto store_chunk = [&](int buf, auto stage) {
    constexpr int S = decltype(stage)::value;
    H (*as)[BMK * LDK] = As + buf * NP;
    H (*bs)[BN * LDK] = Bs + buf * NP;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int pr = 2 * pp, aq = q0 + i * QPP;
      np_store_pairs<MODE, NP, BMK * LDK>(as, areg[S][i][0], areg[S][i][1], aq * 4, pr);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int pr = 2 * pp, bq = q0 + i * QPP;
      np_store_pairs<MODE, NP, BN * LDK>(bs, breg[S][i][0], breg[S][i][1], bq * 4, pr);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lk = lane >> 5;
  const int a_row = wm * (TM * 32) + lrow, b_row = wn * (TN * 32) + lrow;
  auto compute = [&](int cur) {
    const H (*as)[BMK * LDK] = As + cur * NP;
    const H (*bs)[BN * LDK] = Bs + cur * NP;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      H8 a[NP][TM], b[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const H8*>(&as[q][np_sw(a_row + i * 32, ks * 16 + lk * 8)]);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const H8*>(&bs[q][np_sw(b_row + j * 32, ks * 16 + lk * 8)]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (NP == 2) {
            acc[i][j] = np_mfma(a[NP - 1][i], b[0][j], acc[i][j]);
            acc[i][j] = np_mfma(a[0][i], b[NP - 1][j], acc[i][j]);
          }
          acc[i][j] = np_mfma(a[0][i], b[0][j], acc[i][j]);
        }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  if (c_begin < c_end) {
    const int c_last = c_end - 1;
    load_chunk(c_begin, S0());
    store_chunk(0, S0());
    load_chunk(c_begin + 1 < c_end ? c_begin + 1 : c_last, S1());
    __syncthreads();
#pragma unroll 1
    for (int pc = c_begin; pc < c_end; pc += 2) {
      load_chunk(pc + 2 < c_end ? pc + 2 : c_last, S0());
      compute(0);
      store_chunk(1, S1());
      __syncthreads();
      if (pc + 1 >= c_end) break;                    // uniform
      load_chunk(pc + 3 < c_end ? pc + 3 : c_last, S1());
      compute(1);
      store_chunk(0, S0());
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = bn0 + wn * (TN * 32) + j * 32 + lrow;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int k = bi0 + wm * (TM * 32) + i * 32 + row;
        if (k >= p.K) continue;
        float* dst = dwt + (long long)k * p.ldw + co;
        if (p.nsplit > 1) atomicAdd(dst, acc[i][j][r]); else *dst = acc[i][j][r];
      }
  }
}

// =============================================== host side ===================================================
extern "C" int fsv_conv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split,
                             int* tile_out, int* nsplit_out);

static inline void np_pack_taps(const int* ty, const int* tx, int n, unsigned long long& lo, unsigned long long& hi) {
  lo = 0; hi = 0;
  for (int t = 0; t < n; ++t) {
    unsigned long long c = (unsigned long long)((ty[t] + 8) & 15) | ((unsigned long long)((tx[t] + 8) & 15) << 4);
    if (t < 8) lo |= c << (t * 8); else hi |= c << ((t - 8) * 8);
  }
}

// the narrow kernels come in four 256-work-item tiles; thin / wide ids of the fp32 plan map onto the nearest of them
static inline int np_tile_of(int tile) {
  switch (tile) {
    case 0: case 5: case 6: case 7: return 0;      // 128 x 128
    case 1: case 2: case 3: case 8: return 1;      // 128 x 64
    case 9: return 9;                              // 64 x 128
    default: return 4;                             // 64 x 64
  }
}

template <int MODE>
static int np_launch_conv(const NpConvP& p, int nz, hipStream_t stream, int tile) {
  dim3 block(256);
  switch (tile) {
    case 0: {
      if constexpr (MODE == 2) return FSV_ERR_UNSUPPORTED;      // 64 KB of LDS: not instantiated (callers map it to 128x64)
      else { dim3 g(fsv_cdiv(p.Mz, 128), fsv_cdiv(p.Cout, 128), nz);
        FSV_LAUNCH((fsv_np_conv_kernel<128, 128, 2, 2, MODE>), g, block, stream, p); }
      break; }
    case 1: { dim3 g(fsv_cdiv(p.Mz, 128), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_np_conv_kernel<128, 64, 2, 2, MODE>), g, block, stream, p); break; }
    case 9: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 128), nz);
      FSV_LAUNCH((fsv_np_conv_kernel<64, 128, 2, 2, MODE>), g, block, stream, p); break; }
    case 4: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_np_conv_kernel<64, 64, 2, 2, MODE>), g, block, stream, p); break; }
    default: return FSV_ERR_BAD_ARG;
  }
  return fsv_check_launch();
}

template <int MODE>
static int np_launch_wgrad(const NpWgradP& p, int bmk, int bn, dim3 g, hipStream_t stream) {
  dim3 block(256);
  if (bmk == 128 && bn == 128) {
    if constexpr (MODE == 2) return FSV_ERR_UNSUPPORTED;
    else FSV_LAUNCH((fsv_np_wgrad_kernel<128, 128, 2, 2, MODE>), g, block, stream, p);
  } else if (bmk == 128 && bn == 64) FSV_LAUNCH((fsv_np_wgrad_kernel<128, 64, 2, 2, MODE>), g, block, stream, p);
  else if (bmk == 64 && bn == 128) FSV_LAUNCH((fsv_np_wgrad_kernel<64, 128, 2, 2, MODE>), g, block, stream, p);
  else if (bmk == 64 && bn == 64) FSV_LAUNCH((fsv_np_wgrad_kernel<64, 64, 2, 2, MODE>), g, block, stream, p);
  else return FSV_ERR_BAD_ARG;
  return fsv_check_launch();
}

extern "C" {

int fsv_conv_gather_fwd_np(const float* in, const float* wt, const float* bias, const float* res, float* out,
                           int N, int H, int W, int Cin, int OH, int OW, int Cout,
                           int ntaps, const int* ty, const int* tx, int sy, int sx,
                           int outH, int outW, int osy, int osx, int ooy, int oox,
                           int ldw, long long w_bstride, long long b_bstride, int per_sample,
                           int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                           int mode, hipStream_t stream) {
  if (!in || !wt || !out || ntaps < 1 || ntaps > 16 || N < 1 || Cin < 1 || Cout < 1) return FSV_ERR_BAD_ARG;
  if (mode != 1 && mode != 2) return FSV_ERR_BAD_ARG;
  if ((Cin & 3) != 0) return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((ldw & 3) != 0 || ldw < Cout) return FSV_ERR_BAD_ARG;
  NpConvP p;
  p.in = in; p.wt = wt; p.bias = bias; p.res = res; p.out = out; p.wscale = wscale;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.nchunks = fsv_cdiv(p.K, FSV_NP_BK); p.ldw = ldw;
  p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  np_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi);
  p.outH = outH; p.outW = outW; p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
  p.dense_out = (osy == 1 && osx == 1 && ooy == 0 && oox == 0 && outH == OH && outW == OW) ? 1 : 0;
  p.w_bstride = w_bstride; p.b_bstride = b_bstride; p.per_sample = per_sample ? 1 : 0;
  p.act = act; p.scale = scale;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  const int nsamp = per_sample ? N : 1;
  // same tile / split-K plan as the fp32 launcher (the stride-2 data gradient's zero-fill skip in conv.py asks
  // fsv_conv_plan for the split it will get, so the two must agree)
  int tile = 0, nsplit = 1;
  if (fsv_conv_plan(p.Mz, Cout, p.nchunks, nsamp, force_tile, force_split, &tile, &nsplit)) return FSV_ERR_BAD_ARG;
  tile = np_tile_of(tile);
  if (mode == 2 && tile == 0) tile = 1;      // two planes x two images of a 128x128 tile would take all 64 KB of a static allocation
  p.nsplit = nsplit;
  const long long total = (long long)N * outH * outW * Cout;
  if (accumulate) {
    if (bias || res || act != FSV_ACT_NONE || scale != 1.f) return FSV_ERR_BAD_ARG;
  } else if (nsplit > 1) {
    if (!p.dense_out) return FSV_ERR_UNSUPPORTED;
    (void)hipMemsetAsync(out, 0, (size_t)total * sizeof(float), stream);
  }
  int rc = (mode == 1) ? np_launch_conv<1>(p, nsamp * nsplit, stream, tile) : np_launch_conv<2>(p, nsamp * nsplit, stream, tile);
  if (rc) return rc;
  if (!accumulate && nsplit > 1 && (bias || res || act != FSV_ACT_NONE || scale != 1.f)) {
    int grid = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    FSV_LAUNCH(fsv_np_finish_kernel, dim3(grid), dim3(256), stream, out, bias, res, total, Cout,
               (long long)outH * outW, per_sample ? b_bstride : 0ll, act, scale);
    rc = fsv_check_launch();
  }
  return rc;
}

int fsv_conv_wgrad_np(const float* in, const float* dout, float* dwt,
                      int N, int H, int W, int Cin, int OH, int OW, int Cout,
                      int ntaps, const int* ty, const int* tx, int sy, int sx,
                      int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                      int force_tile, int mode, hipStream_t stream) {
  if (!in || !dout || !dwt || ntaps < 1 || ntaps > 16) return FSV_ERR_BAD_ARG;
  if (mode != 1 && mode != 2) return FSV_ERR_BAD_ARG;
  if ((Cin & 3) != 0) return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  NpWgradP p;
  p.in = in; p.dout = dout; p.dwt = dwt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.ldw = ldw; p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  np_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi);
  p.w_bstride = w_bstride; p.per_sample = per_sample ? 1 : 0;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  p.pchunks = fsv_cdiv(p.Mz, FSV_NP_BK);
  const int nsamp = per_sample ? N : 1;
  // tiles: 64-row tiles for everything but long-K wide layers (the fp32 plan of fsv_conv_wgrad, minus its 32-wide tiles);
  // force_tile: 1 = 64x64, 2 = 128x64, 3 = 64x128, 4 = 128x128 (rows x columns)
  int bmk = 64, bn = 64;
  long long target = 2048;
  if (Cout >= 128 && p.K >= 2304 && p.pchunks >= 64) { bn = 128; target = 1024; }
  if (force_tile == 1) { bmk = 64; bn = 64; }
  else if (force_tile == 2) { bmk = 128; bn = 64; }
  else if (force_tile == 3) { bmk = 64; bn = 128; }
  else if (force_tile == 4) { bmk = 128; bn = (mode == 2) ? 64 : 128; }      // (mode 2: see fsv_conv_gather_fwd_np)
  long long blocks = (long long)fsv_cdiv(p.K, bmk) * fsv_cdiv(Cout, bn) * nsamp;
  int nsplit = 1;
  if (force_split > 0) nsplit = force_split;
  else {
    nsplit = (int)((target + blocks - 1) / blocks);
    int maxs = p.pchunks / 8;                      // keep at least 8 chunks (256 pixels) per split
    if (nsplit > maxs) nsplit = maxs;
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > p.pchunks) nsplit = p.pchunks;
  p.nsplit = nsplit;
  if (nsplit > 1 && !prezeroed)
    (void)hipMemsetAsync(dwt, 0, (size_t)((per_sample ? (long long)N * w_bstride : (long long)Kpad * ldw)) * sizeof(float), stream);
  dim3 g(fsv_cdiv(p.K, bmk), fsv_cdiv(Cout, bn), nsamp * nsplit);
  return (mode == 1) ? np_launch_wgrad<1>(p, bmk, bn, g, stream) : np_launch_wgrad<2>(p, bmk, bn, g, stream);
}

}  // extern "C"
