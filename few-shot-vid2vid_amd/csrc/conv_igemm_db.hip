// Double-buffered variant of the fp32 implicit-GEMM kernel (conv_igemm.hip): two LDS images of the A / B tiles, so a K chunk
// costs ONE workgroup barrier instead of two - while the waves run the MFMAs of chunk c out of image c&1, the staging registers
// (filled by the global loads issued at the top of the iteration) are written into the other image, which every wave stopped
// reading at the previous barrier.  Same operands, same order of the fma chain per output element, same epilogue: results are
// bitwise those of fsv_conv_igemm_kernel.  LDS: 2 x (32 x (BM+1) + 32 x BN) floats = 33 KB (64x64) / 49 KB (64x128, 128x64),
// i.e. 3-4 workgroups per CU; the 128x128 tile would need 66 KB (2 workgroups per CU) and is not offered: it only carries the
// launches that already fill the chip.
//
// Experimental: reachable only through force_tile (ids 13 / 14 / 15) until an A/B on the hardware says where it pays
// (tools/tile_ab.py).  float4-gather layers only (Cin % 4 == 0).
#include "conv_igemm.h"
#include <type_traits>

// the workgroup's work, parameterised by its tile coordinates so that one launch can serve several problems (see the
// merged stride-2 data gradient below); `p` lives in kernel-argument (scalar) memory in both callers
// WSK (split-K through a workspace): instead of adding its partial tile into a zeroed output with atomics, every K split stores
// it to skw[split][tile] and takes a ticket; the workgroup that draws the tile's last ticket adds the splits in the fixed order
// 0, 1, 2, ... and runs the full epilogue (bias, activation, residual) - no zero fill, no finishing launch, and the result does
// not depend on which workgroup arrived when.  tile_lin: index of this output tile among all tiles of the launch.
template <int BM, int BN, int WM, int WN, int PF, bool WSK = false>
__device__ __forceinline__ void fsv_conv_db_body(const ConvP& p, const int bx, const int by, const int bz, float* skw = nullptr,
                                                 int* sk_tickets = nullptr, const int tile_lin = 0, const int ntiles = 0) {
  static_assert(PF == 1 || PF == 2, "prefetch distance");
  constexpr int BK = FSV_BK;
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int LDA = BM + 1;
  constexpr int KV = BK / 4;
  constexpr int RPP = NT / KV;
  constexpr int NPA = BM / RPP;
  constexpr int QB = BN / 4;
  constexpr int RPB = NT / QB;
  constexpr int NPB = BK / RPB;
  constexpr int ASZ = BK * LDA, BSZ = BK * BN;
  static_assert(NPA >= 1 && NPB >= 1 && NPA * RPP == BM && NPB * RPB == BK, "tile / thread-count mismatch");
  __shared__ float As[2 * ASZ];
  __shared__ float Bs[2 * BSZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bm0 = bx * BM, bn0 = by * BN;
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
  const int bq = tid % QB, br0 = tid / QB;
  const int bcol = bn0 + bq * 4;
  const bool bcol_ok = bcol < p.ldw;
  const int bcol_safe = bcol_ok ? bcol : 0;

  const int cps = (p.nchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.nchunks) ? (c_begin + cps) : p.nchunks;

  float areg[PF][NPA][4];       // PF register stages: chunk j waits in stage (j - c_begin) % PF until it is stored to LDS
  float4 breg[PF][NPB];

  auto load_chunk = [&](int kc, auto stage) {
    constexpr int S = decltype(stage)::value;
    const int k = kc * BK + kq * 4;
    const bool kok = k < p.K;
    int t = kok ? (k / p.Cin) : 0;
    int ci = kok ? (k - t * p.Cin) : 0;
    int ty, tx;
    fsv_tap(p, t, ty, tx);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
      bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      long long off = ok ? ((a_base[i] + (long long)iy * p.W + ix) * p.Cin + ci) : 0ll;
      float4 v = *reinterpret_cast<const float4*>(p.in + off);
      areg[S][i][0] = ok ? v.x : 0.f; areg[S][i][1] = ok ? v.y : 0.f; areg[S][i][2] = ok ? v.z : 0.f; areg[S][i][3] = ok ? v.w : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int kr = kc * BK + br0 + i * RPB;
      float4 v = *reinterpret_cast<const float4*>(wt + (long long)kr * p.ldw + bcol_safe);
      breg[S][i] = bcol_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&](int buf, auto stage) {
    constexpr int S = decltype(stage)::value;
    float* as = As + buf * ASZ;
    float* bs = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int r = ar0 + i * RPP;
#pragma unroll
      for (int j = 0; j < 4; ++j) as[(kq * 4 + j) * LDA + r] = areg[S][i][j];
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int kr = br0 + i * RPB;
      *reinterpret_cast<float4*>(&bs[kr * BN + bq * 4]) = breg[S][i];
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
  const int a_off = lk * LDA + wm * (TM * 32) + lrow;
  const int b_off = lk * BN + wn * (TN * 32) + lrow;
  auto compute = [&](int cur) {
    const float* a_frag = As + cur * ASZ + a_off;
    const float* b_frag = Bs + cur * BSZ + b_off;
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = a_frag[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = b_frag[j * 32];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int c = kk & 1, nx = c ^ 1;
      if (kk + 1 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[nx][i] = a_frag[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[nx][j] = b_frag[(kk + 1) * 2 * BN + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF - 1>;
  if (c_begin < c_end) {
    const int c_last = c_end - 1;
    load_chunk(c_begin, S0());
    store_chunk(0, S0());
    if constexpr (PF == 1) {
      __syncthreads();
#pragma unroll 1
      for (int kc = c_begin; kc < c_end; ++kc) {
        const int cur = (kc - c_begin) & 1;
        const bool more = kc + 1 < c_end;
        // the last iteration re-reads its own chunk (no per-lane branch around the loads); that copy is never stored
        load_chunk(more ? kc + 1 : kc, S0());
        compute(cur);
        if (more) store_chunk(cur ^ 1, S0());       // uniform over the workgroup
        __syncthreads();
      }
    } else {
      // prefetch distance 2: the global loads of chunk c+2 are issued before the MFMAs of chunk c, so they have two chunks of
      // matrix work (about 2 us) to arrive - what a lone workgroup on a CU needs to ride out an HBM miss.  Chunk j waits in
      // register stage (j - c_begin) & 1 and is stored into LDS image (j - c_begin) & 1 one iteration before it is multiplied.
      // Loads and stores past the last chunk repeat it (valid addresses, an image nobody reads any more): no branches.
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
  }

  bool finish = (p.nsplit == 1);          // this workgroup applies bias / activation / residual and stores the final values
  if constexpr (WSK) {
    __shared__ int s_last;
    constexpr int FR = TM * TN * 16;        // accumulator values per lane
    float* mine = skw + ((long long)zk * ntiles + tile_lin) * (BM * BN) + (wave * FR) * 64 + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];      // 64 lanes: 256 contiguous bytes
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      const int ticket = atomicAdd(sk_tickets + tile_lin, 1);
      s_last = (ticket == p.nsplit - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (tid == 0) sk_tickets[tile_lin] = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
          for (int sp = 0; sp < p.nsplit; ++sp)
            v += __hip_atomic_load(skw + ((long long)sp * ntiles + tile_lin) * (BM * BN) + (wave * FR + (i * TN + j) * 16 + r) * 64 + lane,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[i][j][r] = v;
        }
    finish = true;
  }
  const float* bias = p.bias ? (p.bias + (long long)zs * p.b_bstride) : nullptr;
  const float ws = p.wscale ? p.wscale[0] : 1.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = bn0 + wn * (TN * 32) + j * 32 + lrow;
    if (co >= p.Cout) continue;
    const float bv = (bias && finish) ? bias[co] : 0.f;
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
        if (!finish) {
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

template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_db_kernel(ConvP p) {
  fsv_conv_db_body<BM, BN, WM, WN, PF>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- XCD-aware tile order -------------------------------------------------------------------------------------------------------
// The 256 CUs sit in 8 XCDs with one L2 each, and consecutive workgroup ids are dealt to the XCDs round-robin.  In the plain
// (x = pixel tile, y = channel tile) grid the workgroups that share a pixel tile - the expensive operand: the gathered activations -
// have ids GX apart and land on different XCDs unless GX is a multiple of 8, so that tile is fetched from HBM once per L2 that
// needs it.  Here the grid is one-dimensional in x and id L is decoded as xcd = L % 8, slot = L / 8, channel tile = slot % GY,
// pixel tile = (slot / GY) * 8 + xcd: all channel tiles of a pixel tile run back to back on ONE XCD.
template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_dbx_kernel(ConvP p, int GX, int GY) {
  const int L = blockIdx.x;
  const int xcd = L & 7, slot = L >> 3;
  const int by = slot % GY, bx = (slot / GY) * 8 + xcd;
  if (bx >= GX) return;                              // padding of the last group of 8 pixel tiles (uniform per workgroup)
  fsv_conv_db_body<BM, BN, WM, WN, PF>(p, bx, by, blockIdx.z);
}

// ---- split-K through a workspace (see WSK above) ------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_dbk_kernel(ConvP p, float* skw, int* sk_tickets) {
  const int zs = blockIdx.z / p.nsplit;
  const int tile_lin = (zs * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
  const int ntiles = (int)(gridDim.z / p.nsplit) * (int)gridDim.y * (int)gridDim.x;
  fsv_conv_db_body<BM, BN, WM, WN, PF, true>(p, blockIdx.x, blockIdx.y, blockIdx.z, skw, sk_tickets, tile_lin, ntiles);
}

// ---- merged stride-2 data gradient: the four output-parity classes of one layer in ONE launch ----------------------------------
// A stride-2 convolution's data gradient is four independent gather-GEMMs (one per parity class of the output pixel, each with
// its own taps, weight layout and strided placement).  Launched one by one they are four small grids; here blockIdx.z selects
// the class (and, within it, the sample of a per-sample problem), blockIdx.x walks the tiles of the largest class - workgroups
// beyond a smaller class's tile count leave at once.  Only for the no-split case (every class stores its own pixels).
struct ConvP4 {
  ConvP c[4];
  int nz;                 // z extent of one class (samples of a per-sample problem, else 1)
};

template <int BM, int BN, int WM, int WN, int PF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_db4_kernel(ConvP4 q) {
  const int cls = blockIdx.z / q.nz;
  const ConvP& p = q.c[cls];
  if ((int)blockIdx.x * BM >= p.Mz) return;          // uniform per workgroup
  fsv_conv_db_body<BM, BN, WM, WN, PF>(p, blockIdx.x, blockIdx.y, blockIdx.z - cls * q.nz);
}

int fsv_launch_conv_db(const ConvP& p, int nz, hipStream_t stream, int tile) {
  dim3 block(256);
  switch (tile) {
    case 13: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<64, 64, 2, 2, 1>), g, block, stream, p); break; }
    case 14: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 128), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<64, 128, 2, 2, 1>), g, block, stream, p); break; }
    case 15: { dim3 g(fsv_cdiv(p.Mz, 128), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<128, 64, 2, 2, 1>), g, block, stream, p); break; }
    // 16 / 17 / 18: the same three tiles with the global loads issued two chunks ahead (two register stages)
    case 16: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<64, 64, 2, 2, 2>), g, block, stream, p); break; }
    case 17: { dim3 g(fsv_cdiv(p.Mz, 64), fsv_cdiv(p.Cout, 128), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<64, 128, 2, 2, 2>), g, block, stream, p); break; }
    case 18: { dim3 g(fsv_cdiv(p.Mz, 128), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_db_kernel<128, 64, 2, 2, 2>), g, block, stream, p); break; }
    // 19 / 20 / 21: double-buffered + XCD-aware tile order
    case 19: { const int gx = fsv_cdiv(p.Mz, 64), gy = fsv_cdiv(p.Cout, 64);
      FSV_LAUNCH((fsv_conv_igemm_dbx_kernel<64, 64, 2, 2, 1>), dim3(fsv_cdiv(gx, 8) * 8 * gy, 1, nz), block, stream, p, gx, gy); break; }
    case 20: { const int gx = fsv_cdiv(p.Mz, 64), gy = fsv_cdiv(p.Cout, 128);
      FSV_LAUNCH((fsv_conv_igemm_dbx_kernel<64, 128, 2, 2, 1>), dim3(fsv_cdiv(gx, 8) * 8 * gy, 1, nz), block, stream, p, gx, gy); break; }
    case 21: { const int gx = fsv_cdiv(p.Mz, 128), gy = fsv_cdiv(p.Cout, 64);
      FSV_LAUNCH((fsv_conv_igemm_dbx_kernel<128, 64, 2, 2, 1>), dim3(fsv_cdiv(gx, 8) * 8 * gy, 1, nz), block, stream, p, gx, gy); break; }
    default: return FSV_ERR_BAD_ARG;
  }
  return fsv_check_launch();
}

// ---- weight gradient, double-buffered: same restructuring of fsv_conv_wgrad_kernel (one barrier per 32-pixel chunk) ---------
template <int BMK, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_wgrad_db_kernel(WgradP p) {
  constexpr int BK = FSV_BK;   // pixels per chunk
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BMK / (WM * 32), TN = BN / (WN * 32);
  constexpr int QA = BMK / 4, RPA = NT / QA, NPA = BK / RPA;
  constexpr int QB = BN / 4, RPB = NT / QB, NPB = BK / RPB;
  constexpr int ASZ = BK * BMK, BSZ = BK * BN;
  static_assert(TM >= 1 && TN >= 1, "tile");
  static_assert(NPA >= 1 && NPB >= 1 && RPA >= 1 && NPA * RPA == BK && NPB * RPB == BK, "tile / thread-count mismatch");
  __shared__ float As[2 * ASZ];
  __shared__ float Bs[2 * BSZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = blockIdx.z / p.nsplit, zk = blockIdx.z % p.nsplit;
  const int bi0 = blockIdx.x * BMK, bn0 = blockIdx.y * BN;
  float* dwt = p.dwt + (long long)zs * p.w_bstride;
  const int ohw = p.OH * p.OW;

  const int aq = tid % QA, apr0 = tid / QA;
  const int kcol = bi0 + aq * 4;
  const bool kok = kcol < p.K;
  int t = kok ? kcol / p.Cin : 0;
  const int ci = kcol - t * p.Cin;
  int ty, tx;
  {
    unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
    int sh = (t & 7) * 8;
    ty = (int)((code >> sh) & 15ull) - 8;
    tx = (int)((code >> (sh + 4)) & 15ull) - 8;
  }
  const int bq = tid % QB, bpr0 = tid / QB;
  const int bcol = bn0 + bq * 4;

  const int cps = (p.pchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.pchunks) ? (c_begin + cps) : p.pchunks;

  float areg[NPA][4];
  float4 breg[NPB];
  const bool cout4 = (p.Cout & 3) == 0;
  auto load_chunk = [&](int pc) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int m = pc * BK + apr0 + i * RPA;
      bool ok = kok && m < p.Mz;
      int mm = ok ? m : 0;
      int n, rem;
      if (p.per_sample) { n = zs; rem = mm; } else { n = mm / ohw; rem = mm - n * ohw; }
      int oy = rem / p.OW, ox = rem - oy * p.OW;
      int iy = oy * p.sy + ty, ix = ox * p.sx + tx;
      ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      long long off = ok ? ((((long long)n * p.H + iy) * p.W + ix) * p.Cin + ci) : 0ll;
      float4 v = *reinterpret_cast<const float4*>(p.in + off);
      areg[i][0] = ok ? v.x : 0.f; areg[i][1] = ok ? v.y : 0.f; areg[i][2] = ok ? v.z : 0.f; areg[i][3] = ok ? v.w : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int m = pc * BK + bpr0 + i * RPB;
      bool rok = m < p.Mz;
      long long pix = (long long)zs * (p.per_sample ? p.Mz : 0) + (rok ? m : 0);
      if (cout4) {
        bool ok = rok && bcol < p.Cout;
        const float* src = p.dout + (ok ? (pix * p.Cout + bcol) : 0ll);
        float4 v = *reinterpret_cast<const float4*>(src);
        breg[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = p.dout + pix * p.Cout;
        bool o0 = rok && bcol + 0 < p.Cout, o1 = rok && bcol + 1 < p.Cout, o2 = rok && bcol + 2 < p.Cout, o3 = rok && bcol + 3 < p.Cout;
        float t0 = src[o0 ? bcol + 0 : 0], t1 = src[o1 ? bcol + 1 : 0], t2 = src[o2 ? bcol + 2 : 0], t3 = src[o3 ? bcol + 3 : 0];
        v.x = o0 ? t0 : 0.f; v.y = o1 ? t1 : 0.f; v.z = o2 ? t2 : 0.f; v.w = o3 ? t3 : 0.f;
        breg[i] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
    float* as = As + buf * ASZ;
    float* bs = Bs + buf * BSZ;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int pr = apr0 + i * RPA;
#pragma unroll
      for (int j = 0; j < 4; ++j) as[pr * BMK + aq * 4 + j] = areg[i][j];
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int pr = bpr0 + i * RPB;
      *reinterpret_cast<float4*>(&bs[pr * BN + bq * 4]) = breg[i];
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
  const int a_off = lk * BMK + wm * (TM * 32) + lrow;
  const int b_off = lk * BN + wn * (TN * 32) + lrow;
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk(0);
    __syncthreads();
#pragma unroll 1
    for (int pc = c_begin; pc < c_end; ++pc) {
      const int cur = (pc - c_begin) & 1;
      const bool more = pc + 1 < c_end;
      load_chunk(more ? pc + 1 : pc);
      const float* a_frag = As + cur * ASZ + a_off;
      const float* b_frag = Bs + cur * BSZ + b_off;
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = a_frag[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = b_frag[j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int c = kk & 1, nx = c ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[nx][i] = a_frag[(kk + 1) * 2 * BMK + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[nx][j] = b_frag[(kk + 1) * 2 * BN + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
      }
      if (more) store_chunk(cur ^ 1);
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

int fsv_launch_wgrad_db(const WgradP& p, int bmk, int bn, dim3 grid, hipStream_t stream) {
  dim3 block(256);
  if (bmk == 64 && bn == 64) FSV_LAUNCH((fsv_conv_wgrad_db_kernel<64, 64, 2, 2>), grid, block, stream, p);
  else if (bmk == 64 && bn == 128) FSV_LAUNCH((fsv_conv_wgrad_db_kernel<64, 128, 2, 2>), grid, block, stream, p);
  else return FSV_ERR_BAD_ARG;
  return fsv_check_launch();
}


extern "C" int fsv_conv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split, int* tile_out,
                             int* nsplit_out);

static inline void fsv_db_pack_taps(const int* ty, const int* tx, int n, unsigned long long& lo, unsigned long long& hi) {
  lo = 0; hi = 0;
  for (int t = 0; t < n; ++t) {
    unsigned long long c = (unsigned long long)((ty[t] + 8) & 15) | ((unsigned long long)((tx[t] + 8) & 15) << 4);
    if (t < 8) lo |= c << (t * 8); else hi |= c << ((t - 8) * 8);
  }
}

// The four parity classes of a stride-2 data gradient in one launch (see fsv_conv_igemm_db4_kernel).  Class k: weights wt[k]
// ([Kpad_k][ldw], K-major), ntaps[k] taps ty/tx[k*16 ..], iteration grid sub_h[k] x sub_w[k], output pixel (2*y + py[k], 2*x + px[k])
// of the [N][outH][outW][Cout] tensor.  Returns FSV_ERR_UNSUPPORTED when the launch plan wants split-K or a tile that has no
// double-buffered variant: the caller then issues the classes one by one.
extern "C" int fsv_conv_dgrad_s2(const float* in, const float* const* wt, float* out, int N, int H, int W, int Cin, int Cout,
                                 const int* ntaps, const int* ty, const int* tx, const int* sub_h, const int* sub_w, const int* py,
                                 const int* px, int outH, int outW, int ldw, const long long* w_bstride, int per_sample,
                                 const float* wscale, int prefetch, hipStream_t stream) {
  if (!in || !wt || !out || !ntaps || !ty || !tx || !sub_h || !sub_w || !py || !px || N < 1 || Cin < 1 || Cout < 1) return FSV_ERR_BAD_ARG;
  if ((Cin & 3) != 0) return FSV_ERR_UNSUPPORTED;
  if ((ldw & 3) != 0 || ldw < Cout) return FSV_ERR_BAD_ARG;
  ConvP4 q;
  const int nsamp = per_sample ? N : 1;
  q.nz = nsamp;
  int tile = -1, max_mz = 0;
  for (int k = 0; k < 4; ++k) {
    if (ntaps[k] < 1 || ntaps[k] > 16 || sub_h[k] < 1 || sub_w[k] < 1 || !wt[k]) return FSV_ERR_UNSUPPORTED;
    for (int t = 0; t < ntaps[k]; ++t)
      if (ty[k * 16 + t] < -8 || ty[k * 16 + t] > 7 || tx[k * 16 + t] < -8 || tx[k * 16 + t] > 7) return FSV_ERR_UNSUPPORTED;
    ConvP& p = q.c[k];
    p.in = in; p.wt = wt[k]; p.bias = nullptr; p.res = nullptr; p.out = out; p.wscale = wscale;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = sub_h[k]; p.OW = sub_w[k]; p.Cout = Cout;
    p.K = ntaps[k] * Cin; p.nchunks = fsv_cdiv(p.K, FSV_BK); p.ldw = ldw;
    p.sy = 1; p.sx = 1; p.ntaps = ntaps[k];
    fsv_db_pack_taps(ty + k * 16, tx + k * 16, ntaps[k], p.taps_lo, p.taps_hi);
    p.outH = outH; p.outW = outW; p.osy = 2; p.osx = 2; p.ooy = py[k]; p.oox = px[k]; p.dense_out = 0;
    p.w_bstride = w_bstride ? w_bstride[k] : 0; p.b_bstride = 0; p.per_sample = per_sample ? 1 : 0;
    p.act = FSV_ACT_NONE; p.scale = 1.f;
    p.Mz = per_sample ? sub_h[k] * sub_w[k] : N * sub_h[k] * sub_w[k];
    int t = 0, ns = 1;
    if (fsv_conv_plan(p.Mz, Cout, p.nchunks, nsamp, -1, 0, &t, &ns)) return FSV_ERR_BAD_ARG;
    if (ns != 1) return FSV_ERR_UNSUPPORTED;
    p.nsplit = 1;
    if (p.Mz > max_mz) { max_mz = p.Mz; tile = t; }
  }
  dim3 block(256);
  const int pf = prefetch == 2 ? 2 : 1;
  if (tile == 4) {
    dim3 g(fsv_cdiv(max_mz, 64), fsv_cdiv(Cout, 64), 4 * nsamp);
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_db4_kernel<64, 64, 2, 2, 2>), g, block, stream, q);
    else FSV_LAUNCH((fsv_conv_igemm_db4_kernel<64, 64, 2, 2, 1>), g, block, stream, q);
  } else if (tile == 9) {
    dim3 g(fsv_cdiv(max_mz, 64), fsv_cdiv(Cout, 128), 4 * nsamp);
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_db4_kernel<64, 128, 2, 2, 2>), g, block, stream, q);
    else FSV_LAUNCH((fsv_conv_igemm_db4_kernel<64, 128, 2, 2, 1>), g, block, stream, q);
  } else if (tile == 1) {
    dim3 g(fsv_cdiv(max_mz, 128), fsv_cdiv(Cout, 64), 4 * nsamp);
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_db4_kernel<128, 64, 2, 2, 2>), g, block, stream, q);
    else FSV_LAUNCH((fsv_conv_igemm_db4_kernel<128, 64, 2, 2, 1>), g, block, stream, q);
  } else {
    return FSV_ERR_UNSUPPORTED;          // 128x128 / thin tiles have no double-buffered variant
  }
  return fsv_check_launch();
}


// Split-K convolution / linear layer WITHOUT zero fill, atomics and finishing pass (opt-in): same arguments as
// fsv_conv_gather_fwd (no `accumulate` mode) plus the workspace - skw: nsplit x tiles x BM*BN floats, sk_tickets: one int per
// output tile, zero between launches.  *ws_floats / *n_tickets: when skw is null the call only reports the sizes this launch
// needs (0 / 0 and FSV_ERR_UNSUPPORTED when the plan does not split or its tile has no double-buffered variant).
extern "C" int fsv_conv_gather_fwd_splitws(const float* in, const float* wt, const float* bias, const float* res, float* out,
                                           int N, int H, int W, int Cin, int OH, int OW, int Cout,
                                           int ntaps, const int* ty, const int* tx, int sy, int sx,
                                           int outH, int outW, int osy, int osx, int ooy, int oox,
                                           int ldw, long long w_bstride, long long b_bstride, int per_sample,
                                           int act, float scale, const float* wscale, float* skw, int* sk_tickets,
                                           long long* ws_floats, int* n_tickets, int prefetch, hipStream_t stream) {
  if (!in || !wt || !out || ntaps < 1 || ntaps > 16 || N < 1 || Cin < 1 || Cout < 1 || !ws_floats || !n_tickets) return FSV_ERR_BAD_ARG;
  *ws_floats = 0; *n_tickets = 0;
  if ((Cin & 3) != 0) return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((ldw & 3) != 0 || ldw < Cout) return FSV_ERR_BAD_ARG;
  ConvP p;
  p.in = in; p.wt = wt; p.bias = bias; p.res = res; p.out = out; p.wscale = wscale;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.nchunks = fsv_cdiv(p.K, FSV_BK); p.ldw = ldw;
  p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  fsv_db_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi);
  p.outH = outH; p.outW = outW; p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
  p.dense_out = (osy == 1 && osx == 1 && ooy == 0 && oox == 0 && outH == OH && outW == OW) ? 1 : 0;
  p.w_bstride = w_bstride; p.b_bstride = b_bstride; p.per_sample = per_sample ? 1 : 0;
  p.act = act; p.scale = scale;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  const int nsamp = per_sample ? N : 1;
  int tile = 0, nsplit = 1;
  if (fsv_conv_plan(p.Mz, Cout, p.nchunks, nsamp, -1, 0, &tile, &nsplit)) return FSV_ERR_BAD_ARG;
  if (nsplit < 2 || (tile != 4 && tile != 9 && tile != 1)) return FSV_ERR_UNSUPPORTED;
  p.nsplit = nsplit;
  const int bm = (tile == 1) ? 128 : 64, bn = (tile == 9) ? 128 : 64;
  const int gx = fsv_cdiv(p.Mz, bm), gy = fsv_cdiv(Cout, bn);
  *n_tickets = gx * gy * nsamp;
  *ws_floats = (long long)nsplit * gx * gy * nsamp * bm * bn;
  if (!skw || !sk_tickets) return FSV_OK;            // size query
  dim3 g(gx, gy, nsamp * nsplit), block(256);
  const int pf = prefetch == 2 ? 2 : 1;
  if (tile == 4) {
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<64, 64, 2, 2, 2>), g, block, stream, p, skw, sk_tickets);
    else FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<64, 64, 2, 2, 1>), g, block, stream, p, skw, sk_tickets);
  } else if (tile == 9) {
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<64, 128, 2, 2, 2>), g, block, stream, p, skw, sk_tickets);
    else FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<64, 128, 2, 2, 1>), g, block, stream, p, skw, sk_tickets);
  } else {
    if (pf == 2) FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<128, 64, 2, 2, 2>), g, block, stream, p, skw, sk_tickets);
    else FSV_LAUNCH((fsv_conv_igemm_dbk_kernel<128, 64, 2, 2, 1>), g, block, stream, p, skw, sk_tickets);
  }
  return fsv_check_launch();
}
