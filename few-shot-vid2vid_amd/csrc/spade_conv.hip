// SPADE modulation of the shortcut branch fused with its 1x1 convolution: ONE kernel for
//     x_s = conv_s(bn_s(x, maps))            (reference models/networks/architecture.py:95,103-108;
//                                              bn_s = SPADE.forward, normalization.py:37-52; conv_s: 1x1, bias-free, spectral norm)
// The modulated tensor hs = bn_s(x) never reaches HBM in a forward pass that keeps no graph (the D step's generator pass,
// inference); a training forward may ask for it as a side output (conv_s' weight gradient reads it).
//
// How the two GEMMs chain without a trip through LDS.  The modulation kernel (spade.hip) computes gamma / beta as
// D[pixel][channel] = map[pixel][k] x W[k][channel]: a lane then holds ONE channel of 16 pixels, which is the wrong way round to
// feed a second matrix instruction (its A operand wants lane = row = pixel).  Here the first GEMM is issued with its operands
// swapped - same LDS images, same fragment reads, v_mfma(W fragment, map fragment) - so the accumulators hold
// D[channel][pixel]: lane l owns pixel (l & 31) of the wave's 32-pixel block and its 16 registers are the channels
// c(r) = (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the wave's 32-channel block.  After the modulation (registers only) register r of the
// two half-waves IS the A fragment of a 32x32x2 step of the second GEMM over the channel pair (c(r, 0), c(r, 1)): 16 matrix
// instructions per 32 output channels, with the rows of the conv_s weight picked in the same permuted order.  The 64 channels of a
// channel tile are split over two waves (the K dimension of the second GEMM): the two partial [32 pixels][Cout] results meet in
// LDS once per pixel tile, each wave finishing and storing half of the rows.
//
// Everything else follows spade.hip: the K chunks of all maps, channel tiles and pixel tiles form one flat sequence whose loads run
// one chunk ahead (buffer loads with hardware zero fill, double-buffered LDS, one barrier per chunk), workgroups walk pixel tiles,
// x of the next (pixel tile, channel tile) is requested as soon as the first modulation has consumed the current one.  Per-channel
// constants (statistics, gamma / beta biases) live in LDS: in this layout they vary per REGISTER, a b128 broadcast read fetches the four
// consecutive channels of a register quad.  The conv_s weight rows a lane needs stay in registers for the whole launch.
// x is read as 16-byte vectors (a lane's register quad = four consecutive channels of one pixel), through the nearest-x2
// up-sampling index when up != 0 (generator.py:124 folded in, as in spade.hip).
#include <type_traits>
#include "conv_igemm.h"

#define FSV_SC_BK 32
#define FSV_SC_MAXMAPS 3

struct SpadeConvP {
  const float* x;         // [N][HW or HW/4][C]
  const float* mean;      // [C] (+ z * stat_bstride)
  const float* rstd;
  float* hs;              // optional: the modulated tensor [N][HW][C]
  float* xs;              // [N][HW][Cout]
  const float* map[FSV_SC_MAXMAPS];   // [N][HW][Ch_k]
  const float* wg[FSV_SC_MAXMAPS];    // K-major [Kpad_k][ldw] (+ z * w_bstride_k)
  const float* wb[FSV_SC_MAXMAPS];
  const float* bg[FSV_SC_MAXMAPS];    // [C] (+ z * b_bstride_k)
  const float* bb[FSV_SC_MAXMAPS];
  int ch[FSV_SC_MAXMAPS];
  long long w_bstride[FSV_SC_MAXMAPS];
  long long b_bstride[FSV_SC_MAXMAPS];
  int nmaps, N, HW, C, ldw;
  long long stat_bstride;
  int W, up;
  const float* ws;        // conv_s weight, K-major [>= C rows][ldws]  (F16: N-major IEEE half [>= Cout rows][ldws], K contiguous)
  const float* wscale;    // optional device scalar on the result (spectral-norm 1 / sigma)
  int Cout, ldws;
};

// F16 (the `--amp` path): the gamma / beta GEMMs on v_mfma_f32_32x32x16_f16 exactly as the F16 form of fsv_spade_mod_kernel - maps
// IEEE half ([N][HW][Ch]), wg / wb pointing at the gamma rows / beta rows of the N-major half operand of fsv_spade_prep_h (row length
// ceil32(Ch) halves, w_bstride in halves), tiles [rows][32 k] halves with the four 16-byte slots XOR-swizzled by (row >> 2) & 3 - again
// with the operands swapped.  The modulated values are rounded to half ONCE (what the two-launch form does at its store) and the
// eight registers (8 t .. 8 t + 7) of the two half-waves are the A fragment of one 32x32x16 step over 16 channels; the conv_s weight
// arrives as the N-major half operand of the half-precision convolutions (hconv.prep_weight_h), accumulation and epilogue fp32, the
// side output hs is written as half.
typedef _Float16 fsv_sc_h16;
typedef _Float16 fsv_sc_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fsv_sc_h16x4 __attribute__((ext_vector_type(4)));

// NCT channel tiles of 64 (C = 64 NCT), TN2 output column tiles of 32 (Cout = 32 TN2)
template <int NCT, int TN2, bool F16 = false>
__global__ __launch_bounds__(256, 2) void fsv_spade_conv_s_kernel(SpadeConvP p) {
  constexpr int BM = 64, BN = 64, BK = FSV_SC_BK;
  constexpr int A_ST = BM * BK, B_ST = BK * BN;
  constexpr int NPA = 2, RPA = 32;                      // A: 8 work-items per row (one quad of 4 k each), 32 rows per pass
  constexpr int QB = BN / 4, RPB = 256 / QB, NPB = BK / RPB;
  constexpr int NKIND = 2 + 2 * FSV_SC_MAXMAPS;         // mean, rstd, (gamma bias, beta bias) per map
  constexpr int CT = 64 * NCT;
  __shared__ __attribute__((aligned(16))) float smem[(F16 ? 1 : 2) * (A_ST + 2 * B_ST)];
  __shared__ __attribute__((aligned(16))) float cst[NKIND * CT];
  __shared__ __attribute__((aligned(16))) float xch[4 * 8 * TN2 * 64];
  float* const As = smem;
  float* const Bs = smem + 2 * A_ST;
  fsv_sc_h16* const Ah = reinterpret_cast<fsv_sc_h16*>(smem);       // F16: [2 buffers][BM][32] halves, then [2][2][BN][32]
  fsv_sc_h16* const Bh = Ah + 2 * A_ST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hr0 = tid >> 2, hs4 = tid & 3;              // F16 loads: 64 rows x 4 slots of 16 bytes per pass
  const int wm = wave >> 1, wn = wave & 1;              // pixel half / channel half of the 64 x 64 tile
  const int lrow = lane & 31, lk = lane >> 5;
  const int z = blockIdx.z;
  const int tile_step = gridDim.x * BM;
  const int kq = tid & 7, ar0 = tid >> 3;
  const int bq = tid % QB, br0 = tid / QB;
  const long long pix0 = (long long)z * p.HW;
  const int C = p.C;

  // ---- per-channel constants into LDS, conv_s weight rows into registers ---------------------------------------------------------
  for (int i = tid; i < NKIND * CT; i += 256) {
    const int kind = i / CT, c = i - kind * CT;
    float v = 0.f;
    if (kind == 0) v = (p.mean + z * p.stat_bstride)[c];
    else if (kind == 1) v = (p.rstd + z * p.stat_bstride)[c];
    else {
      const int k = (kind - 2) >> 1;
      if (k < p.nmaps) v = (((kind & 1) ? p.bb[k] : p.bg[k]) + z * p.b_bstride[k])[c];
    }
    cst[i] = v;
  }
  float wsf[F16 ? 1 : NCT][F16 ? 1 : 16][TN2];
  fsv_sc_h16x8 wsh[F16 ? NCT : 1][2][TN2];
  if constexpr (F16) {
    // element e of step t: channel (r & 3) + 8 (r >> 2) + 4 lk with r = 8 t + e, i.e. two runs of four consecutive k of row n
    const fsv_sc_h16* wh = reinterpret_cast<const fsv_sc_h16*>(p.ws);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) {
          const fsv_sc_h16* row = wh + (long long)(32 * jn + lrow) * p.ldws + 64 * ct + 32 * wn + 16 * t + 4 * lk;
          const fsv_sc_h16x4 lo = *reinterpret_cast<const fsv_sc_h16x4*>(row), hi = *reinterpret_cast<const fsv_sc_h16x4*>(row + 8);
          fsv_sc_h16x8 v;
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
          wsh[ct][t][jn] = v;
        }
  } else {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn)
          wsf[ct][r][jn] = p.ws[(long long)(64 * ct + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * lk) * p.ldws + 32 * jn + lrow];
  }
  const float sc = p.wscale ? *p.wscale : 1.f;

  // ---- x of one (pixel tile, channel tile): four 16-byte vectors per lane ---------------------------------------------------------
  const long long xpix_n = p.up ? (p.HW >> 2) : p.HW;
  const fsv_buf xbuf = fsv_make_buf(p.x + (long long)z * xpix_n * C, xpix_n * C * 4);
  float4 xq[4];
  auto load_x = [&](int tb0, int ct) {
    const int m = tb0 + 32 * wm + lrow;
    int sp = m;
    if (p.up) {
      const int y = m / p.W, xx = m - y * p.W;
      sp = (y >> 1) * (p.W >> 1) + (xx >> 1);
    }
    const bool ok = m < p.HW;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      xq[q] = fsv_buf_load4(xbuf, ok ? (unsigned)((sp * C + 64 * ct + 32 * wn + 8 * q + 4 * lk) * 4) : FSV_BUF_OOB);
  };

  // ---- the flat chunk sequence: (pixel tile, channel tile, map, K chunk) -----------------------------------------------------------
  int nch[FSV_SC_MAXMAPS];
#pragma unroll
  for (int k = 0; k < FSV_SC_MAXMAPS; ++k) nch[k] = (k < p.nmaps) ? (p.ch[k] + BK - 1) / BK : 0;
  int ld_k = 0, ld_c = 0, ld_ct = 0, ld_bm0 = blockIdx.x * BM;
  auto advance_loader = [&]() {
    ++ld_c;
    const int n_k = ld_k == 0 ? nch[0] : (ld_k == 1 ? nch[1] : nch[2]);
    if (ld_c >= n_k) {
      ld_c = 0; ++ld_k;
      if (ld_k >= p.nmaps) {
        ld_k = 0; ++ld_ct;
        if (ld_ct >= NCT) { ld_ct = 0; ld_bm0 += tile_step; }
      }
    }
  };
  float4 areg[NPA], breg[2][NPB];
  auto issue_loads = [&]() {
    const int k = ld_k;
    const int Ch = p.ch[k];
    const fsv_buf abuf = fsv_make_buf(p.map[k] + pix0 * Ch, (long long)p.HW * Ch * 4);
    const int kk = ld_c * BK + kq * 4;
    const bool live = ld_bm0 < p.HW;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int m = ld_bm0 + ar0 + i * RPA;
      const bool ok = (kk < Ch) & (m < p.HW);
      areg[i] = fsv_buf_load4(abuf, ok ? (unsigned)((m * Ch + kk) * 4) : FSV_BUF_OOB);
    }
    const long long wbytes = (long long)((Ch + BK - 1) / BK) * BK * p.ldw * 4;
    const fsv_buf gbuf = fsv_make_buf(p.wg[k] + z * p.w_bstride[k], wbytes);
    const fsv_buf bbuf = fsv_make_buf(p.wb[k] + z * p.w_bstride[k], wbytes);
    const int bcol = 64 * ld_ct + bq * 4;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int kr = ld_c * BK + br0 + i * RPB;
      const unsigned off = live ? (unsigned)((kr * p.ldw + bcol) * 4) : FSV_BUF_OOB;
      breg[0][i] = fsv_buf_load4(gbuf, off);
      breg[1][i] = fsv_buf_load4(bbuf, off);
    }
    advance_loader();
  };
  auto store_chunk = [&](int buf) {
    float* a_dst = As + buf * A_ST;
    float* b_dst = Bs + buf * (2 * B_ST);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int r = ar0 + i * RPA;
      // quad (k0 k1 k2 k3) stored as (k0 k2 | k1 k3), rows with bit 4 set as (k1 k3 | k0 k2): one ds_read_b64 per lane and k-group
      // half (spade.hip / conv_igemm.hip)
      const bool hi = (r >> 4) & 1;
      float4 v;
      v.x = hi ? areg[i].y : areg[i].x; v.y = hi ? areg[i].w : areg[i].z;
      v.z = hi ? areg[i].x : areg[i].y; v.w = hi ? areg[i].z : areg[i].w;
      *reinterpret_cast<float4*>(&a_dst[r * BK + ((kq ^ ((r >> 1) & 7)) << 2)]) = v;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        *reinterpret_cast<float4*>(&b_dst[q * B_ST + (br0 + i * RPB) * BN + bq * 4]) = breg[q][i];
  };

  float4 hareg, hbreg[2];
  auto issue_loads_h = [&]() {
    const int k = ld_k;
    const int Ch = p.ch[k];
    const fsv_buf abuf = fsv_make_buf(reinterpret_cast<const fsv_sc_h16*>(p.map[k]) + pix0 * Ch, (long long)p.HW * Ch * 2);
    const int kk = ld_c * BK + hs4 * 8;
    const int m = ld_bm0 + hr0;
    hareg = fsv_buf_load4(abuf, ((kk < Ch) & (m < p.HW)) ? (unsigned)((m * Ch + kk) * 2) : FSV_BUF_OOB);
    const int ldk = (Ch + 31) & ~31;
    const long long wbytes = (long long)C * ldk * 2;
    const int c = 64 * ld_ct + hr0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const fsv_sc_h16* w = reinterpret_cast<const fsv_sc_h16*>(q ? p.wb[k] : p.wg[k]) + z * p.w_bstride[k];
      const fsv_buf wbuf = fsv_make_buf(w, wbytes);
      hbreg[q] = fsv_buf_load4(wbuf, (ld_bm0 < p.HW) ? (unsigned)((c * ldk + kk) * 2) : FSV_BUF_OOB);
    }
    advance_loader();
  };
  auto store_chunk_h = [&](int buf) {
    fsv_sc_h16* a_dst = Ah + buf * A_ST;
    fsv_sc_h16* b_dst = Bh + buf * (2 * B_ST);
    const int slot = ((hs4 ^ (hr0 >> 2)) & 3) << 3;
    *reinterpret_cast<float4*>(&a_dst[hr0 * BK + slot]) = hareg;
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(&b_dst[q * B_ST + hr0 * BK + slot]) = hbreg[q];
  };

  f32x16 acc[2];            // gamma^T, beta^T of the wave's [32 channels][32 pixels] block
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const int a_row = wm * 32 + lrow;                     // this lane's pixel row of the map tile
  const int a_off = a_row * BK + 2 * (lk ^ ((a_row >> 4) & 1));
  const int a_swz = (a_row >> 1) & 7;
  const int b_off = lk * BN + wn * 32 + lrow;

  auto read_group = [&](const float* a_src, const float* b_src, int g, float2 (&a4)[2], float (&b)[4][2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) a4[q] = *reinterpret_cast<const float2*>(&a_src[a_off + (((2 * g + q) ^ a_swz) << 2)]);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int q = 0; q < 2; ++q) b[s4][q] = b_src[q * B_ST + b_off + (8 * g + 2 * s4) * BN];
  };
  auto mma_group = [&](const float2 (&a4)[2], const float (&b)[4][2]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float2 v = a4[s4 >> 1];
      const float a = (s4 & 1) ? v.y : v.x;
      // operands swapped: rows of D = channels (the weight fragment), columns = pixels (the map fragment)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s4][q], a, acc[q], 0, 0, 0);
    }
  };

  int buf = 0;
  auto chunk = [&]() {
    if constexpr (F16) {
      issue_loads_h();
      const fsv_sc_h16* a_src = Ah + buf * A_ST;
      const fsv_sc_h16* b_src = Bh + buf * (2 * B_ST);
      const int ar = wm * 32 + lrow, br = wn * 32 + lrow;
      fsv_sc_h16x8 fa[2], fb[2][2];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        fa[st] = *reinterpret_cast<const fsv_sc_h16x8*>(&a_src[ar * BK + ((((2 * st + lk) ^ (ar >> 2)) & 3) << 3)]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          fb[st][q] = *reinterpret_cast<const fsv_sc_h16x8*>(&b_src[q * B_ST + br * BK + ((((2 * st + lk) ^ (br >> 2)) & 3) << 3)]);
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int q = 0; q < 2; ++q)      // operands swapped: rows of D = channels, columns = pixels
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[st][q], fa[st], acc[q], 0, 0, 0);
      store_chunk_h(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      return;
    }
    issue_loads();                      // past the end: every lane is out of range -> zeros, never used
    const float* a_src = As + buf * A_ST;
    const float* b_src = Bs + buf * (2 * B_ST);
    if constexpr (NCT * TN2 >= 4) {
      // 64 weight registers: one fragment set (the two resident workgroups cover each other's LDS latency)
      float2 fa[2];
      float fb[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        read_group(a_src, b_src, g, fa, fb);
        mma_group(fa, fb);
        if (g == 2) { FSV_SCHED_FENCE(); store_chunk(buf ^ 1); FSV_SCHED_FENCE(); }
      }
      __syncthreads();
      buf ^= 1;
      return;
    }
    float2 fa[2][2];
    float fb[2][4][2];
    read_group(a_src, b_src, 0, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    read_group(a_src, b_src, 1, fa[1], fb[1]);
    FSV_SCHED_FENCE();
    mma_group(fa[0], fb[0]);
    FSV_SCHED_FENCE();
    read_group(a_src, b_src, 2, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    mma_group(fa[1], fb[1]);
    FSV_SCHED_FENCE();
    read_group(a_src, b_src, 3, fa[1], fb[1]);
    FSV_SCHED_FENCE();
    mma_group(fa[0], fb[0]);
    FSV_SCHED_FENCE();
    store_chunk(buf ^ 1);
    FSV_SCHED_FENCE();
    mma_group(fa[1], fb[1]);
    __syncthreads();
    buf ^= 1;
  };

  // running value of the normalised + modulated activation: register r = channel c(r) of the lane's pixel
  f32x16 outv;
  auto cst4 = [&](int kind, int ct, int q) {
    return *reinterpret_cast<const float4*>(&cst[kind * CT + 64 * ct + 32 * wn + 8 * q + 4 * lk]);
  };
  auto modulate = [&](int k, bool first, int ct) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g4 = cst4(2 + 2 * k, ct, q), b4 = cst4(3 + 2 * k, ct, q);
      const float gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq4[4] = {b4.x, b4.y, b4.z, b4.w};
      float mu4[4] = {0.f, 0.f, 0.f, 0.f}, rs4[4] = {0.f, 0.f, 0.f, 0.f};
      if (first) {
        const float4 m4 = cst4(0, ct, q), r4 = cst4(1, ct, q);
        mu4[0] = m4.x; mu4[1] = m4.y; mu4[2] = m4.z; mu4[3] = m4.w;
        rs4[0] = r4.x; rs4[1] = r4.y; rs4[2] = r4.z; rs4[3] = r4.w;
      }
      const float xe[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * q + e;
        const float o = first ? (xe[e] - mu4[e]) * rs4[e] : outv[r];
        const float gk = acc[0][r] + gq[e];
        outv[r] = o * (1.f + gk) + (acc[1][r] + bq4[e]);
        acc[0][r] = 0.f; acc[1][r] = 0.f;
      }
    }
  };

  // ---- the workgroup's pixel tiles ----------------------------------------------------------------------------------------------------
  // (the widest fp32 form - 128 channels to 64: 64 weight registers next to 32 + 32 accumulators - requests x of a channel tile at
  // the top of that tile's own chunks instead of one tile ahead: 16 registers less across the second GEMM, no spills)
  constexpr bool XLATE = !F16 && NCT * TN2 >= 4;
  if constexpr (!XLATE) load_x(blockIdx.x * BM, 0);
  if constexpr (F16) { issue_loads_h(); store_chunk_h(0); } else { issue_loads(); store_chunk(0); }
  __syncthreads();                       // (also publishes cst)
#pragma unroll 1
  for (int bm0 = blockIdx.x * BM; bm0 < p.HW; bm0 += tile_step) {
    f32x16 acc2[TN2];
#pragma unroll
    for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[jn][r] = 0.f;
    const int m = bm0 + 32 * wm + lrow;   // this lane's pixel
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if constexpr (XLATE) load_x(bm0, ct);
#pragma unroll
      for (int k = 0; k < FSV_SC_MAXMAPS; ++k) {
        if (k < p.nmaps) {
#pragma unroll 1
          for (int c = 0; c < nch[k]; ++c) chunk();
          modulate(k, k == 0, ct);
          if (k == 0 && !XLATE) {         // x of the next (pixel tile, channel tile)
            if (ct + 1 < NCT) load_x(bm0, ct + 1); else load_x(bm0 + tile_step, 0);
          }
        }
      }
      if constexpr (F16) {
        fsv_sc_h16x8 ha[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) ha[t][e] = (fsv_sc_h16)outv[8 * t + e];       // the one rounding of the modulated value
        if (p.hs && m < p.HW) {           // side output (half) for the weight gradient of conv_s
          fsv_sc_h16* hrow = reinterpret_cast<fsv_sc_h16*>(p.hs) + (pix0 + m) * C + 64 * ct + 32 * wn + 4 * lk;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            fsv_sc_h16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ha[q >> 1][4 * (q & 1) + e];
            *reinterpret_cast<fsv_sc_h16x4*>(hrow + 8 * q) = v;
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int jn = 0; jn < TN2; ++jn)
            acc2[jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[t], wsh[ct][t][jn], acc2[jn], 0, 0, 0);
      } else {
        if (p.hs && m < p.HW) {           // side output for the weight gradient of conv_s (training forward)
          float* hrow = p.hs + (pix0 + m) * C + 64 * ct + 32 * wn + 4 * lk;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(hrow + 8 * q) = make_float4(outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]);
        }
        // second GEMM: register r of the two half-waves = the channel pair (c(r, 0), c(r, 1)) of 32 pixels
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int jn = 0; jn < TN2; ++jn)
            acc2[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(outv[r], wsf[ct][r][jn], acc2[jn], 0, 0, 0);
      }
    }
    // the two channel halves of a pixel block meet in LDS: wave (wm, wn) finishes rows with (r >> 3) == wn and hands the others over
    auto hand_over = [&](auto WNC) {
      constexpr int w_ = decltype(WNC)::value;
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int t = 0; t < 8; ++t) xch[((wave * TN2 + jn) * 8 + t) * 64 + lane] = acc2[jn][8 * (1 - w_) + t];
    };
    auto finish = [&](auto WNC) {
      constexpr int w_ = decltype(WNC)::value;
      float* xs_z = p.xs + pix0 * p.Cout;
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int r2 = 8 * w_ + t;
          const int mm = bm0 + 32 * wm + (r2 & 3) + 8 * (r2 >> 2) + 4 * lk;
          const float other = xch[(((wave ^ 1) * TN2 + jn) * 8 + t) * 64 + lane];
          // (channel half 0) + (channel half 1) in both waves
          const float v = w_ == 0 ? acc2[jn][r2] + other : other + acc2[jn][r2];
          if (mm < p.HW) xs_z[(long long)mm * p.Cout + 32 * jn + lrow] = v * sc;
        }
    };
    if (wn == 0) hand_over(std::integral_constant<int, 0>{}); else hand_over(std::integral_constant<int, 1>{});
    __syncthreads();
    if (wn == 0) finish(std::integral_constant<int, 0>{}); else finish(std::integral_constant<int, 1>{});
  }
}

extern "C" {

// 1 when fsv_spade_conv_s_fwd has a kernel for this geometry
int fsv_spade_conv_s_supported(int C, int Cout, int nmaps) {
  return (C == 64 || C == 128) && (Cout == 32 || Cout == 64) && nmaps >= 1 && nmaps <= FSV_SC_MAXMAPS;
}

// x_s = conv_s(bn_s(x)) in one launch (see the file comment).  Operands of the modulation as fsv_spade_mod_fwd (spade.hip; act is
// FSV_ACT_NONE: bn_s has no activation, architecture.py:103); ws = the K-major forward operand of conv_s' 1x1 weight ([>= C rows][ldws],
// fsv_prep_weight), wscale = optional device scalar (1 / sigma), xs [N][HW][Cout].  hs (optional) receives the modulated tensor.
// FSV_ERR_UNSUPPORTED for geometries without a kernel (fsv_spade_conv_s_supported).
static int fsv_spade_conv_s_impl(const float* x, const float* mean, const float* rstd, float* hs, float* xs,
                         int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                         const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                         const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int W, int up,
                         const float* ws, int ldws, int Cout, const float* wscale, bool f16, hipStream_t stream) {
  if (!x || !mean || !rstd || !xs || !ws || !maps || !wg || !wb || !bg || !bb || !ch || N < 1 || HW < 1 || (ldw & 3) ||
      ldws < (f16 ? C : Cout))
    return FSV_ERR_BAD_ARG;
  if (!fsv_spade_conv_s_supported(C, Cout, nmaps)) return FSV_ERR_UNSUPPORTED;
  if (up && (W < 2 || (W & 1) || HW % W != 0 || ((HW / W) & 1))) return FSV_ERR_BAD_ARG;
  if ((long long)HW * C * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  SpadeConvP p;
  p.x = x; p.mean = mean; p.rstd = rstd; p.hs = hs; p.xs = xs;
  for (int k = 0; k < FSV_SC_MAXMAPS; ++k) {
    const bool on = k < nmaps;
    p.map[k] = on ? maps[k] : nullptr; p.wg[k] = on ? wg[k] : nullptr; p.wb[k] = on ? wb[k] : nullptr;
    p.bg[k] = on ? bg[k] : nullptr; p.bb[k] = on ? bb[k] : nullptr;
    p.ch[k] = on ? ch[k] : 0; p.w_bstride[k] = on ? w_bstride[k] : 0; p.b_bstride[k] = on ? b_bstride[k] : 0;
    if (on && (!maps[k] || !wg[k] || !wb[k] || !bg[k] || !bb[k] || ch[k] < 1 || (ch[k] & (f16 ? 7 : 3)))) return FSV_ERR_UNSUPPORTED;
    if (on && (long long)HW * ch[k] * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  }
  p.nmaps = nmaps; p.N = N; p.HW = HW; p.C = C; p.ldw = ldw; p.stat_bstride = stat_bstride;
  p.W = up ? W : 1; p.up = up ? 1 : 0;
  p.ws = ws; p.wscale = wscale; p.Cout = Cout; p.ldws = ldws;
  // as many pixel-tile walkers as stay resident (two workgroups per CU)
  const int ntiles = fsv_cdiv(HW, 64);
  long long cap = (256ll * 2) / N;
  const char* e = getenv("FSV_SPADE_MAX_GX");               // tests: a multi-tile walk on a small map
  if (e && atoi(e) > 0) cap = atoi(e);
  if (cap < 1) cap = 1;
  dim3 g((unsigned)(ntiles < cap ? ntiles : cap), 1, N);
  if (f16) {
    if (C == 64 && Cout == 32) FSV_LAUNCH((fsv_spade_conv_s_kernel<1, 1, true>), g, dim3(256), stream, p);
    else if (C == 64) FSV_LAUNCH((fsv_spade_conv_s_kernel<1, 2, true>), g, dim3(256), stream, p);
    else if (Cout == 32) FSV_LAUNCH((fsv_spade_conv_s_kernel<2, 1, true>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_conv_s_kernel<2, 2, true>), g, dim3(256), stream, p);
  } else {
    if (C == 64 && Cout == 32) FSV_LAUNCH((fsv_spade_conv_s_kernel<1, 1>), g, dim3(256), stream, p);
    else if (C == 64) FSV_LAUNCH((fsv_spade_conv_s_kernel<1, 2>), g, dim3(256), stream, p);
    else if (Cout == 32) FSV_LAUNCH((fsv_spade_conv_s_kernel<2, 1>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_conv_s_kernel<2, 2>), g, dim3(256), stream, p);
  }
  return fsv_check_launch();
}

int fsv_spade_conv_s_fwd(const float* x, const float* mean, const float* rstd, float* hs, float* xs,
                         int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                         const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                         const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int W, int up,
                         const float* ws, int ldws, int Cout, const float* wscale, hipStream_t stream) {
  return fsv_spade_conv_s_impl(x, mean, rstd, hs, xs, nmaps, maps, wg, wb, bg, bb, ch, w_bstride, b_bstride, N, HW, C, ldw,
                               stat_bstride, W, up, ws, ldws, Cout, wscale, false, stream);
}

// the `--amp` form: maps / wg / wb as fsv_spade_mod_fwd_h with the f16 GEMMs (IEEE half maps, N-major half gamma | beta operand of
// fsv_spade_prep_h; Ch % 8 == 0), ws_h = N-major half operand of conv_s ([>= Cout rows][ldws halves], K contiguous,
// fsv_hconv_prep_weight), hs_h (optional) receives the modulated tensor as half; xs stays fp32
int fsv_spade_conv_s_fwd_h(const float* x, const float* mean, const float* rstd, void* hs_h, float* xs,
                           int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                           const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                           const long long* b_bstride, int N, int HW, int C, long long stat_bstride, int W, int up,
                           const void* ws_h, int ldws, int Cout, const float* wscale, hipStream_t stream) {
  return fsv_spade_conv_s_impl(x, mean, rstd, reinterpret_cast<float*>(hs_h), xs, nmaps, reinterpret_cast<const float* const*>(maps),
                               reinterpret_cast<const float* const*>(wg), reinterpret_cast<const float* const*>(wb), bg, bb, ch,
                               w_bstride, b_bstride, N, HW, C, 0, stat_bstride, W, up, reinterpret_cast<const float*>(ws_h), ldws,
                               Cout, wscale, true, stream);
}

}  // extern "C"
