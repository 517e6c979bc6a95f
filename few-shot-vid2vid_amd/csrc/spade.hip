// Fused adaptive-SPADE modulation on the matrix cores.
//
// Reference semantics (models/networks/normalization.py:37-52, SPADE.forward):
//     out = BN_nofaffine(x)
//     for each non-None map k:   gamma_k = conv1x1(m_k; Wg_k) + bg_k ;  beta_k = conv1x1(m_k; Wb_k) + bb_k
//                                out = out * (1 + gamma_k) + beta_k
// followed (in SPADEResnetBlock, architecture.py:95-97) by leaky_relu(0.2) for bn_0 / bn_1 (not bn_s).
// Map 0 of the adaptive layers uses per-sample generated weights (batch_conv, base_network.py:56-71); the
// extra "spade_combine" maps use fixed weights (mlp_gamma2/3).
//
// One launch does all of it: for a tile of BM pixels x BN channels the gamma and beta 1x1 convolutions are
// two fp32 MFMA GEMMs that share the LDS-staged label-map tile, and the epilogue applies the BatchNorm
// denormalisation, the (1 + gamma) * . + beta modulation of every map in sequence and the activation before
// the single write of h.  gamma/beta never touch HBM.  Layout NHWC; grid.z = sample so that per-sample and
// shared weights can be mixed (a per-map batch stride of 0 means shared).
#include "conv_igemm.h"

#define FSV_SP_BK 32
#define FSV_SP_MAXMAPS 3

#define FSV_SP_SITES 2
struct SpadeP {
  const float* x;         // [N][HW][C]
  const float* mean;      // [C] (or [N][C] when stat_bstride != 0)
  const float* rstd;
  float* h[FSV_SP_SITES]; // [N][HW][C] per norm site
  const float* map[FSV_SP_MAXMAPS];    // [N][HW][Ch_k]
  const float* wg[FSV_SP_SITES][FSV_SP_MAXMAPS];     // K-major [Kpad_k][ldw] (+ z * w_bstride_k)
  const float* wb[FSV_SP_SITES][FSV_SP_MAXMAPS];
  const float* bg[FSV_SP_SITES][FSV_SP_MAXMAPS];     // [C] (+ z * b_bstride_k)
  const float* bb[FSV_SP_SITES][FSV_SP_MAXMAPS];
  int ch[FSV_SP_MAXMAPS];
  long long w_bstride[FSV_SP_SITES][FSV_SP_MAXMAPS];
  long long b_bstride[FSV_SP_SITES][FSV_SP_MAXMAPS];
  int nmaps;
  int N, HW, C, ldw;
  long long stat_bstride;
  int act[FSV_SP_SITES];
  // backward twin (BWD = true): upstream gradient in, d(gamma|beta) per map ([P][2C]: gamma columns [0, C), beta [C, 2C)) and
  // d(xhat) out; h is not written
  const float* dh;
  float* dgb[FSV_SP_MAXMAPS];
  float* dxhat;
  int h_half;             // forward: h is written as IEEE half (the `--amp` path: its only consumers are convolutions that read half);
                          // backward twin: bit 0 - dh arrives as half, bit 1 - d(gamma|beta) is written as half
  // backward twin, optional: per-channel sums of d(gamma|beta) = the bias gradients, added (fp64 atomics) into the ZEROED buffer
  // dbsum + z * db_zstride[k] + k * 2C  ([gamma sums (C) | beta sums (C)] per map; db_zstride 0: summed over the batch) - the
  // column-sum pass over the [P][2C] tensors disappears
  double* dbsum;
  long long db_zstride[FSV_SP_MAXMAPS];
  int dbg;                // FSV_SPADE_DBG (in-box experiments only; 0 in production): 1 no output stores, 2 no x loads, 4 no dh loads
  int db_slots;           // power of two: pixel tile t adds into copy t % db_slots of the buffer (db_slot_stride doubles apart)
  long long db_slot_stride;
  int W, up;              // up = 1: x is the HALF-resolution tensor [N][H/2][W/2][C] and is read through the nearest-x2
                          // up-sampling index (generator.py:124 folded into this kernel: the up-sampled tensor is never written)
};

// source element of x for output pixel m of sample z (identity, or the parent pixel of the nearest x2 up-sampling)
__device__ __forceinline__ long long fsv_sp_xpix(const SpadeP& p, int z, int m) {
  if (!p.up) return (long long)z * p.HW + m;
  const int y = m / p.W, xx = m - y * p.W;
  return (long long)z * (p.HW >> 2) + (long long)(y >> 1) * (p.W >> 1) + (xx >> 1);
}

// NS norm sites per launch (NS = 2: bn_0 and bn_s of one SPADEResnetBlock, architecture.py:95-96,103 - the same x, the same
// statistics, the same maps, their own gamma / beta weights and activation: x, the statistics and the label-map tiles are read
// once for both, the map tile in LDS feeds four GEMMs instead of two).
// BWD = true is the backward twin: the same two GEMMs recompute gamma / beta of every map in registers (they never reach HBM
// in either direction), the forward chain o_0 = xhat, o_{k+1} = o_k (1 + g_k) + b_k is replayed keeping g_k and o_k, and the
// epilogue walks it backwards:  d = dh * act'(o_n);  for k = n-1 .. 0:  dbeta_k = d,  dgamma_k = d * o_k,  d *= (1 + g_k);
// dxhat = d.  (Round 1 materialised gamma | beta as [P][2C] per map with the gather-GEMM kernel and read them back in an
// element-wise pass: 236 MB read + 175 MB written per launch more than this.)
//
// Main loop (round 3: the structure of the gather-GEMM kernel, conv_igemm.hip).  The K chunks of ALL maps form one sequence
// (the full-resolution levels have one or two 32-wide chunks per map: a per-map loop never gets a pipeline going): the loads of
// chunk t + 1 - label-map rows and the gamma / beta weight rows of every site, through buffer descriptors with hardware zero
// fill for rows / columns that do not exist - are issued at the top of chunk t and stored into the OTHER LDS buffer behind its
// MFMAs (one barrier per chunk).  A image [BM rows][8 quads], quad q of row r in slot q ^ ((r >> 1) & 7): ds_write_b128 /
// ds_read_b128 (two reads feed the four k steps of a k-group), B images [32 k][BN] as they lie in HBM.  The modulation of a
// map (registers only) runs when its last chunk has been multiplied.
// Pixel tiles (round 4): a workgroup walks the tiles blockIdx.x, + gridDim.x, ... of its channel tile, and the chunk sequence runs
// ACROSS tiles - the first chunk of the next tile is loaded and stored behind the last MFMAs of this one, x of the next tile is
// requested as soon as the first modulation has consumed this tile's, the per-channel constants (statistics, gamma / beta biases)
// are loaded once.  At the full-resolution levels a tile is ONE chunk: with a workgroup per tile nothing overlapped - kernel
// arguments, one load round trip, four MFMAs, a bias round trip, stores; 6 us per workgroup, 60 us for 17 us worth of traffic at
// 512 K pixels (profiles/r04_notes.md).  The host launches as many workgroups as stay resident (fsv_sp_grid_x).
// F16 (the `--amp` path): the gamma / beta GEMMs on v_mfma_f32_32x32x16_f16 - the label maps arrive as IEEE half ([N][HW][Ch] halves),
// the weights as the N-major half operand of fsv_spade_prep_h (wcat_h [2C][Kh], K contiguous: p.wg / p.wb point into it, Kh =
// ceil32(Ch) halves per row; p.ldw unused), both tiles are [rows][32 k] halves (64-byte rows, the four 16-byte slots XOR-swizzled by (row >> 2) & 3: conflict-free
// ds_read_b128 fragments, csrc/conv_np.hip) staged through registers as 16-byte vectors; accumulation, normalisation, modulation and
// the backward chain stay fp32.  At the deep levels (512 ... 2048 pixels, Ch up to 1024) the fp32 form is MFMA-latency bound - 128
// workgroups, 32 chunks of 32 fp32 MFMAs each: 61 us for 4 GFLOP - which this removes.
template <int BM, int BN, int WM, int WN, int NS, bool BWD, bool F16 = false, int NM = FSV_SP_MAXMAPS>
__global__ __launch_bounds__(256, (NS == 2 || (BWD && NM > 1)) ? 2 : (BWD || !F16 || BM == 128) ? 3 : 4) void fsv_spade_mod_kernel(SpadeP p) {
  constexpr int BK = FSV_SP_BK;
  // NM: the most maps this instantiation handles (the backward twin keeps g_k and o_k of every map in registers: 32 per map;
  // the one-map form - every layer without warp_ref / spade_combine - has room to keep the next tile's x and this tile's dh in
  // flight across the chunks, the three-map form requests them late)
  constexpr bool XEARLY = NS == 1 && !BWD;
  constexpr bool DVEARLY = false;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int RPA = 256 / 8, NPA = BM / RPA;          // A: 8 work-items per row (one quad of 4 k each)
  constexpr int QB = BN / 4, RPB = 256 / QB, NPB = BK / RPB;
  constexpr int NB = NS * 2;                            // B images per chunk: (site, gamma | beta)
  constexpr int A_ST = BM * BK, B_ST = BK * BN;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(NPA >= 1 && NPB >= 1 && NPA * RPA == BM && NPB * RPB == BK, "tile");
  static_assert(!BWD || NS == 1, "the backward twin handles one site");
  static_assert(!F16 || NS == 1, "the half form handles one site");
  static_assert(FSV_SP_MAXMAPS == 3, "advance_loader selects among three maps");
  __shared__ __attribute__((aligned(16))) float smem[(F16 ? 1 : 2) * (A_ST + NB * B_ST)];
  float* const As = smem;
  float* const Bs = smem + 2 * A_ST;
  typedef _Float16 h16;
  typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
  h16* const Ah = reinterpret_cast<h16*>(smem);                  // F16: [2 buffers][BM][32] halves, then [2][NB][BN][32]
  h16* const Bh = Ah + 2 * A_ST;
  constexpr int HNPA = (BM + 63) / 64, HNPB = (BN + 63) / 64;    // 16-byte vectors per work-item: 4 per 64-byte row, 64 rows per pass
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hr0 = tid >> 2, hs = tid & 3;
  // Half tensors are written two channels per work-item: in the MFMA C / D layout a lane holds ONE channel of consecutive pixels,
  // its neighbour (lane ^ 1) the next channel of the same pixels - for a pair of pixels (p0, p1) the even lane stores (c, c + 1)
  // of p0 and the odd lane (c - 1, c) of p1, one exchange each way.  (2-byte stores: the backward twin with half d(gamma|beta) took
  // 219 us where the same launch writing twice the bytes as fp32 took 134.)  e0 / e1: element index of this lane's own value.
  auto store_pair_h = [&](h16* base, int e0, int e1, float v0, float v1, bool ok0, bool ok1) {
    const float n0 = __shfl_xor(v0, 1), n1 = __shfl_xor(v1, 1);
    const bool odd = (lane & 1) != 0;
    h16x2 pk;
    pk.x = (h16)(odd ? n1 : v0); pk.y = (h16)(odd ? v1 : n0);
    if (odd ? ok1 : ok0) *reinterpret_cast<h16x2*>(base + (odd ? e1 - 1 : e0)) = pk;
  };
  const int wm = wave / WN, wn = wave % WN;
  const int z = blockIdx.z;
  const int bn0 = blockIdx.y * BN;
  const int tile_step = gridDim.x * BM;   // a workgroup walks the pixel tiles blockIdx.x, + gridDim.x, ... (see the header comment)
  const int lrow = lane & 31, lk = lane >> 5;
  const int kq = tid & 7, ar0 = tid >> 3;
  const int bq = tid % QB, br0 = tid / QB;
  const int bcol = bn0 + bq * 4;
  const bool bcol_ok = bcol < p.C;        // columns >= C only feed output channels that are never stored
  const long long pix0 = (long long)z * p.HW;

  // ---- x (raw) of one pixel tile into registers: consumed by the tile's first modulation ------------------------------------------
  // A lane's 16 rows are four runs of four consecutive pixels (D layout: row = (r & 3) + 8 * (r >> 2) + 4 * lk) that start at a
  // multiple of 4: with W % 4 == 0 a run lies in one image row, so the up-sampling index costs one division per run.
  float xv[TM][TN][16];
  const long long xpix_n = p.up ? (p.HW >> 2) : p.HW;
  const fsv_buf xbuf = fsv_make_buf(p.x + (long long)z * xpix_n * p.C, xpix_n * p.C * 4);
  const bool runs = (p.W & 3) == 0;
  auto load_x = [&](int tb0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m0 = tb0 + wm * (TM * 32) + i * 32 + 8 * q + 4 * lk;
        int src[4];
        if (!p.up) {
#pragma unroll
          for (int e = 0; e < 4; ++e) src[e] = m0 + e;
        } else if (runs) {
          const int y = m0 / p.W, xx = m0 - y * p.W;
          const int b = (y >> 1) * (p.W >> 1) + (xx >> 1);
          src[0] = b; src[1] = b; src[2] = b + 1; src[3] = b + 1;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int y = (m0 + e) / p.W, xx = (m0 + e) - y * p.W;
            src[e] = (y >> 1) * (p.W >> 1) + (xx >> 1);
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
          const bool cok = c < p.C;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = cok & (m0 + e < p.HW) & !(p.dbg & 2);
            xv[i][j][4 * q + e] = fsv_buf_load1(xbuf, ok ? (unsigned)((src[e] * p.C + c) * 4) : FSV_BUF_OOB);
          }
        }
      }
  };
  // per-channel constants of this workgroup's channel tile, once: statistics and the gamma / beta biases of every map (a bias
  // loaded where it is used costs a full memory latency between the last MFMA of a map and its modulation)
  constexpr bool BIAS_REGS = NS == 1;     // (the two-site form has no registers to spare: it loads the biases where it uses them)
  float mu[TN], rs[TN], bgv[BIAS_REGS ? NS : 1][NM][TN], bbv[BIAS_REGS ? NS : 1][NM][TN];
  {
    const float* mean = p.mean + z * p.stat_bstride;
    const float* rstd = p.rstd + z * p.stat_bstride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
      const bool cok = c < p.C;
      mu[j] = cok ? mean[c] : 0.f; rs[j] = cok ? rstd[c] : 0.f;
#pragma unroll
      for (int s = 0; s < (BIAS_REGS ? NS : 0); ++s)
#pragma unroll
        for (int k = 0; k < NM; ++k) {
          const bool on = cok & (k < p.nmaps);
          bgv[s][k][j] = on ? (p.bg[s][k] + z * p.b_bstride[s][k])[c] : 0.f;
          bbv[s][k][j] = on ? (p.bb[s][k] + z * p.b_bstride[s][k])[c] : 0.f;
        }
    }
  }

  // ---- the flat chunk sequence: (pixel tile, map, K chunk) ---------------------------------------------------------------------
  int nch[FSV_SP_MAXMAPS], total = 0;
#pragma unroll
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) { nch[k] = (k < p.nmaps) ? (p.ch[k] + BK - 1) / BK : 0; total += nch[k]; }
  // loader state: (tile ld_bm0, map ld_k, chunk ld_c) of the NEXT chunk to load; past the last tile every row is out of range
  int ld_k = 0, ld_c = 0, ld_bm0 = blockIdx.x * BM;
  auto advance_loader = [&]() {
    ++ld_c;
    const int n_k = ld_k == 0 ? nch[0] : (ld_k == 1 ? nch[1] : nch[2]);
    if (ld_c >= n_k) {
      ld_c = 0; ++ld_k;
      if (ld_k >= p.nmaps) { ld_k = 0; ld_bm0 += tile_step; }
    }
  };
  float4 areg[NPA], breg[NB][NPB];
  float4 hareg[HNPA], hbreg[NB][HNPB];
  auto issue_loads_h = [&]() {
    const int k = ld_k;
    const int Ch = p.ch[k];
    const fsv_buf abuf = fsv_make_buf(reinterpret_cast<const h16*>(p.map[k]) + pix0 * Ch, (long long)p.HW * Ch * 2);
    const int kk = ld_c * BK + hs * 8;
#pragma unroll
    for (int i = 0; i < HNPA; ++i) {
      const int r = hr0 + i * 64;
      const int m = ld_bm0 + r;
      const bool ok = (kk < Ch) & (m < p.HW) & (r < BM);
      hareg[i] = fsv_buf_load4(abuf, ok ? (unsigned)((m * Ch + kk) * 2) : FSV_BUF_OOB);
    }
    const int ldk = (Ch + 31) & ~31;                              // row length Kh of this map's operand
    const long long wbytes = (long long)p.C * ldk * 2;            // one half (gamma or beta) of the operand: C rows of Kh halves
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const h16* w = reinterpret_cast<const h16*>(q ? p.wb[0][k] : p.wg[0][k]) + z * p.w_bstride[0][k];
      const fsv_buf wbuf = fsv_make_buf(w, wbytes);
#pragma unroll
      for (int i = 0; i < HNPB; ++i) {
        const int r = hr0 + i * 64;
        const int c = bn0 + r;
        const bool ok = (ld_bm0 < p.HW) & (c < p.C) & (r < BN);
        hbreg[q][i] = fsv_buf_load4(wbuf, ok ? (unsigned)((c * ldk + kk) * 2) : FSV_BUF_OOB);
      }
    }
    advance_loader();
  };
  auto store_chunk_h = [&](int buf) {
    h16* a_dst = Ah + buf * A_ST;
    h16* b_dst = Bh + buf * (NB * B_ST);
#pragma unroll
    for (int i = 0; i < HNPA; ++i) {
      const int r = hr0 + i * 64;
      if (r < BM) *reinterpret_cast<float4*>(&a_dst[r * BK + (((hs ^ (r >> 2)) & 3) << 3)]) = hareg[i];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int i = 0; i < HNPB; ++i) {
        const int r = hr0 + i * 64;
        if (r < BN) *reinterpret_cast<float4*>(&b_dst[q * B_ST + r * BK + (((hs ^ (r >> 2)) & 3) << 3)]) = hbreg[q][i];
      }
  };
  auto issue_loads = [&]() {
    // uniform: descriptors of the loader's current map
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
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const fsv_buf gbuf = fsv_make_buf(p.wg[s][k] + z * p.w_bstride[s][k], wbytes);
      const fsv_buf bbuf = fsv_make_buf(p.wb[s][k] + z * p.w_bstride[s][k], wbytes);
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int kr = ld_c * BK + br0 + i * RPB;
        const unsigned off = (live & bcol_ok) ? (unsigned)((kr * p.ldw + bcol) * 4) : FSV_BUF_OOB;
        breg[2 * s][i] = fsv_buf_load4(gbuf, off);
        breg[2 * s + 1][i] = fsv_buf_load4(bbuf, off);
      }
    }
    advance_loader();
  };
  auto store_chunk = [&](int buf) {
    float* a_dst = As + buf * A_ST;
    float* b_dst = Bs + buf * (NB * B_ST);
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int r = ar0 + i * RPA;
      // quad (k0 k1 k2 k3) stored as (k0 k2 | k1 k3), rows with bit 4 set as (k1 k3 | k0 k2): a lane reads the two values its
      // MFMA steps need with one ds_read_b64, no select between the MFMAs (conv_igemm.hip, AF)
      const bool hi = (r >> 4) & 1;
      float4 v;
      v.x = hi ? areg[i].y : areg[i].x; v.y = hi ? areg[i].w : areg[i].z;
      v.z = hi ? areg[i].x : areg[i].y; v.w = hi ? areg[i].z : areg[i].w;
      *reinterpret_cast<float4*>(&a_dst[r * BK + ((kq ^ ((r >> 1) & 7)) << 2)]) = v;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int i = 0; i < NPB; ++i)
        *reinterpret_cast<float4*>(&b_dst[q * B_ST + (br0 + i * RPB) * BN + bq * 4]) = breg[q][i];
  };

  // running value of the normalised + modulated activation per site, in MFMA C/D layout
  f32x16 outv[NS][TM][TN];
  // backward twin: g_k and o_k of every map (BWD only; dead code otherwise)
  f32x16 keep_g[BWD ? NM : 1][TM][TN], keep_o[BWD ? NM : 1][TM][TN];
  f32x16 acc[NB][TM][TN];
#pragma unroll
  for (int q = 0; q < NB; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][i][j][r] = 0.f;

  int a_off[TM], a_swz[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (TM * 32) + i * 32 + lrow;
    a_off[i] = r * BK + 2 * (lk ^ ((r >> 4) & 1));        // this lane's half of every quad
    a_swz[i] = (r >> 1) & 7;
  }
  const int b_off = lk * BN + wn * (TN * 32) + lrow;

  // fragments of one k-group (8 k): two quads of A per row tile, four B values per (image, column tile); read one group ahead
  // of their MFMAs and pinned there with scheduling fences (conv_igemm.hip)
  auto read_group = [&](const float* a_src, const float* b_src, int g, float2 (&a4)[2][TM], float (&b)[4][NB][TN]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a4[q][i] = *reinterpret_cast<const float2*>(&a_src[a_off[i] + (((2 * g + q) ^ a_swz[i]) << 2)]);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) b[s4][q][j] = b_src[q * B_ST + b_off + (8 * g + 2 * s4) * BN + j * 32];
  };
  auto mma_group = [&](const float2 (&a4)[2][TM], const float (&b)[4][NB][TN]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float2 v = a4[s4 >> 1][i];
        const float a = (s4 & 1) ? v.y : v.x;
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[s4][q][j], acc[q][i][j], 0, 0, 0);
      }
  };

  // modulation of map k: every site folds its gamma / beta accumulators into its running value and clears them
  auto modulate = [&](int k, bool first) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (BIAS_REGS) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float o = first ? (xv[i][j][r] - mu[j]) * rs[j] : outv[s][i][j][r];
              const float gk = acc[2 * s][i][j][r] + bgv[s][k][j];
              if constexpr (BWD) { keep_g[k][i][j][r] = gk; keep_o[k][i][j][r] = o; }
              outv[s][i][j][r] = o * (1.f + gk) + (acc[2 * s + 1][i][j][r] + bbv[s][k][j]);
              acc[2 * s][i][j][r] = 0.f; acc[2 * s + 1][i][j][r] = 0.f;
            }
        } else {
          const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
          const float bg_l = (c < p.C) ? (p.bg[s][k] + z * p.b_bstride[s][k])[c] : 0.f;
          const float bb_l = (c < p.C) ? (p.bb[s][k] + z * p.b_bstride[s][k])[c] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float o = first ? (xv[i][j][r] - mu[j]) * rs[j] : outv[s][i][j][r];
              const float gk = acc[2 * s][i][j][r] + bg_l;
              outv[s][i][j][r] = o * (1.f + gk) + (acc[2 * s + 1][i][j][r] + bb_l);
              acc[2 * s][i][j][r] = 0.f; acc[2 * s + 1][i][j][r] = 0.f;
            }
        }
      }
    }
  };

  int buf = 0;
  // one chunk: loads of the NEXT chunk of the flat sequence (whatever map and pixel tile it belongs to) at the top, MFMAs of
  // this one, the loaded registers stored into the other LDS buffer behind three quarters of them
  auto chunk = [&]() {
    if constexpr (F16) {
      issue_loads_h();
      const h16* a_src = Ah + buf * A_ST;
      const h16* b_src = Bh + buf * (NB * B_ST);
      h16x8 fa[2][TM], fb[2][NB][TN];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wm * (TM * 32) + i * 32 + lrow;
          fa[st][i] = *reinterpret_cast<const h16x8*>(&a_src[r * BK + ((((2 * st + lk) ^ (r >> 2)) & 3) << 3)]);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int r = wn * (TN * 32) + j * 32 + lrow;
            fb[st][q][j] = *reinterpret_cast<const h16x8*>(&b_src[q * B_ST + r * BK + ((((2 * st + lk) ^ (r >> 2)) & 3) << 3)]);
          }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int q = 0; q < NB; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st][i], fb[st][q][j], acc[q][i][j], 0, 0, 0);
      store_chunk_h(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      return;
    }
    issue_loads();                      // past the end: every lane is out of range -> zeros, never used
    const float* a_src = As + buf * A_ST;
    const float* b_src = Bs + buf * (NB * B_ST);
    if constexpr (NS == 1 && !BWD) {
      // fragments one k-group ahead of their MFMAs (two register sets)
      float2 fa[2][2][TM];
      float fb[2][4][NB][TN];
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
    } else {
      // the two-site / backward forms carry 64 - 144 more live registers (four accumulators, or g_k / o_k of three maps):
      // one fragment set, two workgroups per CU cover each other's LDS latency
      float2 fa[2][TM];
      float fb[4][NB][TN];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        read_group(a_src, b_src, g, fa, fb);
        mma_group(fa, fb);
        if (g == 2) { FSV_SCHED_FENCE(); store_chunk(buf ^ 1); FSV_SCHED_FENCE(); }
      }
    }
    __syncthreads();
    buf ^= 1;
  };

  // ---- the workgroup's pixel tiles ---------------------------------------------------------------------------------------------
  load_x(blockIdx.x * BM);
  if (total > 0) {
    if constexpr (F16) { issue_loads_h(); store_chunk_h(0); } else { issue_loads(); store_chunk(0); }
    __syncthreads();
  }
  const bool dh_half = (p.h_half & 1) != 0;
  const fsv_buf dhbuf = !BWD ? fsv_make_buf(nullptr, 0)
                             : (dh_half ? fsv_make_buf(reinterpret_cast<const _Float16*>(p.dh) + pix0 * p.C, (long long)p.HW * p.C * 2)
                                        : fsv_make_buf(p.dh + pix0 * p.C, (long long)p.HW * p.C * 4));
#pragma unroll 1
  for (int bm0 = blockIdx.x * BM; bm0 < p.HW; bm0 += tile_step) {
    // backward twin: this tile's dh values, requested before its chunks when there is room (the stores of the epilogue may alias
    // p.dh as far as the compiler can tell: a load inside the store loop would wait for its full latency once per element)
    float dv[BWD ? TM : 1][BWD ? TN : 1][16];
    auto load_dv = [&]() {
      if constexpr (BWD) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if (dh_half) {          // uniform.  One dword = channels (c & ~1, + 1) of one pixel; the lane pair shares its two loads
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int m = bm0 + wm * (TM * 32) + i * 32 + row + (lane & 1);        // even lane: pixel r, odd lane: pixel r + 1
                const bool ok = (c < p.C) & (m < p.HW) & !(p.dbg & 4);
                const float w = fsv_buf_load1(dhbuf, ok ? (unsigned)(m * p.C + (c & ~1)) * 2u : FSV_BUF_OOB);
                const float nw = __shfl_xor(w, 1);
                const h16x2 mine = __builtin_bit_cast(h16x2, w), theirs = __builtin_bit_cast(h16x2, nw);
                dv[i][j][r] = (lane & 1) ? (float)theirs.y : (float)mine.x;
                dv[i][j][r + 1] = (lane & 1) ? (float)mine.y : (float)theirs.x;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
                const int m = bm0 + wm * (TM * 32) + i * 32 + row;
                const bool ok = (c < p.C) & (m < p.HW) & !(p.dbg & 4);
                dv[i][j][r] = fsv_buf_load1(dhbuf, ok ? (unsigned)(m * p.C + c) * 4u : FSV_BUF_OOB);
              }
            }
          }
        }
      }
    };
    if constexpr (DVEARLY) load_dv();
    if (total > 0) {
      // per map: its chunks, then its modulation (registers only; the next chunk - of the next map or of the next pixel tile -
      // is already in LDS); x of the next tile is requested as soon as the first modulation has consumed this tile's
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        if (k < p.nmaps) {
#pragma unroll 1
          for (int c = 0; c < nch[k]; ++c) chunk();
          modulate(k, k == 0);
          if constexpr (XEARLY) { if (k == 0) load_x(bm0 + tile_step); }
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) outv[s][i][j][r] = (xv[i][j][r] - mu[j]) * rs[j];
      if constexpr (XEARLY) load_x(bm0 + tile_step);
    }
    if constexpr (!DVEARLY) load_dv();

    // epilogues: per-sample base pointers are uniform (scalar registers), the per-lane part is a 32-bit element offset (the host
    // checked that one sample of every tensor stays below 2^31 bytes)
    if constexpr (BWD) {
      float* dx_z = p.dxhat + pix0 * p.C;
      float* dg_z[FSV_SP_MAXMAPS];
#pragma unroll
      for (int k = 0; k < FSV_SP_MAXMAPS; ++k) dg_z[k] = (k < p.nmaps) ? p.dgb[k] + pix0 * 2 * p.C : nullptr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
        const bool cok = c < p.C;
        float sg[FSV_SP_MAXMAPS], sb[FSV_SP_MAXMAPS];
#pragma unroll
        for (int k = 0; k < FSV_SP_MAXMAPS; ++k) { sg[k] = 0.f; sb[k] = 0.f; }
        const bool dg_half = (p.h_half & 2) != 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            // two consecutive pixels at a time (the half stores pair them); every lane runs the arithmetic, the stores and the
            // bias sums are predicated
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int m = bm0 + wm * (TM * 32) + i * 32 + row;
            const bool ok0 = cok & (m < p.HW), ok1 = cok & (m + 1 < p.HW);
            float d0 = dv[i][j][r], d1 = dv[i][j][r + 1];
            if ((p.dbg & 1) && d0 != 1.2345e30f) { sb[0] += d0 * keep_o[0][i][j][r] + keep_g[0][i][j][r] + d1; continue; }
            if (p.act[0] == FSV_ACT_LRELU) {
              d0 = (outv[0][i][j][r] > 0.f) ? d0 : 0.2f * d0;
              d1 = (outv[0][i][j][r + 1] > 0.f) ? d1 : 0.2f * d1;
            }
#pragma unroll
            for (int k = NM - 1; k >= 0; --k) {
              if (k < p.nmaps) {
                const int e2 = m * 2 * p.C + c;
                const float g0 = d0 * keep_o[k][i][j][r], g1 = d1 * keep_o[k][i][j][r + 1];
                if (dg_half) {
                  h16* dgh = reinterpret_cast<h16*>(p.dgb[k]) + pix0 * 2 * p.C;
                  store_pair_h(dgh + p.C, e2, e2 + 2 * p.C, d0, d1, ok0, ok1);
                  store_pair_h(dgh, e2, e2 + 2 * p.C, g0, g1, ok0, ok1);
                } else {
                  float* dg = dg_z[k];
                  if (ok0) { dg[e2 + p.C] = d0; dg[e2] = g0; }
                  if (ok1) { dg[e2 + 3 * p.C] = d1; dg[e2 + 2 * p.C] = g1; }
                }
                sb[k] += (ok0 ? d0 : 0.f) + (ok1 ? d1 : 0.f); sg[k] += (ok0 ? g0 : 0.f) + (ok1 ? g1 : 0.f);
                d0 = d0 * (1.f + keep_g[k][i][j][r]); d1 = d1 * (1.f + keep_g[k][i][j][r + 1]);
              }
            }
            if (ok0) dx_z[m * p.C + c] = d0;
            if (ok1) dx_z[(m + 1) * p.C + c] = d1;
          }
        if (p.dbsum) {               // uniform: bias gradients from here (the two half-waves hold the same 32 channels)
          // 2 * HW / 64 adds per address and sample would serialise in the L2 (measured + 120 us at 512 K pixels): the pixel tiles
          // spread over db_slots copies of the buffer, which the caller sums
          double* slot = p.dbsum + (long long)((bm0 / BM) & (p.db_slots - 1)) * p.db_slot_stride;
#pragma unroll
          for (int k = 0; k < NM; ++k) {
            if (k < p.nmaps) {
              const float g2 = sg[k] + __shfl_xor(sg[k], 32), b2 = sb[k] + __shfl_xor(sb[k], 32);
              if (lk == 0 && cok) {
                double* dst = slot + z * p.db_zstride[k] + (long long)k * 2 * p.C;
                atomicAdd(dst + c, (double)g2);
                atomicAdd(dst + p.C + c, (double)b2);
              }
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float* h_z = p.h[s] + pix0 * p.C;
        h16* h_zh = reinterpret_cast<h16*>(p.h[s]) + pix0 * p.C;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
          const bool cok = c < p.C;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
              const int m = bm0 + wm * (TM * 32) + i * 32 + row;
              const bool ok0 = cok & (m < p.HW), ok1 = cok & (m + 1 < p.HW);
              const float v0 = fsv_act(outv[s][i][j][r], p.act[s]), v1 = fsv_act(outv[s][i][j][r + 1], p.act[s]);
              if ((p.dbg & 1) && v0 != 1.2345e30f) continue;
              if (NS == 1 && p.h_half) {          // (the two-site form has no half output)
                store_pair_h(h_zh, m * p.C + c, (m + 1) * p.C + c, v0, v1, ok0, ok1);
              } else {
                if (ok0) h_z[m * p.C + c] = v0;
                if (ok1) h_z[(m + 1) * p.C + c] = v1;
              }
            }
        }
      }
    }
    if constexpr (!XEARLY) load_x(bm0 + tile_step);
  }
}

// ---- backward, elementwise part -------------------------------------------------------------------------------
// gb_k: materialised [P][2C] (gamma | beta) per map (recomputed by the caller with the gather-GEMM kernel).
// Computes, per element, the chain  o_0 = xhat, o_k = o_{k-1} (1 + g_k) + b_k,  h = act(o_n)  backwards:
//   d = dh * act'(h);  for k = n..1:  dbeta_k = d; dgamma_k = d * o_{k-1}; d = d * (1 + g_k);   dxhat = d
struct SpadeBwdP {
  const float* x; const float* mean; const float* rstd;
  const float* dh; const float* h;
  const float* gb[FSV_SP_MAXMAPS];
  float* dgb[FSV_SP_MAXMAPS];
  float* dxhat;
  int nmaps, C, act;
  long long total;        // N*HW*C
  long long HWC;          // per-sample elements (for per-sample statistics)
  long long stat_bstride;
  int HW, W, up;          // up = 1: x is the half-resolution tensor (see SpadeP); dxhat is still written per full-resolution pixel
};

__global__ __launch_bounds__(256) void fsv_spade_bwd_elem_kernel(SpadeBwdP p) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < p.total; i += stride) {
    const int c = (int)(i % p.C);
    const long long pix = i / p.C;
    const long long n = i / p.HWC;
    const float mu = p.mean[n * p.stat_bstride + c], rs = p.rstd[n * p.stat_bstride + c];
    float o[FSV_SP_MAXMAPS + 1];
    float g[FSV_SP_MAXMAPS];
    long long xi = i;
    if (p.up) {
      const int rem = (int)(pix - n * p.HW);
      const int y = rem / p.W, xx = rem - y * p.W;
      xi = ((n * (p.HW >> 2)) + (long long)(y >> 1) * (p.W >> 1) + (xx >> 1)) * p.C + c;
    }
    o[0] = (p.x[xi] - mu) * rs;
#pragma unroll
    for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
      if (k < p.nmaps) {
        const float* gbp = p.gb[k] + pix * 2 * p.C;
        g[k] = gbp[c];
        o[k + 1] = o[k] * (1.f + g[k]) + gbp[p.C + c];
      }
    }
    float d = p.dh[i];
    if (p.act == FSV_ACT_LRELU) d = (p.h[i] > 0.f) ? d : 0.2f * d;
#pragma unroll
    for (int k = FSV_SP_MAXMAPS - 1; k >= 0; --k) {
      if (k < p.nmaps) {
        float* dg = p.dgb[k] + pix * 2 * p.C;
        dg[p.C + c] = d;
        dg[c] = d * o[k];
        d = d * (1.f + g[k]);
      }
    }
    p.dxhat[i] = d;
  }
}

// ---- one-launch operand preparation for a (gamma, beta) weight pair ------------------------------------------------------
// wg / wb: [B][C][Ch] (1x1 OIHW, sample strides swg / swb; 0 = shared), bg / bb: [B][C].  Outputs (per sample):
//   wcat_t [Kt = ceil32(Ch)][2C]      forward operand of the combined convolution  map -> [gamma | beta]
//   wcat_d [ceil32(2C)][Ld = ceil32(Ch)]  data-gradient operand (optional)
//   bcat   [2C]
// The modulation kernel reads gamma weights from columns [0, C) and beta weights from [C, 2C) of wcat_t, the backward
// pass re-uses all three as they are (no concatenation / re-arrangement launches).  Needs C % 16 == 0.
__global__ __launch_bounds__(256) void fsv_spade_prep_kernel(const float* wg, const float* wb, const float* bg, const float* bb,
                                                             long long swg, long long swb, long long sbg, long long sbb,
                                                             float* wcat_t, float* wcat_d, float* bcat, int C, int Ch) {
  const int z = blockIdx.y;
  const int Kt = (Ch + 31) / 32 * 32, Lt = 2 * C;
  const int Kd = (2 * C + 31) / 32 * 32, Ld = (Ch + 31) / 32 * 32;
  const long long nt = (long long)Kt * Lt, nd = wcat_d ? (long long)Kd * Ld : 0, nb = 2 * C;
  const float* g = wg + z * swg;
  const float* b = wb + z * swb;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nt + nd + nb; i += (long long)gridDim.x * 256) {
    if (i < nt) {
      const int k = (int)(i / Lt), j = (int)(i - (long long)k * Lt);
      float v = 0.f;
      if (k < Ch) v = j < C ? g[(long long)j * Ch + k] : b[(long long)(j - C) * Ch + k];
      wcat_t[z * nt + i] = v;
    } else if (i < nt + nd) {
      const long long e = i - nt;
      const int r = (int)(e / Ld), k = (int)(e - (long long)r * Ld);
      float v = 0.f;
      if (k < Ch && r < 2 * C) v = r < C ? g[(long long)r * Ch + k] : b[(long long)(r - C) * Ch + k];
      wcat_d[z * nd + e] = v;
    } else {
      const int j = (int)(i - nt - nd);
      bcat[z * nb + j] = j < C ? bg[z * sbg + j] : bb[z * sbb + (j - C)];
    }
  }
}

extern "C" {

int fsv_spade_prep(const float* wg, const float* wb, const float* bg, const float* bb, long long swg, long long swb,
                   long long sbg, long long sbb, float* wcat_t, float* wcat_d, float* bcat, int B, int C, int Ch,
                   hipStream_t stream) {
  if (!wg || !wb || !bg || !bb || !wcat_t || !bcat || B < 1 || C < 16 || (C & 15) || Ch < 1) return FSV_ERR_BAD_ARG;
  const long long Kt = (Ch + 31) / 32 * 32, Kd = (2 * C + 31) / 32 * 32;
  long long total = Kt * 2 * C + (wcat_d ? Kd * Kt : 0) + 2 * C;
  long long g = (total + 255) / 256;
  if (g > 1024) g = 1024;
  FSV_LAUNCH(fsv_spade_prep_kernel, dim3((unsigned)g, B), dim3(256), stream, wg, wb, bg, bb, swg, swb, sbg, sbb, wcat_t,
             wcat_d, bcat, C, Ch);
  return fsv_check_launch();
}

static inline int fsv_sp_fill_site(SpadeP& p, int s, int nmaps, const float* const* wg, const float* const* wb,
                                   const float* const* bg, const float* const* bb, const long long* w_bstride,
                                   const long long* b_bstride) {
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
    const bool on = k < nmaps;
    p.wg[s][k] = on ? wg[k] : nullptr; p.wb[s][k] = on ? wb[k] : nullptr;
    p.bg[s][k] = on ? bg[k] : nullptr; p.bb[s][k] = on ? bb[k] : nullptr;
    p.w_bstride[s][k] = on ? w_bstride[k] : 0; p.b_bstride[s][k] = on ? b_bstride[k] : 0;
    if (on && (!wg[k] || !wb[k] || !bg[k] || !bb[k])) return FSV_ERR_UNSUPPORTED;
  }
  return FSV_OK;
}

static inline int fsv_sp_fill_common(SpadeP& p, const float* x, const float* mean, const float* rstd, int nmaps,
                                     const float* const* maps, const int* ch, int N, int HW, int C, int ldw,
                                     long long stat_bstride, int W, int up) {
  if (!x || !mean || !rstd || nmaps < 0 || nmaps > FSV_SP_MAXMAPS || C < 1 || (ldw & 3)) return FSV_ERR_BAD_ARG;
  if (up && (W < 2 || (W & 1) || HW % W != 0 || ((HW / W) & 1))) return FSV_ERR_BAD_ARG;
  // 32-bit byte offsets inside one sample of a map / inside x (buffer descriptors)
  if ((long long)N * HW * C * 4 > FSV_BUF_MAX_BYTES || (long long)HW * 2 * C * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  p.x = x; p.mean = mean; p.rstd = rstd; p.nmaps = nmaps;
  for (int s = 0; s < FSV_SP_SITES; ++s) {
    p.h[s] = nullptr; p.act[s] = FSV_ACT_NONE;
    for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
      p.wg[s][k] = p.wb[s][k] = p.bg[s][k] = p.bb[s][k] = nullptr;
      p.w_bstride[s][k] = p.b_bstride[s][k] = 0;
    }
  }
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
    const bool on = k < nmaps;
    p.map[k] = on ? maps[k] : nullptr; p.ch[k] = on ? ch[k] : 0; p.dgb[k] = nullptr;
    if (on && ((ch[k] & 3) || !maps[k])) return FSV_ERR_UNSUPPORTED;
    if (on && (long long)HW * ch[k] * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  }
  p.N = N; p.HW = HW; p.C = C; p.ldw = ldw; p.stat_bstride = stat_bstride;
  p.W = up ? W : 1; p.up = up ? 1 : 0;
  p.dh = nullptr; p.dxhat = nullptr;
  p.h_half = 0;
  p.dbsum = nullptr; p.db_slots = 1; p.db_slot_stride = 0;
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) p.db_zstride[k] = 0;
  {
    const char* e = getenv("FSV_SPADE_DBG");
    p.dbg = e ? atoi(e) : 0;
  }
  return FSV_OK;
}

// pixel-tile workgroups of a launch: as many as stay resident at once (256 CUs x wgs_per_cu), each walking its share of the tiles
static inline unsigned fsv_sp_grid_x(int ntiles, int gy, int n, int wgs_per_cu) {
  const char* e = getenv("FSV_SPADE_WGS_PER_CU");           // in-box experiments; 0 = one workgroup per tile
  if (e) wgs_per_cu = atoi(e);
  if (wgs_per_cu <= 0) return (unsigned)ntiles;
  long long cap = (256ll * wgs_per_cu) / ((long long)gy * n);
  const char* m = getenv("FSV_SPADE_MAX_GX");               // tests: a multi-tile walk on a small map
  if (m && atoi(m) > 0) cap = atoi(m);
  if (cap < 1) cap = 1;
  return (unsigned)(ntiles < cap ? ntiles : cap);
}

// maps/wg/wb/bg/bb: arrays of nmaps device pointers; ch / w_bstride / b_bstride: per-map ints / strides.
static int fsv_spade_mod_fwd_impl(const float* x, const float* mean, const float* rstd, float* h,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                      int W, int up, int flags, hipStream_t stream) {
  if (!h) return FSV_ERR_BAD_ARG;
  SpadeP p;
  int rc = fsv_sp_fill_common(p, x, mean, rstd, nmaps, maps, ch, N, HW, C, ldw, stat_bstride, W, up);
  if (rc) return rc;
  rc = fsv_sp_fill_site(p, 0, nmaps, wg, wb, bg, bb, w_bstride, b_bstride);
  if (rc) return rc;
  p.h[0] = h; p.act[0] = act; p.h_half = flags & 1;
  const bool f16 = (flags & 4) != 0;
  if (f16) {            // half maps / half N-major weights: 16-byte vectors of 8 k
    for (int k = 0; k < nmaps; ++k)
      if (ch[k] & 7) return FSV_ERR_UNSUPPORTED;
  }
  if (C <= 32) {
    const int gy = fsv_cdiv(C, 32);
    dim3 g(fsv_sp_grid_x(fsv_cdiv(HW, 128), gy, N, 3), gy, N);
    if (f16) FSV_LAUNCH((fsv_spade_mod_kernel<128, 32, 4, 1, 1, false, true>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_mod_kernel<128, 32, 4, 1, 1, false>), g, dim3(256), stream, p);
  } else {
    const int gy = fsv_cdiv(C, 64);
    dim3 g(fsv_sp_grid_x(fsv_cdiv(HW, 64), gy, N, f16 ? 4 : 3), gy, N);
    if (f16) FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, false, true>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, false>), g, dim3(256), stream, p);
  }
  return fsv_check_launch();
}

int fsv_spade_mod_fwd(const float* x, const float* mean, const float* rstd, float* h,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                      int W, int up, hipStream_t stream) {
  return fsv_spade_mod_fwd_impl(x, mean, rstd, h, nmaps, maps, wg, wb, bg, bb, ch, w_bstride, b_bstride, N, HW, C, ldw, stat_bstride,
                                act, W, up, 0, stream);
}

// the `--amp` forms.  flags bit 0: h written as IEEE half ([N][HW][C] halves) - the modulated tensor is only ever read by the
// half-precision convolutions (csrc/conv_h.hip): one rounding at the store instead of a fp32 tensor + a conversion pass.  bit 2:
// the gamma / beta GEMMs on the f16 matrix instructions - `maps` are IEEE half ([N][HW][Ch], Ch % 8 == 0), wg / wb point at the
// gamma rows / beta rows of the N-major half operand of fsv_spade_prep_h (row length ceil32(Ch) halves; ldw unused, w_bstride in halves).
int fsv_spade_mod_fwd_h(const float* x, const float* mean, const float* rstd, void* h,
                        int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                        int W, int up, int flags, hipStream_t stream) {
  return fsv_spade_mod_fwd_impl(x, mean, rstd, reinterpret_cast<float*>(h), nmaps, reinterpret_cast<const float* const*>(maps),
                                reinterpret_cast<const float* const*>(wg), reinterpret_cast<const float* const*>(wb), bg, bb, ch,
                                w_bstride, b_bstride, N, HW, C, ldw, stat_bstride, act, W, up, flags, stream);
}

// Two norm sites of one SPADEResnetBlock in ONE launch (architecture.py:95-96,103: bn_0 and bn_s normalise the same x with the
// same statistics and read the same maps): h0 = act0(SPADE_0(x)), h1 = act1(SPADE_s(x)).  Arrays are [2 * nmaps]: site 0's
// entries first.
int fsv_spade_mod_fwd2(const float* x, const float* mean, const float* rstd, float* h0, float* h1,
                       int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                       const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                       const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act0, int act1,
                       int W, int up, hipStream_t stream) {
  if (!h0 || !h1 || nmaps < 1) return FSV_ERR_BAD_ARG;
  SpadeP p;
  int rc = fsv_sp_fill_common(p, x, mean, rstd, nmaps, maps, ch, N, HW, C, ldw, stat_bstride, W, up);
  if (rc) return rc;
  for (int s = 0; s < 2; ++s) {
    rc = fsv_sp_fill_site(p, s, nmaps, wg + s * nmaps, wb + s * nmaps, bg + s * nmaps, bb + s * nmaps, w_bstride + s * nmaps,
                          b_bstride + s * nmaps);
    if (rc) return rc;
  }
  p.h[0] = h0; p.h[1] = h1; p.act[0] = act0; p.act[1] = act1;
  if (C <= 32) {
    dim3 g(fsv_cdiv(HW, 128), fsv_cdiv(C, 32), N);
    FSV_LAUNCH((fsv_spade_mod_kernel<128, 32, 4, 1, 2, false>), g, dim3(256), stream, p);
  } else {
    dim3 g(fsv_cdiv(HW, 64), fsv_cdiv(C, 64), N);
    FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 2, false>), g, dim3(256), stream, p);
  }
  return fsv_check_launch();
}

// Backward twin of fsv_spade_mod_fwd (see the kernel comment): same operands, dh in, dgb[k] ([P][2C] per map) and dxhat out.
static int fsv_spade_mod_bwd_impl(const float* x, const float* mean, const float* rstd, const float* dh,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, float* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                      long long stat_bstride, int act, int W, int up, int flags, double* dbsum, const long long* db_zstride,
                      int db_slots, long long db_slot_stride, hipStream_t stream) {
  if (!dh || !dgb || !dxhat) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_LRELU && act != FSV_ACT_NONE) return FSV_ERR_UNSUPPORTED;
  SpadeP p;
  int rc = fsv_sp_fill_common(p, x, mean, rstd, nmaps, maps, ch, N, HW, C, ldw, stat_bstride, W, up);
  if (rc) return rc;
  rc = fsv_sp_fill_site(p, 0, nmaps, wg, wb, bg, bb, w_bstride, b_bstride);
  if (rc) return rc;
  p.dh = dh; p.dxhat = dxhat; p.act[0] = act; p.h_half = flags & 3;
  const bool f16 = (flags & 4) != 0;
  if (f16) {
    for (int k = 0; k < nmaps; ++k)
      if (ch[k] & 7) return FSV_ERR_UNSUPPORTED;
  }
  for (int k = 0; k < nmaps; ++k) {
    if (!dgb[k]) return FSV_ERR_UNSUPPORTED;
    p.dgb[k] = dgb[k];
  }
  if (dbsum) {
    if (!db_zstride) return FSV_ERR_BAD_ARG;
    long long span = 0;
    for (int k = 0; k < nmaps; ++k) {
      p.db_zstride[k] = db_zstride[k];
      const long long e = (long long)(N - 1) * db_zstride[k] + (long long)(k + 1) * 2 * C;
      if (e > span) span = e;
    }
    if (db_slots < 1 || (db_slots & (db_slots - 1)) || (db_slots > 1 && db_slot_stride < span)) return FSV_ERR_BAD_ARG;
    p.dbsum = dbsum; p.db_slots = db_slots; p.db_slot_stride = db_slots > 1 ? db_slot_stride : 0;
    (void)hipMemsetAsync(dbsum, 0, (size_t)((long long)(db_slots - 1) * p.db_slot_stride + span) * sizeof(double), stream);
  }
  // 64 x 64 tiles: every wave keeps g_k and o_k of up to three maps for a 32 x 32 sub-tile (96 + 48 accumulator registers),
  // which leaves room for several workgroups per CU
  const int gy = fsv_cdiv(C, 64);
  dim3 g(fsv_sp_grid_x(fsv_cdiv(HW, 64), gy, N, nmaps <= 1 ? 3 : 2), gy, N);
  if (nmaps <= 1) {
    if (f16) FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, true, true, 1>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, true, false, 1>), g, dim3(256), stream, p);
  } else {
    if (f16) FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, true, true>), g, dim3(256), stream, p);
    else FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, 1, true>), g, dim3(256), stream, p);
  }
  return fsv_check_launch();
}

int fsv_spade_mod_bwd(const float* x, const float* mean, const float* rstd, const float* dh,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, float* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                      long long stat_bstride, int act, int W, int up, hipStream_t stream) {
  return fsv_spade_mod_bwd_impl(x, mean, rstd, dh, nmaps, maps, wg, wb, bg, bb, ch, w_bstride, b_bstride, dgb, dxhat, N, HW, C, ldw,
                                stat_bstride, act, W, up, 0, nullptr, nullptr, 1, 0, stream);
}

// General form of the backward twin.  flags bit 0: dh is IEEE half (the gradient of a half h); bit 1: every d(gamma|beta) tensor is
// written as half ([P][2C] halves: its consumers are the half-precision data / weight gradient GEMMs); bit 2: f16 GEMMs (maps / wg /
// wb as in fsv_spade_mod_fwd_h).  dbsum (optional, any flags): the bias gradients from this launch - per-channel sums of
// d(gamma|beta) added into dbsum + z * db_zstride[k] + k * 2C as doubles (zeroed here; db_zstride[k] = 0 sums map k over the batch);
// with db_slots > 1 (a power of two) pixel tile t adds into the copy at + (t % db_slots) * db_slot_stride and the caller sums the
// copies - thousands of adds per address serialise in the L2 otherwise.
int fsv_spade_mod_bwd_h(const float* x, const float* mean, const float* rstd, const void* dh,
                        int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, void* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                        long long stat_bstride, int act, int W, int up, int flags, double* dbsum, const long long* db_zstride,
                        int db_slots, long long db_slot_stride, hipStream_t stream) {
  return fsv_spade_mod_bwd_impl(x, mean, rstd, reinterpret_cast<const float*>(dh), nmaps, reinterpret_cast<const float* const*>(maps),
                                reinterpret_cast<const float* const*>(wg), reinterpret_cast<const float* const*>(wb), bg, bb, ch,
                                w_bstride, b_bstride, reinterpret_cast<float* const*>(dgb), dxhat, N, HW, C, ldw, stat_bstride, act, W,
                                up, flags, dbsum, db_zstride, db_slots, db_slot_stride, stream);
}

// half N-major operand of the f16 forms: wcat_h [B][2C][Kh] (Kh = ceil32(Ch) halves per row, zero padded): row j < C = gamma weights
// of channel j, row C + j = beta weights - from wg / wb [B][C][Ch] (sample strides swg / swb; 0 = shared)
__global__ __launch_bounds__(256) void fsv_spade_prep_h_kernel(const float* wg, const float* wb, long long swg, long long swb,
                                                               _Float16* wcat_h, int C, int Ch, int Kh) {
  const int z = blockIdx.y;
  const long long total = (long long)2 * C * Kh;
  const float* g = wg + z * swg;
  const float* b = wb + z * swb;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int j = (int)(i / Kh), k = (int)(i - (long long)j * Kh);
    float v = 0.f;
    if (k < Ch) v = j < C ? g[(long long)j * Ch + k] : b[(long long)(j - C) * Ch + k];
    wcat_h[z * total + i] = (_Float16)v;
  }
}

int fsv_spade_prep_h(const float* wg, const float* wb, long long swg, long long swb, void* wcat_h, int B, int C, int Ch,
                     hipStream_t stream) {
  if (!wg || !wb || !wcat_h || B < 1 || C < 16 || (C & 15) || Ch < 1) return FSV_ERR_BAD_ARG;
  const int Kh = (Ch + 31) / 32 * 32;
  long long g = ((long long)2 * C * Kh + 255) / 256;
  if (g > 1024) g = 1024;
  FSV_LAUNCH(fsv_spade_prep_h_kernel, dim3((unsigned)g, B), dim3(256), stream, wg, wb, swg, swb, reinterpret_cast<_Float16*>(wcat_h), C, Ch, Kh);
  return fsv_check_launch();
}

int fsv_spade_bwd_elem(const float* x, const float* mean, const float* rstd, const float* dh, const float* h,
                       int nmaps, const float* const* gb, float* const* dgb, float* dxhat,
                       int N, int HW, int C, long long stat_bstride, int act, int W, int up, hipStream_t stream) {
  if (!x || !mean || !rstd || !dh || !dxhat || nmaps < 0 || nmaps > FSV_SP_MAXMAPS) return FSV_ERR_BAD_ARG;
  if (up && (W < 2 || (W & 1) || HW % W != 0 || ((HW / W) & 1))) return FSV_ERR_BAD_ARG;
  if (act == FSV_ACT_LRELU && !h) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_LRELU && act != FSV_ACT_NONE) return FSV_ERR_UNSUPPORTED;
  SpadeBwdP p;
  p.x = x; p.mean = mean; p.rstd = rstd; p.dh = dh; p.h = h; p.dxhat = dxhat;
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) { p.gb[k] = k < nmaps ? gb[k] : nullptr; p.dgb[k] = k < nmaps ? dgb[k] : nullptr; }
  p.nmaps = nmaps; p.C = C; p.act = act;
  p.total = (long long)N * HW * C; p.HWC = (long long)HW * C; p.stat_bstride = stat_bstride;
  p.HW = HW; p.W = up ? W : 1; p.up = up ? 1 : 0;
  long long g = (p.total + 256 * 4 - 1) / (256 * 4);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  FSV_LAUNCH(fsv_spade_bwd_elem_kernel, dim3((unsigned)g), dim3(256), stream, p);
  return fsv_check_launch();
}

}  // extern "C"
