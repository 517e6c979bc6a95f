// SPADE modulation fused with the 3x3 convolution that consumes it: ONE kernel for
//     dx = conv_0(actvn(bn_0(x, maps)))      and      out = x_s + conv_1(actvn(bn_1(dx, maps)))
// (reference models/networks/architecture.py:92-99: SPADEResnetBlock.forward; bn_* = SPADE.forward, normalization.py:37-52;
// actvn = leaky_relu(0.2), architecture.py:111-112; conv_*: 3x3, padding 1, spectral norm).  The north-star kernel "SPADE
// denorm + modulate + conv in one launch" for the 3x3 half of the block (the 1x1 half is spade_conv.hip).
//
// The modulated tensor h = actvn(bn(x)) never reaches HBM in a forward pass that keeps no graph; a training forward asks for it as a
// side output (the convolution's weight gradient reads it).
//
// Form.  gamma / beta are 1x1 convolutions of the label maps, so h at a pixel needs x and the maps at THAT pixel only; the 3x3
// convolution needs h in a one-pixel halo around its output tile.  A workgroup owns an 8 x 16 output tile:
//   phase 1  h of the 10 x 18 haloed tile (180 pixels, six 32-pixel blocks) is computed ONCE - 1.41x the gamma / beta GEMM work of
//            the tile's own pixels, against 9x for a modulation in the convolution's gather - and stored in LDS ([pixel][C + 4]).
//            The GEMM is issued with its operands swapped exactly as in spade_conv.hip (D[channel][pixel]: a lane owns one pixel and
//            16 channels in register quads of four consecutive channels), every operand comes straight from global memory / L1 in the
//            fragment layout (a lane's 16-byte map vector = four k steps; weight rows are 128-byte coalesced loads) - no staging, no
//            barrier: the four waves run independently.  Pixels outside the image are stored as zeros: the convolution's padding.
//   phase 2  the convolution reads its A fragments from the LDS patch at the tap's offset (one ds_read_b128 = four k steps,
//            conflict-free at a row length of C + 4) and streams its K-major weight in 64-row (Cout 32; FSV_S3_RW=32: 32-row) /
//            32-row (Cout 64) chunks through a double-buffered LDS tile (one barrier per chunk).  Each wave finishes a 2 x 16 pixel block for
//            every output channel.
// 74.8 KB (Cout 32) / 72.7 KB (Cout 64) of LDS: two workgroups per CU.
#include <type_traits>
#include "conv_igemm.h"

#define FSV_S3_MAXMAPS 3

// A base address that is the same in every lane (kernel arguments and blockIdx only) but that the compiler evaluated on the vector
// unit (a 64-bit multiply by blockIdx.z): pinned into scalar registers, so that the buffer descriptor built from it is scalar and the
// loads through it need no per-load waterfall loop (first build of this kernel: 139 of them, guide T20).
#ifdef FSV_EMU
template <typename T> static inline T* fsv_s3_uniform(T* q) { return q; }
#else
template <typename T> __device__ __forceinline__ T* fsv_s3_uniform(T* q) {
  const unsigned long long a = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
#endif

struct SpadeConv3P {
  const float* x;         // [N][H W or H W / 4][C]
  const float* mean;      // [C] (+ z * stat_bstride)
  const float* rstd;
  float* hs;              // optional: the modulated + activated tensor [N][H W][C]
  float* out;             // [N][H W][Cout]
  const float* map[FSV_S3_MAXMAPS];   // [N][H W][Ch_k]
  const float* wg[FSV_S3_MAXMAPS];    // K-major [ceil32(Ch_k)][ldw] (+ z * w_bstride_k)
  const float* wb[FSV_S3_MAXMAPS];
  const float* bg[FSV_S3_MAXMAPS];    // [C] (+ z * b_bstride_k)
  const float* bb[FSV_S3_MAXMAPS];
  int ch[FSV_S3_MAXMAPS];
  long long w_bstride[FSV_S3_MAXMAPS];
  long long b_bstride[FSV_S3_MAXMAPS];
  int nmaps, N, H, W, C, ldw;
  long long stat_bstride;
  int up, act;
  const float* wc;        // the convolution's forward operand, K-major [(tap, ci)][ldwc] (fsv_prep_weight mode 0)
  const float* bias;      // [Cout] or null
  const float* res;       // [N][H W][Cout] or null: added behind the bias
  const float* wscale;    // optional device scalar on the accumulator (spectral-norm 1 / sigma)
  int Cout, ldwc;
  int tiles_x, ntiles;
  double* stats;          // optional: zeroed partials [stats_slots][Cout][2] = (sum v, sum v^2) of the stored output over all samples -
  int stats_slots;        // the BatchNorm statistics of the normalisation that follows (layout of ConvP::stats, one group)
};

// TN2 output column blocks of 32 (Cout = 32 TN2); C = 64
template <int TN2, int RW>
__global__ __launch_bounds__(256, 2) void fsv_spade_conv3_kernel(SpadeConv3P p) {
  constexpr int C = 64, TH = 8, TW = 16, HWD = TW + 2, HP = (TH + 2) * HWD, PS = C + 4;
  constexpr int NU = 3;                                  // (pixel block, channel block) units per wave: 6 x 2 over 4 waves
  constexpr int BNC = 32 * TN2 + 8;                      // row length of a weight chunk in LDS (+ 8: the k halves on different banks)
  constexpr int NKIND = 2 + 2 * FSV_S3_MAXMAPS;
  // RW: weight rows per chunk of phase 2 (one barrier per chunk): 32, or 64 for Cout 32
  constexpr int NCHUNK = 9 * C / RW;
  constexpr int WQ = RW * 8 * TN2 / 256;                 // 16-byte vectors of a chunk per work-item
  __shared__ __attribute__((aligned(16))) float patch[192 * PS];
  __shared__ __attribute__((aligned(16))) float wch[2 * RW * BNC];
  __shared__ __attribute__((aligned(16))) float cst[NKIND * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int z = blockIdx.z;
  // workgroups go to the eight XCDs round-robin: XCD i takes the i-th CONTIGUOUS eighth of the tiles, so that the halo rows two
  // neighbouring tiles share (and the label maps under them) are fetched into one L2.  (One workgroup per tile: a resident grid of
  // tile walkers was measured 3 - 4 % slower and read 17 % more - the tiles in flight at one time are then a stride apart.)
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
  if (tile >= p.ntiles) return;
  const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
  const int y0 = tyi * TH, x0 = txi * TW;
  const int H = p.H, W = p.W, HWp = H * W;

  for (int i = tid; i < NKIND * C; i += 256) {
    const int kind = i / C, c = i - kind * C;
    float v = 0.f;
    if (kind == 0) v = (p.mean + z * p.stat_bstride)[c];
    else if (kind == 1) v = (p.rstd + z * p.stat_bstride)[c];
    else {
      const int k = (kind - 2) >> 1;
      if (k < p.nmaps) v = (((kind & 1) ? p.bb[k] : p.bg[k]) + z * p.b_bstride[k])[c];
    }
    cst[i] = v;
  }

  // the convolution's first weight chunk is requested now and lands in LDS behind phase 1
  const fsv_buf cbuf = fsv_make_buf(fsv_s3_uniform(p.wc), (long long)9 * C * p.ldwc * 4);
  float4 wreg[WQ];
  auto load_wchunk = [&](int ci) {
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
      const int e = tid + 256 * i;                       // RW rows x (8 TN2) quads
      const int row = e / (8 * TN2), q4 = e - row * (8 * TN2);
      wreg[i] = fsv_buf_load4(cbuf, (unsigned)(((ci * RW + row) * p.ldwc + 4 * q4) * 4));
    }
  };
  auto store_wchunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
      const int e = tid + 256 * i;
      const int row = e / (8 * TN2), q4 = e - row * (8 * TN2);
      *reinterpret_cast<float4*>(&wch[(buf * RW + row) * BNC + 4 * q4]) = wreg[i];
    }
  };
  load_wchunk(0);
  __syncthreads();                                       // publishes cst

  // ---- phase 1: h of the haloed tile into the LDS patch ------------------------------------------------------------------------------
  {
    const int cb = wave & 1, pb0 = NU * (wave >> 1);
    int pix[NU];                                         // image pixel of this lane's halo pixel per unit, -1 outside
    bool inner[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int t = 32 * (pb0 + u) + lrow;
      const int hy = t / HWD, hx = t - hy * HWD;
      const int y = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = (t < HP) & (y >= 0) & (y < H) & (xx >= 0) & (xx < W);
      pix[u] = ok ? y * W + xx : -1;
      inner[u] = ok & (hy >= 1) & (hy <= TH) & (hx >= 1) & (hx <= TW);
    }
    const long long xpix_n = p.up ? (HWp >> 2) : HWp;
    const fsv_buf xbuf = fsv_make_buf(fsv_s3_uniform(p.x + (long long)z * xpix_n * C), xpix_n * C * 4);
    float xv[NU][16];                                    // x, then the running modulated value: register r = channel c(r)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      int sp = pix[u];
      if (p.up && sp >= 0) {
        const int y = sp / W, xx = sp - y * W;
        sp = (y >> 1) * (W >> 1) + (xx >> 1);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = fsv_buf_load4(xbuf, sp >= 0 ? (unsigned)((sp * C + 32 * cb + 8 * q + 4 * lk) * 4) : FSV_BUF_OOB);
        xv[u][4 * q] = v.x; xv[u][4 * q + 1] = v.y; xv[u][4 * q + 2] = v.z; xv[u][4 * q + 3] = v.w;
      }
    }
    f32x16 ag[NU], ab[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ag[u][r] = 0.f; ab[u][r] = 0.f; }

    // per map: its k groups (eight map channels each) with the operands one group ahead of the matrix instructions; the first group
    // of the NEXT map is requested under the last group of this one (a loop that started every map cold exposed one memory latency
    // per map).  (A flat sequence over all maps with run-time map indices was measured slower: descriptors rebuilt per group.)
    fsv_buf abuf[FSV_S3_MAXMAPS], gbuf[FSV_S3_MAXMAPS], bbuf[FSV_S3_MAXMAPS];
#pragma unroll
    for (int k = 0; k < FSV_S3_MAXMAPS; ++k) {
      const int kk = k < p.nmaps ? k : 0;
      const int Ch = p.ch[kk];
      const long long wbytes = (long long)((Ch + 31) / 32) * 32 * p.ldw * 4;
      abuf[k] = fsv_make_buf(fsv_s3_uniform(p.map[kk] + (long long)z * HWp * Ch), (long long)HWp * Ch * 4);
      gbuf[k] = fsv_make_buf(fsv_s3_uniform(p.wg[kk] + z * p.w_bstride[kk]), wbytes);
      bbuf[k] = fsv_make_buf(fsv_s3_uniform(p.wb[kk] + z * p.w_bstride[kk]), wbytes);
    }
    float4 mp[2][NU];
    float wgv[2][4], wbv[2][4];
    auto load_group = [&](auto KC, int j, int b) {
      constexpr int k = decltype(KC)::value;
      const int Ch = p.ch[k];
      const int kk = 8 * j + 4 * lk;                     // this lane's four k of the group: steps s = 0 .. 3
      const bool kin = (kk < Ch) & (k < p.nmaps);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        mp[b][u] = fsv_buf_load4(abuf[k], (kin & (pix[u] >= 0)) ? (unsigned)((pix[u] * Ch + kk) * 4) : FSV_BUF_OOB);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // no mask: rows [Ch, ceil32(Ch)) of the operand are zeros (fsv_spade_prep), rows past it are outside the descriptor
        // (a select here came back as a branch around the loads with a vmcnt(0) inside - guide trap 4c)
        const unsigned off = (unsigned)(((kk + s) * p.ldw + 32 * cb + lrow) * 4);
        wgv[b][s] = fsv_buf_load1(gbuf[k], off);
        wbv[b][s] = fsv_buf_load1(bbuf[k], off);
      }
    };
    auto mma_group = [&](int b) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const float m = s == 0 ? mp[b][u].x : (s == 1 ? mp[b][u].y : (s == 2 ? mp[b][u].z : mp[b][u].w));
          // operands swapped: rows of D = channels (the weight fragment), columns = pixels (the map fragment)
          ag[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wgv[b][s], m, ag[u], 0, 0, 0);
          ab[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wbv[b][s], m, ab[u], 0, 0, 0);
        }
    };
    auto one_map = [&](auto KC) {
      constexpr int k = decltype(KC)::value;
      constexpr int kn = k + 1 < FSV_S3_MAXMAPS ? k + 1 : k;
      const int ngrp = (p.ch[k] + 7) / 8;
#pragma unroll 1
      for (int j = 0; j < ngrp; j += 2) {
        // (fences: left alone the scheduler sinks the loads below the MFMAs they are meant to hide behind and waits for them with
        // vmcnt(0) at the top of the next group)
        load_group(KC, j + 1, 1);                        // (past the map's end: every lane out of range, zeros, never used)
        FSV_SCHED_FENCE();
        mma_group(0);
        FSV_SCHED_FENCE();
        if (j + 2 < ngrp) load_group(KC, j + 2, 0);
        else if (k + 1 < FSV_S3_MAXMAPS && k + 1 < p.nmaps) load_group(std::integral_constant<int, kn>{}, 0, 0);
        FSV_SCHED_FENCE();
        if (j + 1 < ngrp) mma_group(1);
        FSV_SCHED_FENCE();
      }
      // modulation with map k (registers only)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c4 = 32 * cb + 8 * q + 4 * lk;
        const float4 g4 = *reinterpret_cast<const float4*>(&cst[(2 + 2 * k) * C + c4]);
        const float4 b4 = *reinterpret_cast<const float4*>(&cst[(3 + 2 * k) * C + c4]);
        const float4 m4 = *reinterpret_cast<const float4*>(&cst[c4]);
        const float4 r4 = *reinterpret_cast<const float4*>(&cst[C + c4]);
        const float gq[4] = {g4.x, g4.y, g4.z, g4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
        const float mu[4] = {m4.x, m4.y, m4.z, m4.w}, rs[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float o = k == 0 ? (xv[u][r] - mu[e]) * rs[e] : xv[u][r];
            const float gk = ag[u][r] + gq[e];
            xv[u][r] = o * (1.f + gk) + (ab[u][r] + bq[e]);
            ag[u][r] = 0.f; ab[u][r] = 0.f;
          }
      }
    };
    load_group(std::integral_constant<int, 0>{}, 0, 0);
    one_map(std::integral_constant<int, 0>{});
    if (p.nmaps > 1) one_map(std::integral_constant<int, 1>{});
    if (p.nmaps > 2) one_map(std::integral_constant<int, 2>{});
    // activation, zero outside the image, into the patch (+ the side output for the pixels this tile owns)
    float* hs_z = p.hs ? p.hs + (long long)z * HWp * C : nullptr;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int t = 32 * (pb0 + u) + lrow;
      const bool ok = pix[u] >= 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v;
        v.x = ok ? fsv_act(xv[u][4 * q], p.act) : 0.f;
        v.y = ok ? fsv_act(xv[u][4 * q + 1], p.act) : 0.f;
        v.z = ok ? fsv_act(xv[u][4 * q + 2], p.act) : 0.f;
        v.w = ok ? fsv_act(xv[u][4 * q + 3], p.act) : 0.f;
        *reinterpret_cast<float4*>(&patch[t * PS + 32 * cb + 8 * q + 4 * lk]) = v;
        if (hs_z && inner[u]) *reinterpret_cast<float4*>(&hs_z[(long long)pix[u] * C + 32 * cb + 8 * q + 4 * lk]) = v;
      }
    }
  }
  store_wchunk(0);
  __syncthreads();                                       // patch and the first weight chunk are complete

  // ---- phase 2: the 3x3 convolution from the patch -------------------------------------------------------------------------------------
  // (one output column block: the k steps alternate between two accumulators - two independent chains of matrix instructions -
  // and meet in the epilogue)
  constexpr int NACC = TN2 == 1 ? 2 : TN2;
  f32x16 acc2[NACC];
#pragma unroll
  for (int jn = 0; jn < NACC; ++jn)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[jn][r] = 0.f;
  const int oy_a = 2 * wave + (lrow >> 4), ox_a = lrow & 15;        // this lane's pixel as the A row of its wave's block
  int buf = 0;
  float4 fa[2];
  float fb[2][4][TN2];
#pragma unroll 1
  for (int ci = 0; ci < NCHUNK; ++ci) {
    if (ci + 1 < NCHUNK) load_wchunk(ci + 1);
    const int tap = ci * RW / C, cbase = ci * RW - tap * C;
    const int ty = tap / 3, tx = tap - 3 * ty;
    const float* arow = patch + ((oy_a + ty) * HWD + ox_a + tx) * PS + cbase + 4 * lk;
    const float* brow = wch + (buf * RW + 4 * lk) * BNC + lrow;
    auto read_group = [&](int jg, int b) {
      fa[b] = *reinterpret_cast<const float4*>(arow + 8 * jg);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) fb[b][s][jn] = brow[(8 * jg + s) * BNC + 32 * jn];
    };
    auto mma_group = [&](int b) {
      const float av[4] = {fa[b].x, fa[b].y, fa[b].z, fa[b].w};
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) {
          const int a = TN2 == 1 ? (s & 1) : jn;
          acc2[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], fb[b][s][jn], acc2[a], 0, 0, 0);
        }
    };
    // the next group's fragments are read from LDS under this group's matrix instructions
    read_group(0, 0);
#pragma unroll
    for (int jg = 0; jg < RW / 8; ++jg) {
      if (jg + 1 < RW / 8) read_group(jg + 1, (jg + 1) & 1);
      FSV_SCHED_FENCE();
      mma_group(jg & 1);
      FSV_SCHED_FENCE();
    }
    if (ci + 1 < NCHUNK) store_wchunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  if constexpr (TN2 == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[0][r] += acc2[1][r];
  }

  // ---- epilogue: D layout col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the wave's block) ------------
  const float sc = p.wscale ? p.wscale[0] : 1.f;
  float* out_z = p.out + (long long)z * HWp * p.Cout;
  const float* res_z = p.res ? p.res + (long long)z * HWp * p.Cout : nullptr;
#pragma unroll
  for (int jn = 0; jn < TN2; ++jn) {
    const int co = 32 * jn + lrow;
    const float bv = p.bias ? p.bias[co] : 0.f;
    float s0 = 0.f, q0 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int y = y0 + 2 * wave + (m >> 4), xx = x0 + (m & 15);
      if (y < H && xx < W) {
        const long long o = ((long long)y * W + xx) * p.Cout + co;
        float v = acc2[jn][r] * sc + bv;
        if (res_z) v += res_z[o];
        out_z[o] = v;
        s0 += v; q0 += v * v;
      }
    }
    if (p.stats) {            // uniform
      // the two half-waves hold the same 32 channels: fold them, then one fp64 atomic pair per channel and wave (conv_igemm.hip)
      s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32);
      if (lk == 0) {
        double* d = p.stats + ((long long)(tile % p.stats_slots) * p.Cout + co) * 2;
        atomicAdd(d, (double)s0); atomicAdd(d + 1, (double)q0);
      }
    }
  }
}

extern "C" {

// 1 when fsv_spade_conv3_fwd has a kernel for this geometry
int fsv_spade_conv3_supported(int C, int Cout, int nmaps) {
  return C == 64 && (Cout == 32 || Cout == 64) && nmaps >= 1 && nmaps <= FSV_S3_MAXMAPS;
}

// out = conv3x3(act(spade(x; maps)), wc) * wscale + bias (+ res) in one launch (see the file comment).  Operands of the modulation
// as fsv_spade_mod_fwd (spade.hip): x [N][H W][C] (up != 0: [N][H W / 4][C], read through the nearest-x2 index), mean / rstd [C],
// per map k: maps[k] [N][H W][ch[k]] (ch % 4 == 0), wg[k] / wb[k] K-major [ceil32(ch)][ldw] gamma / beta operands, bg[k] / bb[k] [C],
// w_bstride / b_bstride per-sample strides in floats (0: shared).  wc = the K-major forward operand of the 3x3 weight
// ([9 C rows = (tap, ci)][ldwc], fsv_prep_weight mode 0), padding 1, stride 1.  hs (optional) receives the modulated + activated
// tensor.  stats (optional): fp64 partials [stats_slots][Cout][2] of the stored output's per-channel (sum, sum of squares) over all
// samples - the BatchNorm statistics of the normalisation that follows, finished by fsv_norm_stats_finish like the gather-GEMM's
// (fsv_conv_gather_fwd_stats, one group); zeroed here unless stats_prezeroed.
// FSV_ERR_UNSUPPORTED for geometries without a kernel (fsv_spade_conv3_supported).
int fsv_spade_conv3_fwd(const float* x, const float* mean, const float* rstd, float* hs, float* out,
                        int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, int N, int H, int W, int C, int ldw, long long stat_bstride, int up, int act,
                        const float* wc, int ldwc, int Cout, const float* bias, const float* res, const float* wscale,
                        double* stats, int stats_slots, int stats_prezeroed, hipStream_t stream) {
  if (!x || !mean || !rstd || !out || !wc || !maps || !wg || !wb || !bg || !bb || !ch || N < 1 || H < 1 || W < 1 || (ldw & 3) ||
      ldwc < Cout || (ldwc & 3))
    return FSV_ERR_BAD_ARG;
  if (!fsv_spade_conv3_supported(C, Cout, nmaps)) return FSV_ERR_UNSUPPORTED;
  if (act != FSV_ACT_NONE && act != FSV_ACT_LRELU) return FSV_ERR_UNSUPPORTED;
  if (up && ((W & 1) || (H & 1))) return FSV_ERR_BAD_ARG;
  const long long HW = (long long)H * W;
  if (HW * C * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  SpadeConv3P p;
  p.x = x; p.mean = mean; p.rstd = rstd; p.hs = hs; p.out = out;
  for (int k = 0; k < FSV_S3_MAXMAPS; ++k) {
    const bool on = k < nmaps;
    p.map[k] = on ? maps[k] : nullptr; p.wg[k] = on ? wg[k] : nullptr; p.wb[k] = on ? wb[k] : nullptr;
    p.bg[k] = on ? bg[k] : nullptr; p.bb[k] = on ? bb[k] : nullptr;
    p.ch[k] = on ? ch[k] : 0; p.w_bstride[k] = on ? w_bstride[k] : 0; p.b_bstride[k] = on ? b_bstride[k] : 0;
    if (on && (!maps[k] || !wg[k] || !wb[k] || !bg[k] || !bb[k] || ch[k] < 1 || (ch[k] & 3))) return FSV_ERR_UNSUPPORTED;
    if (on && HW * ch[k] * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  }
  p.nmaps = nmaps; p.N = N; p.H = H; p.W = W; p.C = C; p.ldw = ldw; p.stat_bstride = stat_bstride;
  p.up = up ? 1 : 0; p.act = act;
  p.wc = wc; p.bias = bias; p.res = res; p.wscale = wscale; p.Cout = Cout; p.ldwc = ldwc;
  p.stats = nullptr; p.stats_slots = 1;
  if (stats) {
    if (stats_slots < 1) return FSV_ERR_BAD_ARG;
    p.stats = stats; p.stats_slots = stats_slots;
    if (!stats_prezeroed) (void)hipMemsetAsync(stats, 0, (size_t)stats_slots * Cout * 2 * sizeof(double), stream);
  }
  p.tiles_x = fsv_cdiv(W, 16);
  p.ntiles = p.tiles_x * fsv_cdiv(H, 8);
  dim3 g((unsigned)(((p.ntiles + 7) / 8) * 8), 1, N);
  const char* e = getenv("FSV_S3_RW");                   // in-box A/B: 32-row weight chunks for Cout 32 as well (64: half the
  const bool rw64 = !(e && atoi(e) == 32);               // barriers, measured 1 - 2.5 % faster)
  if (Cout == 32 && rw64) FSV_LAUNCH((fsv_spade_conv3_kernel<1, 64>), g, dim3(256), stream, p);
  else if (Cout == 32) FSV_LAUNCH((fsv_spade_conv3_kernel<1, 32>), g, dim3(256), stream, p);
  else FSV_LAUNCH((fsv_spade_conv3_kernel<2, 32>), g, dim3(256), stream, p);
  return fsv_check_launch();
}

}  // extern "C"
