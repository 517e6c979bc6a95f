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

struct SpadeP {
  const float* x;         // [N][HW][C]
  const float* mean;      // [C] (or [N][C] when stat_bstride != 0)
  const float* rstd;
  float* h;               // [N][HW][C]
  const float* map[FSV_SP_MAXMAPS];    // [N][HW][Ch_k]
  const float* wg[FSV_SP_MAXMAPS];     // K-major [Kpad_k][ldw] (+ z * w_bstride_k)
  const float* wb[FSV_SP_MAXMAPS];
  const float* bg[FSV_SP_MAXMAPS];     // [C] (+ z * b_bstride_k)
  const float* bb[FSV_SP_MAXMAPS];
  int ch[FSV_SP_MAXMAPS];
  long long w_bstride[FSV_SP_MAXMAPS];
  long long b_bstride[FSV_SP_MAXMAPS];
  int nmaps;
  int N, HW, C, ldw;
  long long stat_bstride;
  int act;
  // backward twin (BWD = true): upstream gradient in, d(gamma|beta) per map ([P][2C]: gamma columns [0, C), beta [C, 2C)) and
  // d(xhat) out; h is not written
  const float* dh;
  float* dgb[FSV_SP_MAXMAPS];
  float* dxhat;
  int W, up;              // up = 1: x is the HALF-resolution tensor [N][H/2][W/2][C] and is read through the nearest-x2
                          // up-sampling index (generator.py:124 folded into this kernel: the up-sampled tensor is never written)
};

// source element of x for output pixel m of sample z (identity, or the parent pixel of the nearest x2 up-sampling)
__device__ __forceinline__ long long fsv_sp_xpix(const SpadeP& p, int z, int m) {
  if (!p.up) return (long long)z * p.HW + m;
  const int y = m / p.W, xx = m - y * p.W;
  return (long long)z * (p.HW >> 2) + (long long)(y >> 1) * (p.W >> 1) + (xx >> 1);
}

// BWD = true is the backward twin: the same two GEMMs recompute gamma / beta of every map in registers (they never reach HBM
// in either direction), the forward chain o_0 = xhat, o_{k+1} = o_k (1 + g_k) + b_k is replayed keeping g_k and o_k, and the
// epilogue walks it backwards:  d = dh * act'(o_n);  for k = n-1 .. 0:  dbeta_k = d,  dgamma_k = d * o_k,  d *= (1 + g_k);
// dxhat = d.  (Round 1 materialised gamma | beta as [P][2C] per map with the gather-GEMM kernel and read them back in an
// element-wise pass: 236 MB read + 175 MB written per launch more than this.)
template <int BM, int BN, int WM, int WN, bool BWD>
__global__ __launch_bounds__(256) void fsv_spade_mod_kernel(SpadeP p) {
  constexpr int BK = FSV_SP_BK;
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int LDA = BM + 1;
  constexpr int KV = BK / 4, RPP = 256 / KV, NPA = BM / RPP;
  constexpr int QB = BN / 4, RPB = 256 / QB, NPB = BK / RPB;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(NPA >= 1 && NPB >= 1, "tile");
  __shared__ float As[BK * LDA];
  __shared__ float Bg[BK * BN];
  __shared__ float Bb[BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int z = blockIdx.z;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const int lrow = lane & 31, lk = lane >> 5;
  const int kq = tid % KV, ar0 = tid / KV;
  const int bq = tid % QB, br0 = tid / QB;
  const int bcol = bn0 + bq * 4;
  const bool bcol_ok = bcol < p.ldw;
  const int bcol_safe = bcol_ok ? bcol : 0;

  // running value of the normalised + modulated activation, in MFMA C/D layout
  f32x16 outv[TM][TN];
  // backward twin: g_k and o_k of every map (BWD only; dead code otherwise)
  f32x16 keep_g[BWD ? FSV_SP_MAXMAPS : 1][TM][TN], keep_o[BWD ? FSV_SP_MAXMAPS : 1][TM][TN];
  const float* mean = p.mean + z * p.stat_bstride;
  const float* rstd = p.rstd + z * p.stat_bstride;
  const long long pix0 = (long long)z * p.HW;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
    const bool cok = c < p.C;
    const float mu = cok ? mean[c] : 0.f, rs = cok ? rstd[c] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = bm0 + wm * (TM * 32) + i * 32 + row;
        float v = 0.f;
        if (cok && m < p.HW) v = (p.x[fsv_sp_xpix(p, z, m) * p.C + c] - mu) * rs;
        outv[i][j][r] = v;
      }
  }

#pragma unroll
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
    if (k >= p.nmaps) break;
    const float* mp = p.map[k] + pix0 * p.ch[k];
    const float* wg = p.wg[k] + z * p.w_bstride[k];
    const float* wb = p.wb[k] + z * p.w_bstride[k];
    const int Ch = p.ch[k];
    const int nchunks = (Ch + BK - 1) / BK;
    f32x16 accg[TM][TN], accb[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accg[i][j][r] = 0.f; accb[i][j][r] = 0.f; }

    float4 areg[NPA], gbreg[NPB], bbreg[NPB];
    auto load_chunk = [&](int kc) {
      const int kk = kc * BK + kq * 4;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        int m = bm0 + ar0 + i * RPP;
        bool ok = kk < Ch && m < p.HW;
        float4 v = *reinterpret_cast<const float4*>(mp + (ok ? ((long long)m * Ch + kk) : 0ll));
        areg[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        int kr = kc * BK + br0 + i * RPB;
        float4 vg = *reinterpret_cast<const float4*>(wg + (long long)kr * p.ldw + bcol_safe);
        float4 vb = *reinterpret_cast<const float4*>(wb + (long long)kr * p.ldw + bcol_safe);
        gbreg[i] = bcol_ok ? vg : make_float4(0.f, 0.f, 0.f, 0.f);
        bbreg[i] = bcol_ok ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        int r = ar0 + i * RPP;
        As[(kq * 4 + 0) * LDA + r] = areg[i].x;
        As[(kq * 4 + 1) * LDA + r] = areg[i].y;
        As[(kq * 4 + 2) * LDA + r] = areg[i].z;
        As[(kq * 4 + 3) * LDA + r] = areg[i].w;
      }
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        int kr = br0 + i * RPB;
        *reinterpret_cast<float4*>(&Bg[kr * BN + bq * 4]) = gbreg[i];
        *reinterpret_cast<float4*>(&Bb[kr * BN + bq * 4]) = bbreg[i];
      }
    };
    __syncthreads();            // previous map's LDS reads are finished
    load_chunk(0);
    store_chunk();
    __syncthreads();
#pragma unroll 1
    for (int kc = 0; kc < nchunks; ++kc) {
      const int knext = (kc + 1 < nchunks) ? kc + 1 : kc;
      load_chunk(knext);
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float a[TM], g[TN], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(kk * 2 + lk) * LDA + wm * (TM * 32) + i * 32 + lrow];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          g[j] = Bg[(kk * 2 + lk) * BN + wn * (TN * 32) + j * 32 + lrow];
          b[j] = Bb[(kk * 2 + lk) * BN + wn * (TN * 32) + j * 32 + lrow];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            accg[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], g[j], accg[i][j], 0, 0, 0);
            accb[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], accb[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
      store_chunk();
      __syncthreads();
    }
    // modulation epilogue for this map
    const float* bg = p.bg[k] + z * p.b_bstride[k];
    const float* bb = p.bb[k] + z * p.b_bstride[k];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
      const float bgv = (c < p.C) ? bg[c] : 0.f, bbv = (c < p.C) ? bb[c] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float gk = accg[i][j][r] + bgv;
          if constexpr (BWD) { keep_g[k][i][j][r] = gk; keep_o[k][i][j][r] = outv[i][j][r]; }
          outv[i][j][r] = outv[i][j][r] * (1.f + gk) + (accb[i][j][r] + bbv);
        }
    }
  }

  if constexpr (BWD) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
      if (c >= p.C) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int m = bm0 + wm * (TM * 32) + i * 32 + row;
          if (m >= p.HW) continue;
          const long long pix = pix0 + m;
          float d = p.dh[pix * p.C + c];
          if (p.act == FSV_ACT_LRELU) d = (outv[i][j][r] > 0.f) ? d : 0.2f * d;
#pragma unroll
          for (int k = FSV_SP_MAXMAPS - 1; k >= 0; --k) {
            if (k < p.nmaps) {
              float* dg = p.dgb[k] + pix * 2 * p.C;
              dg[p.C + c] = d;
              dg[c] = d * keep_o[k][i][j][r];
              d = d * (1.f + keep_g[k][i][j][r]);
            }
          }
          p.dxhat[pix * p.C + c] = d;
        }
    }
    return;
  }

#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int c = bn0 + wn * (TN * 32) + j * 32 + lrow;
    if (c >= p.C) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = bm0 + wm * (TM * 32) + i * 32 + row;
        if (m < p.HW) p.h[(pix0 + m) * p.C + c] = fsv_act(outv[i][j][r], p.act);
      }
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

// maps/wg/wb/bg/bb: arrays of nmaps device pointers; ch / w_bstride / b_bstride: per-map ints / strides.
int fsv_spade_mod_fwd(const float* x, const float* mean, const float* rstd, float* h,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                      int W, int up, hipStream_t stream) {
  if (!x || !mean || !rstd || !h || nmaps < 0 || nmaps > FSV_SP_MAXMAPS || C < 1 || (ldw & 3)) return FSV_ERR_BAD_ARG;
  if (up && (W < 2 || (W & 1) || HW % W != 0 || ((HW / W) & 1))) return FSV_ERR_BAD_ARG;
  SpadeP p;
  p.x = x; p.mean = mean; p.rstd = rstd; p.h = h; p.nmaps = nmaps;
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
    bool on = k < nmaps;
    p.map[k] = on ? maps[k] : nullptr; p.wg[k] = on ? wg[k] : nullptr; p.wb[k] = on ? wb[k] : nullptr;
    p.bg[k] = on ? bg[k] : nullptr; p.bb[k] = on ? bb[k] : nullptr;
    p.ch[k] = on ? ch[k] : 0; p.w_bstride[k] = on ? w_bstride[k] : 0; p.b_bstride[k] = on ? b_bstride[k] : 0;
    if (on && ((ch[k] & 3) || !maps[k] || !wg[k] || !wb[k] || !bg[k] || !bb[k])) return FSV_ERR_UNSUPPORTED;
  }
  p.N = N; p.HW = HW; p.C = C; p.ldw = ldw; p.stat_bstride = stat_bstride; p.act = act;
  p.W = up ? W : 1; p.up = up ? 1 : 0;
  if (C <= 32) {
    dim3 g(fsv_cdiv(HW, 128), fsv_cdiv(C, 32), N);
    FSV_LAUNCH((fsv_spade_mod_kernel<128, 32, 4, 1, false>), g, dim3(256), stream, p);
  } else {
    dim3 g(fsv_cdiv(HW, 128), fsv_cdiv(C, 64), N);
    FSV_LAUNCH((fsv_spade_mod_kernel<128, 64, 2, 2, false>), g, dim3(256), stream, p);
  }
  return fsv_check_launch();
}

// Backward twin of fsv_spade_mod_fwd (see the kernel comment): same operands, dh in, dgb[k] ([P][2C] per map) and dxhat out.
int fsv_spade_mod_bwd(const float* x, const float* mean, const float* rstd, const float* dh,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, float* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                      long long stat_bstride, int act, int W, int up, hipStream_t stream) {
  if (!x || !mean || !rstd || !dh || !dgb || !dxhat || nmaps < 0 || nmaps > FSV_SP_MAXMAPS || C < 1 || (ldw & 3)) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_LRELU && act != FSV_ACT_NONE) return FSV_ERR_UNSUPPORTED;
  if (up && (W < 2 || (W & 1) || HW % W != 0 || ((HW / W) & 1))) return FSV_ERR_BAD_ARG;
  SpadeP p;
  p.x = x; p.mean = mean; p.rstd = rstd; p.h = nullptr; p.nmaps = nmaps; p.dh = dh; p.dxhat = dxhat;
  for (int k = 0; k < FSV_SP_MAXMAPS; ++k) {
    bool on = k < nmaps;
    p.map[k] = on ? maps[k] : nullptr; p.wg[k] = on ? wg[k] : nullptr; p.wb[k] = on ? wb[k] : nullptr;
    p.bg[k] = on ? bg[k] : nullptr; p.bb[k] = on ? bb[k] : nullptr; p.dgb[k] = on ? dgb[k] : nullptr;
    p.ch[k] = on ? ch[k] : 0; p.w_bstride[k] = on ? w_bstride[k] : 0; p.b_bstride[k] = on ? b_bstride[k] : 0;
    if (on && ((ch[k] & 3) || !maps[k] || !wg[k] || !wb[k] || !bg[k] || !bb[k] || !dgb[k])) return FSV_ERR_UNSUPPORTED;
  }
  p.N = N; p.HW = HW; p.C = C; p.ldw = ldw; p.stat_bstride = stat_bstride; p.act = act;
  p.W = up ? W : 1; p.up = up ? 1 : 0;
  // 64 x 64 tiles: every wave keeps g_k and o_k of up to three maps for a 32 x 32 sub-tile (96 + 48 accumulator registers),
  // which leaves room for several workgroups per CU - the kernel is HBM bound
  dim3 g(fsv_cdiv(HW, 64), fsv_cdiv(C, 64), N);
  FSV_LAUNCH((fsv_spade_mod_kernel<64, 64, 2, 2, true>), g, dim3(256), stream, p);
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
