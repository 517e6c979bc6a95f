// Implicit-GEMM convolution on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// One "gather-GEMM" kernel covers every dense contraction of the few-shot-vid2vid G/D step
// (SURVEY.md section 8a rows a-1, a-3, a-5..a-9): 3x3 / 4x4 / 1x1 convolutions with stride 1 or 2, their data
// gradients (flipped taps; stride-2 dgrad is split into 4 output-parity classes), nn.Linear, and the
// per-sample "batch_conv" of the reference (models/networks/base_network.py:56-71) where every sample has
// its own generated weight matrix.  A second kernel computes weight gradients (reduction over pixels).
//
//   out[z][m][co] = sum_{t < ntaps} sum_{ci < Cin}  in[n, oy*sy + ty[t], ox*sx + tx[t], ci] * wt[z][t*Cin + ci][co]
//
// Layout: activations NHWC (channels contiguous) so the K dimension of the GEMM is contiguous in HBM;
// weights are pre-arranged K-major ([K_pad][ldw]) by fsv_prep_weight (which also applies the spectral-norm
// 1/sigma).  Tiles are staged through LDS: A is stored transposed ([k][m], row stride BM+1) so that the MFMA
// A-fragment read (32 consecutive m per half-wave) and the staging writes are both bank-conflict free.
// The MFMA result is bitwise an fp32 fma chain (guide section 3), which is what lets the parity tests use a
// 1e-3 relative tolerance against the fp32 CPU oracle with a wide margin.
#include "conv_igemm.h"

template <int BM, int BN, int WM, int WN, int V>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_kernel(ConvP p) {
  constexpr int BK = FSV_BK;
  constexpr int NT = 64 * WM * WN;    // work-items per workgroup (4, 8 or 16 waves)
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int LDA = BM + 1;
  constexpr int KV = BK / V;          // A vectors per pixel row and chunk
  constexpr int RPP = NT / KV;        // A rows per pass
  constexpr int NPA = BM / RPP;       // A passes
  constexpr int QB = BN / 4;          // B float4 per k row
  constexpr int RPB = NT / QB;        // B rows per pass
  constexpr int NPB = BK / RPB;       // B passes
  static_assert(NPA >= 1 && NPB >= 1 && NPA * RPP == BM && NPB * RPB == BK, "tile / thread-count mismatch");
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = blockIdx.z / p.nsplit, zk = blockIdx.z % p.nsplit;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const float* wt = p.wt + (long long)zs * p.w_bstride;

  // ---- per-thread A row bookkeeping ------------------------------------------------------------------
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

  // chunk range of this K split
  const int cps = (p.nchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.nchunks) ? (c_begin + cps) : p.nchunks;

  float areg[NPA][V];
  float4 breg[NPB];

  // Loads are unconditional (out-of-range rows / taps read a valid dummy address and are zeroed by a select):
  // no per-lane branches, so the whole K loop stays one basic block and the accumulators stay in AGPRs.
  auto load_chunk = [&](int kc) {
    const int k = kc * BK + kq * V;
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
      const float* src = p.in + off;
      if constexpr (V == 4) {
        float4 v = *reinterpret_cast<const float4*>(src);
        areg[i][0] = ok ? v.x : 0.f; areg[i][1] = ok ? v.y : 0.f; areg[i][2] = ok ? v.z : 0.f; areg[i][3] = ok ? v.w : 0.f;
      } else {
        float v = *src;
        areg[i][0] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int kr = kc * BK + br0 + i * RPB;
      float4 v = *reinterpret_cast<const float4*>(wt + (long long)kr * p.ldw + bcol_safe);
      breg[i] = bcol_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int r = ar0 + i * RPP;
#pragma unroll
      for (int j = 0; j < V; ++j) As[(kq * V + j) * LDA + r] = areg[i][j];
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int kr = br0 + i * RPB;
      *reinterpret_cast<float4*>(&Bs[kr * BN + bq * 4]) = breg[i];
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
  const float* a_frag = &As[lk * LDA + wm * (TM * 32) + lrow];
  const float* b_frag = &Bs[lk * BN + wn * (TN * 32) + lrow];
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
    __syncthreads();
#pragma unroll 1
    for (int kc = c_begin; kc < c_end; ++kc) {
      // prefetch the next chunk (the last iteration re-reads its own chunk; the copy it stores is never used)
      const int knext = (kc + 1 < c_end) ? kc + 1 : kc;
      load_chunk(knext);
      // fragments are double-buffered in registers: the LDS reads of k-step kk+1 are in flight under the MFMAs of kk
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = a_frag[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = b_frag[j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[nxt][i] = a_frag[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[nxt][j] = b_frag[(kk + 1) * 2 * BN + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      store_chunk();
      __syncthreads();
    }
  }

  // ---- epilogue: D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) -----------
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

// ---- finishing pass for split-K launches: out = act((out + bias) * scale) + res ---------------------------
__global__ __launch_bounds__(256) void fsv_bias_act_kernel(float* out, const float* bias, const float* res,
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

// ---- weight gradient: dwt[z][t*Cin+ci][co] (+)= sum_pixels in[n, oy*sy+ty, ox*sx+tx, ci] * dout[n,oy,ox,co] ----
template <int BMK, int BN, int WM, int WN, int V>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_wgrad_kernel(WgradP p) {
  constexpr int BK = FSV_BK;   // pixels per chunk
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BMK / (WM * 32), TN = BN / (WN * 32);
  constexpr int QA = BMK / V, RPA = NT / QA, NPA = BK / RPA;
  constexpr int QB = BN / 4, RPB = NT / QB, NPB = BK / RPB;
  static_assert(TM >= 1 && TN >= 1, "tile");
  static_assert(NPA >= 1 && NPB >= 1 && RPA >= 1 && NPA * RPA == BK && NPB * RPB == BK, "tile / thread-count mismatch");
  __shared__ float As[BK * BMK];
  __shared__ float Bs[BK * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = blockIdx.z / p.nsplit, zk = blockIdx.z % p.nsplit;
  const int bi0 = blockIdx.x * BMK, bn0 = blockIdx.y * BN;
  float* dwt = p.dwt + (long long)zs * p.w_bstride;
  const int ohw = p.OH * p.OW;

  // A: column (t,ci) handled by this thread is fixed for the whole reduction
  const int aq = tid % QA, apr0 = tid / QA;
  const int kcol = bi0 + aq * V;
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

  float areg[NPA][V];
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
      const float* src = p.in + off;
      if constexpr (V == 4) {
        float4 v = *reinterpret_cast<const float4*>(src);
        areg[i][0] = ok ? v.x : 0.f; areg[i][1] = ok ? v.y : 0.f; areg[i][2] = ok ? v.z : 0.f; areg[i][3] = ok ? v.w : 0.f;
      } else {
        float v = *src;
        areg[i][0] = ok ? v : 0.f;
      }
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
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      int pr = apr0 + i * RPA;
#pragma unroll
      for (int j = 0; j < V; ++j) As[pr * BMK + aq * V + j] = areg[i][j];
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int pr = bpr0 + i * RPB;
      *reinterpret_cast<float4*>(&Bs[pr * BN + bq * 4]) = breg[i];
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
  const float* a_frag = &As[lk * BMK + wm * (TM * 32) + lrow];
  const float* b_frag = &Bs[lk * BN + wn * (TN * 32) + lrow];
  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
    __syncthreads();
#pragma unroll 1
    for (int pc = c_begin; pc < c_end; ++pc) {
      const int pnext = (pc + 1 < c_end) ? pc + 1 : pc;
      load_chunk(pnext);
      float a[2][TM], b[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = a_frag[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = b_frag[j * 32];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[nxt][i] = a_frag[(kk + 1) * 2 * BMK + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[nxt][j] = b_frag[(kk + 1) * 2 * BN + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
      store_chunk();
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

// ---- weight re-arrangement --------------------------------------------------------------------------------
// mode 0 (forward):  wt[z][j*Cin + ci][co] = s * w[z][co][ci][kh_j][kw_j]
// mode 1 (dgrad):    wt[z][j*Cout + co][ci] = s * w[z][co][ci][kh_j][kw_j]
// mode 2 (inverse of mode 0, for gradients): w[z][co][ci][kh_j][kw_j] = s * wt[z][j*Cin + ci][co]
// mode 3: as mode 2 but accumulating (+=) - gradients written straight into a flat optimiser buffer
// taps: (kh | kw<<4) per tap.  Rows >= K and columns >= ncols of wt are written as zero (modes 0/1).
__global__ __launch_bounds__(256) void fsv_prep_weight_kernel(const float* w, float* wt, const float* scale_ptr,
                                                              int mode_in, int Cout, int Cin, int KH, int KW,
                                                              int ntaps, unsigned long long taps_lo,
                                                              unsigned long long taps_hi, int Kpad, int ldw,
                                                              long long w_bstride, long long wt_bstride) {
  const int z = blockIdx.z;
  const float s = scale_ptr ? *scale_ptr : 1.f;
  const long long total = (long long)Kpad * ldw;
  const bool accum = mode_in == 3;
  const int mode = accum ? 2 : mode_in;
  const int rowlen = (mode == 1) ? Cout : Cin;      // channels per tap in the K dimension
  const int ncols = (mode == 1) ? Cin : Cout;
  const int K = ntaps * rowlen;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    int r = (int)(i / ldw), c = (int)(i - (long long)r * ldw);
    bool ok = r < K && c < ncols;
    int j = ok ? r / rowlen : 0;
    int a = r - j * rowlen;
    unsigned long long code = (j < 8) ? taps_lo : taps_hi;
    int sh = (j & 7) * 8;
    int kh = (int)((code >> sh) & 15ull), kw = (int)((code >> (sh + 4)) & 15ull);
    int co = (mode == 1) ? a : c, ci = (mode == 1) ? c : a;
    long long widx = (((long long)co * Cin + ci) * KH + kh) * KW + kw + (long long)z * w_bstride;
    if (mode == 2) {
      if (ok) {
        float g = s * wt[(long long)z * wt_bstride + i];
        ((float*)w)[widx] = accum ? ((float*)w)[widx] + g : g;
      }
    } else {
      wt[(long long)z * wt_bstride + i] = ok ? s * w[widx] : 0.f;
    }
  }
}

// ---- grouped re-arrangement: every parameter weight of an optimiser in ONE launch (run right after the Adam step) ----
// desc arrays (device): src / dst pointers as 64-bit integers; dims[l] = {Cout, Cin_pad, Cin_real, KH, KW, ntaps, Kpad, ldw,
// mode}; taps[l] = {lo, hi} packed (kh | kw << 4) codes.  No scaling here: the spectral-norm 1/sigma is applied in the GEMM
// epilogue.
struct PrepGroup {
  const long long* src; const long long* dst; const int* dims; const unsigned long long* taps;
};
// tmap[b] = (layout, 32-wide co tile, ci tile): the OIHW source tile [32 co][CI_T ci][KH*KW] is read in contiguous runs of
// CI_T * KK floats per output channel, staged in LDS and written out as rows of the K-major layout (CI_T = 32 for <= 8 source
// taps, else 16).  Only the valid region is written: the padding rows / columns of a layout are zero from allocation on.
__global__ __launch_bounds__(256) void fsv_prep_group_kernel(PrepGroup g, const int* tmap) {
  __shared__ float t[32 * 257];
  const int layer = tmap[blockIdx.x * 3], cot = tmap[blockIdx.x * 3 + 1], cit = tmap[blockIdx.x * 3 + 2];
  const int* d = g.dims + layer * 9;
  const int Cout = d[0], CinP = d[1], CinR = d[2], KW = d[4], ntaps = d[5], ldw = d[7], mode = d[8];
  const int KK = d[3] * KW;
  const int CI_T = KK <= 8 ? 32 : 16;
  const int run = CI_T * KK, lds = run + 1;
  const float* w = reinterpret_cast<const float*>(g.src[layer]);
  float* wt = reinterpret_cast<float*>(g.dst[layer]);
  const unsigned long long lo = g.taps[layer * 2], hi = g.taps[layer * 2 + 1];
  const int co0 = cot * 32, ci0 = cit * CI_T;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int col = wave; col < 32; col += 4) {
    const int co = co0 + col;
    for (int idx = lane; idx < run; idx += 64) {
      const int ci = ci0 + idx / KK;
      t[col * lds + idx] = (co < Cout && ci < CinR) ? w[((long long)co * CinR + ci0) * KK + idx] : 0.f;
    }
  }
  __syncthreads();
  if (mode == 0) {          // rows (tap j, ci), 32 consecutive output channels each
    const int col = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int co = co0 + col;
    for (int r = r0; r < ntaps * CI_T; r += 8) {
      const int j = r / CI_T, cil = r - j * CI_T;
      const int ci = ci0 + cil;
      if (ci >= CinP || co >= Cout) continue;
      const unsigned long long code = (j < 8) ? lo : hi;
      const int sh = (j & 7) * 8;
      const int tk = (int)((code >> sh) & 15ull) * KW + (int)((code >> (sh + 4)) & 15ull);
      wt[((long long)j * CinP + ci) * ldw + co] = t[col * lds + cil * KK + tk];
    }
  } else {                  // rows (tap j, co), CI_T consecutive input channels each
    const int cil = threadIdx.x % CI_T, c0 = threadIdx.x / CI_T, cstep = 256 / CI_T;
    const int ci = ci0 + cil;
    for (int j = 0; j < ntaps; ++j) {
      const unsigned long long code = (j < 8) ? lo : hi;
      const int sh = (j & 7) * 8;
      const int tk = (int)((code >> sh) & 15ull) * KW + (int)((code >> (sh + 4)) & 15ull);
      for (int col = c0; col < 32; col += cstep) {
        const int co = co0 + col;
        if (co >= Cout || ci >= CinP) continue;
        wt[((long long)j * Cout + co) * ldw + ci] = t[col * lds + cil * KK + tk];
      }
    }
  }
}

extern "C" int fsv_prep_weight_grouped(const long long* src, const long long* dst, const int* dims,
                                       const unsigned long long* taps, const int* tmap, int nblocks, hipStream_t stream) {
  if (!src || !dst || !dims || !taps || !tmap || nblocks < 1) return FSV_ERR_BAD_ARG;
  PrepGroup g; g.src = src; g.dst = dst; g.dims = dims; g.taps = taps;
  FSV_LAUNCH(fsv_prep_group_kernel, dim3(nblocks), dim3(256), stream, g, tmap);
  return fsv_check_launch();
}

// =============================================== host side ===================================================
static inline void fsv_pack_taps(const int* ty, const int* tx, int n, unsigned long long& lo, unsigned long long& hi,
                                 int bias) {
  lo = 0; hi = 0;
  for (int t = 0; t < n; ++t) {
    unsigned long long c = (unsigned long long)((ty[t] + bias) & 15) | ((unsigned long long)((tx[t] + bias) & 15) << 4);
    if (t < 8) lo |= c << (t * 8); else hi |= c << ((t - 8) * 8);
  }
}

template <int V>
static int fsv_launch_conv(const ConvP& p, int M_tiles_rows, int nz, hipStream_t stream, int tile) {
  dim3 block(256);
  if (V == 4) {
    switch (tile) {      // experimental large tiles (8 / 16 waves); only reachable through force_tile
      case 5: { dim3 g(fsv_cdiv(M_tiles_rows, 256), fsv_cdiv(p.Cout, 128), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<256, 128, 4, 2, 4>), g, dim3(512), stream, p); return fsv_check_launch(); }
      case 6: { dim3 g(fsv_cdiv(M_tiles_rows, 128), fsv_cdiv(p.Cout, 256), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<128, 256, 2, 4, 4>), g, dim3(512), stream, p); return fsv_check_launch(); }
      case 7: { dim3 g(fsv_cdiv(M_tiles_rows, 256), fsv_cdiv(p.Cout, 256), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<256, 256, 4, 4, 4>), g, dim3(1024), stream, p); return fsv_check_launch(); }
      case 8: { dim3 g(fsv_cdiv(M_tiles_rows, 256), fsv_cdiv(p.Cout, 64), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<256, 64, 4, 2, 4>), g, dim3(512), stream, p); return fsv_check_launch(); }
      case 9: { dim3 g(fsv_cdiv(M_tiles_rows, 64), fsv_cdiv(p.Cout, 128), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 2, 4>), g, block, stream, p); return fsv_check_launch(); }
      // few-wave workgroups (every wave owns a 64x64 sub-tile: one LDS fragment read per MFMA, and a one-wave workgroup needs
      // no cross-wave barrier traffic); not yet measured, only reachable through force_tile
      case 10: { dim3 g(fsv_cdiv(M_tiles_rows, 64), fsv_cdiv(p.Cout, 64), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 1, 1, 4>), g, dim3(64), stream, p); return fsv_check_launch(); }
      case 11: { dim3 g(fsv_cdiv(M_tiles_rows, 64), fsv_cdiv(p.Cout, 128), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 1, 2, 4>), g, dim3(128), stream, p); return fsv_check_launch(); }
      case 12: { dim3 g(fsv_cdiv(M_tiles_rows, 128), fsv_cdiv(p.Cout, 64), nz);
        FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 2, 1, 4>), g, dim3(128), stream, p); return fsv_check_launch(); }
      case 13: case 14: case 15: case 16: case 17: case 18: case 19: case 20: case 21:      // double-buffered LDS; 16-18 prefetch distance 2, 19-21 XCD-aware order
        return fsv_launch_conv_db(p, nz, stream, tile);
      default: break;
    }
  }
  if (tile == 9 || tile == 10 || tile == 11 || tile == 13 || tile == 14 || tile == 16 || tile == 17 || tile == 19 || tile == 20) tile = 4;      // scalar-gather layers (Cin % 4 != 0): only the basic tiles are instantiated
  if (tile == 12 || tile == 15 || tile == 18 || tile == 21) tile = 1;
  switch (tile) {
    case 0: { dim3 g(fsv_cdiv(M_tiles_rows, 128), fsv_cdiv(p.Cout, 128), nz);
      FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 2, V>), g, block, stream, p); break; }
    case 1: { dim3 g(fsv_cdiv(M_tiles_rows, 128), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 2, 2, V>), g, block, stream, p); break; }
    case 2: { dim3 g(fsv_cdiv(M_tiles_rows, 128), fsv_cdiv(p.Cout, 32), nz);
      FSV_LAUNCH((fsv_conv_igemm_kernel<128, 32, 4, 1, V>), g, block, stream, p); break; }
    case 3: { dim3 g(fsv_cdiv(M_tiles_rows, 256), fsv_cdiv(p.Cout, 32), nz);
      FSV_LAUNCH((fsv_conv_igemm_kernel<256, 32, 4, 1, V>), g, block, stream, p); break; }
    case 4: { dim3 g(fsv_cdiv(M_tiles_rows, 64), fsv_cdiv(p.Cout, 64), nz);
      FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2, V>), g, block, stream, p); break; }
    default: return FSV_ERR_BAD_ARG;
  }
  return fsv_check_launch();
}

static inline int fsv_tile_dims(int tile, int& bm, int& bn) {
  static const int BMs[22] = {128, 128, 128, 256, 64, 256, 128, 256, 256, 64, 64, 64, 128, 64, 64, 128, 64, 64, 128, 64, 64, 128},
                   BNs[22] = {128, 64, 32, 32, 64, 128, 256, 256, 64, 128, 64, 128, 64, 64, 128, 64, 64, 128, 64, 64, 128, 64};
  if (tile < 0 || tile > 21) return -1;
  bm = BMs[tile]; bn = BNs[tile];
  return 0;
}


#include <stdlib.h>
#include <string.h>
static inline long long fsv_tune(int which) {
  static long long vals[4] = {-1, -1, -1, -1};
  if (vals[0] < 0) {
    const char* a = getenv("FSV_SPLIT_BELOW");
    const char* b = getenv("FSV_SPLIT_TARGET");
    const char* c = getenv("FSV_WG_TARGET");
    const char* d = getenv("FSV_WG_MINCH");
    vals[1] = b ? atoll(b) : 512;
    vals[2] = c ? atoll(c) : 1024;
    vals[3] = d ? atoll(d) : 8;
    vals[0] = a ? atoll(a) : 256;
  }
  return vals[which];
}

static inline int fsv_thin_tile() {
  static int v = -1;
  if (v < 0) { const char* a = getenv("FSV_THIN_TILE"); v = a ? atoi(a) : 2; }   // 128x32: in-box A/B 82.2 vs 83.4 ms/step against 256x32
  return v;
}

// Tile / split-K plan shared by the launcher and (through the C ABI) by the host-side profiler labels.
// tile ids: 0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 256x32, 4 = 64x64 (BM x BN, pixels x output channels).
extern "C" int fsv_conv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split,
                             int* tile_out, int* nsplit_out) {
  int tile = force_tile;
  static int plan_v = -1;
  if (plan_v < 0) { const char* e = getenv("FSV_PLAN"); plan_v = e ? atoi(e) : 2; }
  const long long b0 = (long long)fsv_cdiv(Mz, 128) * fsv_cdiv(Cout, 128) * nsamp;      // 128x128 tiles
  const long long b1 = (long long)fsv_cdiv(Mz, 128) * fsv_cdiv(Cout, 64) * nsamp;       // 128x64
  const long long b4 = (long long)fsv_cdiv(Mz, 64) * fsv_cdiv(Cout, 64) * nsamp;        // 64x64
  const long long b9 = (long long)fsv_cdiv(Mz, 64) * fsv_cdiv(Cout, 128) * nsamp;       // 64x128
  bool small_tile_regime = false, mid_tile_regime = false;
  if (tile < 0) {
    if (plan_v == 0) {
      if (Cout <= 32) tile = (Mz >= 256 * 256) ? fsv_thin_tile() : 2;
      else if (Cout <= 64) tile = 1;
      else tile = ((long long)Mz * Cout <= 64 * 64 * 64) ? 4 : 0;
    } else {
      // in-box A/B on the step's layer shapes (tools/tile_ab.py, profiles/r01_tile_ab.jsonl): with fewer than two
      // 128-row tiles per CU the 64x64 tile without split-K beats the big tile with split-K (no atomics, no zero-fill,
      // no separate bias pass) unless K is long enough (>= 4096) to amortise them
      if (Cout <= 32) tile = (Mz >= 256 * 256) ? fsv_thin_tile() : 2;
      else if (Cout <= 64) tile = (b1 < 512) ? 4 : 1;
      else if (plan_v == 1) {
        if (b0 >= 512) tile = 0;
        else if (nchunks >= 128) tile = (b0 < 64) ? 1 : 0;
        else { tile = 4; }
      } else {
        // plan 2 adds the 64x128 tile (profiles/r01_tile_ab.jsonl, second table): half the A-tile re-reads of 64x64
        if (b0 > 1024) tile = 0;
        else if (nchunks >= 128) tile = (b0 < 64) ? 1 : ((b0 <= 128 && nchunks >= 256) ? 0 : 9);
        else if (b0 >= 512) tile = (Cout <= 128) ? 9 : 0;
        else if (b9 >= 512) tile = 9;
        else if (b9 >= 256 && nchunks >= 72) tile = 9;
        else tile = 4;
      }
      small_tile_regime = (tile == 4);
      mid_tile_regime = (tile == 9);
    }
  }
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return -1;
  // split-K for launches that would leave most of the 256 CUs idle
  long long blocks = (long long)fsv_cdiv(Mz, bm) * fsv_cdiv(Cout, bn) * nsamp;
  int nsplit = 1;
  if (force_split > 0) nsplit = force_split;
  else if (small_tile_regime) {
    // 64x64 tiles: split only when even they leave CUs idle, and keep >= 16 chunks (512 K-elements) per split
    if (blocks < 512 && nchunks >= 32) {
      nsplit = (int)((1024 + blocks - 1) / blocks);
      if (nsplit > nchunks / 16) nsplit = nchunks / 16;
      if (nsplit < 1) nsplit = 1;
    }
  } else if (mid_tile_regime) {
    // 64x128 tiles: ~1024 workgroups, at least 18 chunks (576 K-elements) per split
    if (blocks < 768 && nchunks >= 36) {
      nsplit = (int)((1024 + blocks - 1) / blocks);
      if (nsplit > nchunks / 18) nsplit = nchunks / 18;
      if (nsplit < 1) nsplit = 1;
    }
  } else if (blocks < fsv_tune(0) && nchunks >= 8) {
    // aim at a few workgroups per CU; thresholds are tunables (FSV_SPLIT_BELOW / FSV_SPLIT_TARGET) for A/B runs
    nsplit = (int)((fsv_tune(1) + blocks - 1) / blocks);
    if (nsplit > nchunks / 4) nsplit = nchunks / 4;
    if (nsplit < 1) nsplit = 1;
  }
  (void)b4;
  if (nsplit > nchunks) nsplit = nchunks;
  *tile_out = tile; *nsplit_out = nsplit;
  return 0;
}

static inline bool vec4_ok(int cin) { return (cin & 3) == 0; }

// FSV_TILE_REMAP="4:13,9:14,1:15": run the launches the plan gives to tile a with tile b of the same shape instead (whole-step
// A/B of the experimental variants without touching the plan; split-K factors stay those of the planned tile)
static inline int fsv_tile_remap(int tile) {
  static int map[22];
  static int ready = 0;
  if (!ready) {
    for (int i = 0; i < 22; ++i) map[i] = i;
    const char* e = getenv("FSV_TILE_REMAP");
    while (e && *e) {
      int a = atoi(e);
      const char* c = strchr(e, ':');
      if (!c) break;
      int b = atoi(c + 1);
      int am, an, bm, bn;
      if (!fsv_tile_dims(a, am, an) && !fsv_tile_dims(b, bm, bn) && am == bm && an == bn) map[a] = b;
      e = strchr(c, ',');
      if (e) ++e;
    }
    ready = 1;
  }
  return (tile >= 0 && tile < 22) ? map[tile] : tile;
}

extern "C" {

// Generic gather-GEMM (see header comment and include/fsv2v.h: fsv_conv_gather_fwd).
int fsv_conv_gather_fwd(const float* in, const float* wt, const float* bias, const float* res, float* out,
                        int N, int H, int W, int Cin, int OH, int OW, int Cout,
                        int ntaps, const int* ty, const int* tx, int sy, int sx,
                        int outH, int outW, int osy, int osx, int ooy, int oox,
                        int ldw, long long w_bstride, long long b_bstride, int per_sample,
                        int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                        hipStream_t stream) {
  if (!in || !wt || !out || ntaps < 1 || ntaps > 16 || N < 1 || Cin < 1 || Cout < 1) return FSV_ERR_BAD_ARG;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((ldw & 3) != 0 || ldw < Cout) return FSV_ERR_BAD_ARG;
  ConvP p;
  p.in = in; p.wt = wt; p.bias = bias; p.res = res; p.out = out; p.wscale = wscale;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.nchunks = fsv_cdiv(p.K, FSV_BK); p.ldw = ldw;
  p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  fsv_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi, 8);
  p.outH = outH; p.outW = outW; p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
  p.dense_out = (osy == 1 && osx == 1 && ooy == 0 && oox == 0 && outH == OH && outW == OW) ? 1 : 0;
  p.w_bstride = w_bstride; p.b_bstride = b_bstride; p.per_sample = per_sample ? 1 : 0;
  p.act = act; p.scale = scale;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  const int nsamp = per_sample ? N : 1;
  int tile = 0, nsplit = 1;
  if (fsv_conv_plan(p.Mz, Cout, p.nchunks, nsamp, force_tile, force_split, &tile, &nsplit)) return FSV_ERR_BAD_ARG;
  p.nsplit = nsplit;
  if (force_tile < 0) tile = fsv_tile_remap(tile);
  const long long total = (long long)N * outH * outW * Cout;
  // accumulate != 0: `out` was zeroed by the caller and partial results are added atomically (used by the
  // four parity-class launches of a stride-2 data gradient); bias/act/res are not applied in that mode.
  if (accumulate) {
    if (bias || res || act != FSV_ACT_NONE || scale != 1.f) return FSV_ERR_BAD_ARG;
    // nsplit == 1: every output pixel belongs to exactly one parity class and one tile -> plain stores
  } else if (nsplit > 1) {
    if (!p.dense_out) return FSV_ERR_UNSUPPORTED;
    (void)hipMemsetAsync(out, 0, (size_t)total * sizeof(float), stream);
  }
  const bool vec4 = (Cin % 4 == 0);
  int rc = vec4 ? fsv_launch_conv<4>(p, p.Mz, nsamp * nsplit, stream, tile)
                : fsv_launch_conv<1>(p, p.Mz, nsamp * nsplit, stream, tile);
  if (rc) return rc;
  if (!accumulate && nsplit > 1 && (bias || res || act != FSV_ACT_NONE || scale != 1.f)) {
    int grid = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    FSV_LAUNCH(fsv_bias_act_kernel, dim3(grid), dim3(256), stream, out, bias, res, total, Cout,
               (long long)outH * outW, per_sample ? b_bstride : 0ll, act, scale);
    rc = fsv_check_launch();
  }
  return rc;
}

// In-place x = act(x + bias[c]) over an NHWC tensor of `total` elements (the split-K finishing pass, exposed for operators
// that accumulate several launches into one output: > 16-tap convolutions, transposed convolutions)
int fsv_bias_act(float* x, const float* bias, long long total, int C, int act, hipStream_t stream) {
  if (!x || total < 1 || C < 1) return FSV_ERR_BAD_ARG;
  int grid = (int)((total + 256 * 8 - 1) / (256 * 8));
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  FSV_LAUNCH(fsv_bias_act_kernel, dim3(grid), dim3(256), stream, x, bias, (const float*)nullptr, total, C, 1ll, 0ll, act,
             1.f);
  return fsv_check_launch();
}

int fsv_conv_wgrad(const float* in, const float* dout, float* dwt,
                   int N, int H, int W, int Cin, int OH, int OW, int Cout,
                   int ntaps, const int* ty, const int* tx, int sy, int sx,
                   int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                   int force_tile, hipStream_t stream) {
  if (!in || !dout || !dwt || ntaps < 1 || ntaps > 16) return FSV_ERR_BAD_ARG;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  WgradP p;
  p.in = in; p.dout = dout; p.dwt = dwt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.ldw = ldw; p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  fsv_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi, 8);
  p.w_bstride = w_bstride; p.per_sample = per_sample ? 1 : 0;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  p.pchunks = fsv_cdiv(p.Mz, FSV_BK);
  const int nsamp = per_sample ? N : 1;
  int bn = (Cout <= 32) ? 32 : (Cout <= 64 ? 64 : 128);
  // rows of the weight-gradient tile = taps * Cin; the 1x1 SPADE / embedding layers have only 32 or 64 of them and
  // would waste 3/4 or 1/2 of a 128-row tile's MFMA work
  int bmk = 128;
  if (vec4_ok(Cin) && bn >= 64) bmk = (p.K <= 32) ? 32 : (p.K <= 64 ? 64 : 128);
  // force_tile (A/B runs, tools/wgrad_ab.py): 1 = 64x64, 2 = 128x64, 3 = 64x128 (rows x columns), vec4 layers only
  if (force_tile == 1 && vec4_ok(Cin) && Cout > 32) { bmk = 64; bn = 64; }
  else if (force_tile == 2 && vec4_ok(Cin) && Cout > 32) { bmk = 128; bn = 64; }
  else if (force_tile == 3 && vec4_ok(Cin) && Cout > 64) { bmk = 64; bn = 128; }
  // 5 / 6: the 64x64 and 64x128 tiles as one- / two-wave workgroups (every wave owns a 64x64 sub-tile); not yet measured
  int few_waves = 0;
  if (force_tile == 5 && vec4_ok(Cin) && Cout > 32) { bmk = 64; bn = 64; few_waves = 1; }
  else if (force_tile == 6 && vec4_ok(Cin) && Cout > 64) { bmk = 64; bn = 128; few_waves = 1; }
  // 7 / 8: 64x64 / 64x128 with double-buffered LDS (conv_igemm_db.hip); not yet measured
  int dbuf = 0;
  if (force_tile == 7 && vec4_ok(Cin) && Cout > 32) { bmk = 64; bn = 64; dbuf = 1; }
  else if (force_tile == 8 && vec4_ok(Cin) && Cout > 64) { bmk = 64; bn = 128; dbuf = 1; }
  long long target = fsv_tune(2);
  static int wplan_v = -1;
  if (wplan_v < 0) { const char* e = getenv("FSV_WGRAD_PLAN"); wplan_v = e ? atoi(e) : 1; }
  if (force_tile == 0 && wplan_v == 1 && vec4_ok(Cin) && Cout >= 64 && p.K > 64) {
    // in-box A/B on the step's layer shapes (tools/wgrad_ab.py, profiles/r01_wgrad_ab.jsonl): 64-row tiles with ~2048
    // (64x64) / ~1024 (64x128) workgroups beat 128-row tiles at ~1024 by 11 ... 23 % - four times fewer atomic adds per
    // FLOP than the same workgroup count of split 128x128 tiles, and every CU gets several workgroups
    if (Cout >= 128 && p.K >= 2304 && p.pchunks >= 64) { bmk = 64; bn = 128; target = 1024; }
    else { bmk = 64; bn = 64; target = 2048; }
  }
  // FSV_WGRAD_VARIANT=db | fw: run the automatically chosen 64-row tiles as their double-buffered / few-wave variants
  if (force_tile == 0 && vec4_ok(Cin) && bmk == 64 && (bn == 64 || bn == 128)) {
    static int variant = -1;
    if (variant < 0) { const char* e = getenv("FSV_WGRAD_VARIANT"); variant = !e ? 0 : (!strcmp(e, "db") ? 1 : (!strcmp(e, "fw") ? 2 : 0)); }
    if (variant == 1) dbuf = 1; else if (variant == 2) few_waves = 1;
  }
  long long blocks = (long long)fsv_cdiv(p.K, bmk) * fsv_cdiv(Cout, bn) * nsamp;
  int nsplit = 1;
  if (force_split > 0) nsplit = force_split;
  else {
    nsplit = (int)((target + blocks - 1) / blocks);
    int maxs = p.pchunks / (int)fsv_tune(3);       // keep at least this many 32-pixel chunks per split
    if (nsplit > maxs) nsplit = maxs;
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > p.pchunks) nsplit = p.pchunks;
  p.nsplit = nsplit;
  // split reductions add into dwt atomically and need it zeroed; a single split stores every (k < K, co < Cout)
  // entry directly and the padding rows / columns are never read back (fsv_prep_weight mode 2 skips them)
  if (nsplit > 1 && !prezeroed)
    (void)hipMemsetAsync(dwt, 0, (size_t)((per_sample ? (long long)N * w_bstride : (long long)Kpad * ldw)) * sizeof(float), stream);
  dim3 block(256);
  const bool vec4 = (Cin % 4 == 0);
  dim3 g(fsv_cdiv(p.K, bmk), fsv_cdiv(Cout, bn), nsamp * nsplit);
  if (vec4 && dbuf) return fsv_launch_wgrad_db(p, bmk, bn, g, stream);
  if (vec4) {
    if (few_waves && bn == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 64, 1, 1, 4>), g, dim3(64), stream, p);
    else if (few_waves) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 128, 1, 2, 4>), g, dim3(128), stream, p);
    else if (bn == 128 && bmk == 32) FSV_LAUNCH((fsv_conv_wgrad_kernel<32, 128, 1, 4, 4>), g, block, stream, p);
    else if (bn == 128 && bmk == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 128, 2, 2, 4>), g, block, stream, p);
    else if (bn == 64 && bmk == 32) FSV_LAUNCH((fsv_conv_wgrad_kernel<32, 64, 1, 2, 4>), g, dim3(128), stream, p);
    else if (bn == 64 && bmk == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 64, 2, 2, 4>), g, block, stream, p);
    else if (bn == 128) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 128, 2, 2, 4>), g, block, stream, p);
    else if (bn == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 64, 2, 2, 4>), g, block, stream, p);
    else FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 32, 4, 1, 4>), g, block, stream, p);
  } else {
    if (bn == 128) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 128, 2, 2, 1>), g, block, stream, p);
    else if (bn == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 64, 2, 2, 1>), g, block, stream, p);
    else FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 32, 4, 1, 1>), g, block, stream, p);
  }
  return fsv_check_launch();
}

int fsv_prep_weight(const float* w, float* wt, const float* scale_ptr, int mode, int nbatch,
                    int Cout, int Cin, int KH, int KW, int ntaps, const int* kh, const int* kw,
                    int Kpad, int ldw, long long w_bstride, long long wt_bstride, hipStream_t stream) {
  if (!w || !wt || ntaps < 1 || ntaps > 16 || mode < 0 || mode > 3) return FSV_ERR_BAD_ARG;
  unsigned long long lo, hi;
  fsv_pack_taps(kh, kw, ntaps, lo, hi, 0);
  long long total = (long long)Kpad * ldw;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  FSV_LAUNCH(fsv_prep_weight_kernel, dim3(grid, 1, nbatch), dim3(256), stream, w, wt, scale_ptr, mode, Cout, Cin,
             KH, KW, ntaps, lo, hi, Kpad, ldw, w_bstride, wt_bstride);
  return fsv_check_launch();
}

}  // extern "C"
