// Half-precision implicit-GEMM convolution for gfx950: the `--amp O1` arithmetic of the reference (models/models.py:22-26,
// options/base_options.py:127, loss_collector.py:221-224; BASELINE.json configs[4] "fp16 MFMA path") with the activations and
// the weights 16-BIT IN HBM and the gather-GEMM loop of csrc/conv_igemm.hip's LD form (round 3) underneath.
//
//   out[z][m][co] = sum_{t < ntaps} sum_{ci < Cin}  in[n, oy*sy + ty[t], ox*sx + tx[t], ci] * wt[z][co][t*Cin + ci]
//
// What differs from the fp32 kernel, and why:
//   * operands: IEEE half in HBM, v_mfma_f32_32x32x16_f16 (fp32 accumulate, 16x the fp32 matrix rate).  The round-2 narrow
//     kernels (csrc/conv_np.hip) kept fp32 tensors in HBM and narrowed while staging - they fetched 4 bytes per operand value
//     through a loop that waited for its loads at the top of every chunk and ran at 0.026 of the f16 peak.
//   * weights N-MAJOR: wt[co][Kpad] with K contiguous (fsv_hconv_prep_weight writes it from the K-major fp32 layout), because
//     an MFMA B fragment is 8 consecutive k of one output channel - the same shape as an A fragment (8 consecutive k of one
//     pixel).  Both tiles are therefore [rows][64 k] = 128-byte rows of 8 sixteen-byte slots, exactly the LDS image of the fp32
//     kernel's A tile: slot q of row r sits in slot q ^ ((r >> 1) & 7), fragments are ONE ds_read_b128 each, conflict free under
//     the instruction's 16-lane service groups (MI355X_MICROARCH.md, LDS).
//   * both tiles are written by the memory pipe (buffer_load_dwordx4 ... lds: no staging registers, no ds_write - the LDS store
//     path, 64 - 85 B/clk/CU, would be the bottleneck at this MFMA rate); the slot swizzle sits on the global side.  NBUF LDS
//     buffers, loads NBUF - 1 chunks ahead, one barrier per 64-wide chunk, counted vmcnt in front of it.
//   * output: half or fp32 (runtime flag); statistics for the normalisation that follows from the epilogue like the fp32 kernel.
// Conv semantics (the definition oracle/np_oracle.py restates): operands rounded to half BEFORE the call (they are half in HBM),
// products exact, fp32 accumulation, (acc / sigma + bias) * scale -> activation -> + residual in fp32, ONE rounding to half at the
// store when the output is half.
//
// The weight gradient reduces over pixels, so both of its operands arrive with the reduction index as the slow HBM index and are
// transposed on the way into LDS through registers (packed pixel pairs, 4-byte transposing stores: conflict free); its LDS-write
// bound loop is described at the kernel.
#include <stdlib.h>
#include "conv_igemm.h"

typedef _Float16 fsv_h16;
typedef _Float16 fsv_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fsv_h16x2 __attribute__((ext_vector_type(2)));

#define FSV_HBK 64            // k per chunk (halves): 128-byte LDS rows

struct HConvP {
  const fsv_h16* in;       // NHWC half
  const fsv_h16* wt;       // [z][nrows][Kpad] half, K contiguous
  const float* bias;
  const void* res;         // half (res_h) or fp32
  const float* wscale;     // optional device scalar multiplying the accumulator (spectral-norm 1/sigma)
  void* out;               // half (out_h) or fp32
  int N, H, W, Cin;
  int OH, OW, Cout;
  int K, nchunks, Kpad, nrows;      // K = ntaps*Cin; nchunks = ceil(K/64); weight rows of Kpad halves, nrows of them
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;
  int outH, outW, osy, osx, ooy, oox, dense_out;
  long long w_bstride, b_bstride;   // per-sample weight (halves) / bias strides
  int per_sample, nsplit;
  int act; float scale;
  int Mz;
  int out_h, res_h;
  long long res_bytes;     // extent of the residual tensor (0 without one)
  double* stats;
  int stats_slots, stats_ohw;
};

__device__ __forceinline__ void fsv_htap(const HConvP& p, int t, int& ty, int& tx) {
  unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
  int sh = (t & 7) * 8;
  ty = (int)((code >> sh) & 15ull) - 8;
  tx = (int)((code >> (sh + 4)) & 15ull) - 8;
}

// XCD bands (MI355X_MICROARCH.md "Workgroup dispatch": linear workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2).  The grid is
// 1-D and padded to 8 * per workgroups; XCD x owns the CONTIGUOUS run of tiles [x * per, (x + 1) * per) in (pixel tile, channel
// tile) order with the channel tile fastest: the channel tiles of one pixel tile share its gathered activations, and neighbouring
// pixel tiles - the rows above and below that the 3x3 / 4x4 taps reach into - are fetched into the SAME L2.  (The fp32 kernels
// give XCD x the pixel tiles x, x + 8, ...: at a quarter of this kernel's operand rate that was enough; here every L2 ended up
// holding the whole activation tensor and the LDS fill ran at L2-miss latency.)  Returns false for the padding workgroups.
__device__ __forceinline__ bool fsv_h_xcd_tile(int nx, int ny, int& bx, int& by) {
  const int total = nx * ny;
  const int per = (total + 7) >> 3;
  const int b = blockIdx.x;
  const int t = (b & 7) * per + (b >> 3);
  if ((b >> 3) >= per || t >= total) return false;
  by = t % ny;
  bx = t / ny;
  return true;
}

// LDS-direct 16-byte load of halves (the emulated form copies synchronously)
#ifdef FSV_EMU
static inline void fsv_hbuf_load_lds(const fsv_rawbuf& b, unsigned off, fsv_h16* lds_wave_base) {
  const float4 v = fsv_buf_load4(b, off);
  memcpy(lds_wave_base + (threadIdx.x & 63) * 8, &v, 16);
}
#else
__device__ __forceinline__ void fsv_hbuf_load_lds(fsv_rawbuf b, unsigned off, fsv_h16* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l), "v"(off), "s"(b) : "memory");
}
#endif

// Epilogue of the gather-GEMM kernels.  D layout: col = lane & 31 (channel), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (tile
// row).  row_m(tile row) -> linear pixel index m of that row (the output pixel follows from m as in the main loop), or -1 for rows
// outside the problem.  Statistics: rows with m < st_split belong to group st_g0, the others (straddles) to the next one.
template <int TM, int TN, class RowM>
__device__ __forceinline__ void fsv_hconv_epilogue(const HConvP& p, f32x16 (&acc)[TM][TN], const int zs, const int bn0, const int wm,
                                                   const int wn, const int lane, const int bx, const int st_g0, const int st_split,
                                                   const bool straddles, RowM row_m) {
  const int lrow = lane & 31, lk = lane >> 5;
  const int ohw = p.OH * p.OW;
  const float* bias = p.bias ? (p.bias + (long long)zs * p.b_bstride) : nullptr;
  const float ws = p.wscale ? p.wscale[0] : 1.f;
  fsv_h16* const out_h = reinterpret_cast<fsv_h16*>(p.out);
  float* const out_f = reinterpret_cast<float*>(p.out);
  // the residual / LeakyReLU-mask operand through a descriptor (the host checked that the output stays below 2^31 bytes when
  // there is one): all of a lane's values are loaded BEFORE its first store - the stores may alias p.res as far as the compiler
  // can tell, so a load inside the store loop waits for its full memory latency once per element (16 - 32 times per tile)
  const fsv_buf rbuf = fsv_make_buf(p.res, p.res ? p.res_bytes : 0);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = bn0 + wn * (TN * 32) + j * 32 + lrow;
    const bool cok = co < p.Cout;
    const float bv = (bias && p.nsplit == 1 && cok) ? bias[co] : 0.f;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    int oix[TM][16];          // element index of the output (the host checked < 2^31 elements), -1: not stored
    float aux[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = row_m(wm * (TM * 32) + i * 32 + row);          // linear pixel index of this tile row, or -1
        long long opix;
        if (p.dense_out) {
          opix = (long long)zs * (p.per_sample ? p.Mz : 0) + m;
        } else {
          int n, rem;
          if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
          const int oy = rem / p.OW, ox = rem - oy * p.OW;
          opix = ((long long)n * p.outH + (oy * p.osy + p.ooy)) * p.outW + (ox * p.osx + p.oox);
        }
        const bool ok = cok & (m >= 0);
        oix[i][r] = ok ? (int)(opix * p.Cout + co) : -1;
        aux[i][r] = 0.f;
      }
    }
    // Half tensors move two channels per work-item: a lane holds ONE channel of consecutive pixels, its neighbour (lane ^ 1) the
    // next channel of the same pixels - for a pair of rows (r, r + 1) the even lane handles channels (co, co + 1) of row r and the
    // odd lane (co - 1, co) of row r + 1, one exchange each way (Cout is a multiple of 8 here).  2-byte loads / stores run at half
    // the rate of the same launch moving twice the bytes as fp32 (csrc/spade.hip, round 4).
    const bool odd = (lane & 1) != 0;
    const int oddm = -(int)(lane & 1);          // bit mux between the two rows' indices: a select of two array elements would be
                                                // turned into a dynamically indexed load and send the array to scratch memory
    const bool pairs = (p.Cout & 1) == 0;       // uniform (an odd channel count: one element per access)
    if (p.res) {              // uniform
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (p.res_h && !pairs) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            aux[i][r] = fsv_buf_load_h(rbuf, oix[i][r] >= 0 ? (unsigned)oix[i][r] * 2u : FSV_BUF_OOB);
        } else if (p.res_h) {        // uniform
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int src = oix[i][r] ^ ((oix[i][r] ^ oix[i][r + 1]) & oddm);        // odd lanes: row r + 1
            const float w = fsv_buf_load1(rbuf, src >= 0 ? (unsigned)(src + oddm) * 2u : FSV_BUF_OOB);
            const float nw = __shfl_xor(w, 1);
            const fsv_h16x2 mine = __builtin_bit_cast(fsv_h16x2, w), theirs = __builtin_bit_cast(fsv_h16x2, nw);
            aux[i][r] = odd ? (float)theirs.y : (float)mine.x;
            aux[i][r + 1] = odd ? (float)mine.y : (float)theirs.x;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            aux[i][r] = fsv_buf_load1(rbuf, oix[i][r] >= 0 ? (unsigned)oix[i][r] * 4u : FSV_BUF_OOB);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = row_m(wm * (TM * 32) + i * 32 + row);
        const bool ok0 = oix[i][r] >= 0, ok1 = oix[i][r + 1] >= 0;
        float v0 = acc[i][j][r] * ws, v1 = acc[i][j][r + 1] * ws;
        if (p.nsplit > 1) {                      // uniform
          // split launches accumulate into a zeroed fp32 buffer (the host's workspace)
          if (ok0) atomicAdd(out_f + oix[i][r], v0);
          if (ok1) atomicAdd(out_f + oix[i][r + 1], v1);
        } else {
          v0 = (v0 + bv) * p.scale; v1 = (v1 + bv) * p.scale;
          if (p.act == FSV_ACT_DLRELU) {
            v0 = aux[i][r] > 0.f ? v0 : 0.2f * v0;
            v1 = aux[i][r + 1] > 0.f ? v1 : 0.2f * v1;
          } else {
            v0 = fsv_act(v0, p.act) + aux[i][r];
            v1 = fsv_act(v1, p.act) + aux[i][r + 1];
          }
          if (p.out_h) {                         // uniform
            const fsv_h16 h0 = (fsv_h16)v0, h1 = (fsv_h16)v1;
            v0 = (float)h0; v1 = (float)h1;      // the statistics are those of the stored tensor
            if (pairs) {
              const float n0 = __shfl_xor(v0, 1), n1 = __shfl_xor(v1, 1);
              fsv_h16x2 pk;
              pk.x = (fsv_h16)(odd ? n1 : v0); pk.y = (fsv_h16)(odd ? v1 : n0);
              const int src = oix[i][r] ^ ((oix[i][r] ^ oix[i][r + 1]) & oddm);      // odd lanes: row r + 1, one channel down
              if (src >= 0) *reinterpret_cast<fsv_h16x2*>(out_h + (src + oddm)) = pk;
            } else {
              if (ok0) out_h[oix[i][r]] = h0;
              if (ok1) out_h[oix[i][r + 1]] = h1;
            }
          } else {
            if (ok0) out_f[oix[i][r]] = v0;
            if (ok1) out_f[oix[i][r + 1]] = v1;
          }
          if (p.stats) {
            if (ok0) { if (m < st_split) { s0 += v0; q0 += v0 * v0; } else { s1 += v0; q1 += v0 * v0; } }
            if (ok1) { if (m + 1 < st_split) { s0 += v1; q0 += v1 * v1; } else { s1 += v1; q1 += v1 * v1; } }
          }
        }
      }
    }
    if (p.stats) {            // uniform
      s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32);
      s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
      if (lk == 0 && cok) {
        const int slot = bx % p.stats_slots;
        double* d = p.stats + (((long long)st_g0 * p.stats_slots + slot) * p.Cout + co) * 2;
        atomicAdd(d, (double)s0); atomicAdd(d + 1, (double)q0);
        if (straddles) {
          d += (long long)p.stats_slots * p.Cout * 2;
          atomicAdd(d, (double)s1); atomicAdd(d + 1, (double)q1);
        }
      }
    }
  }
}

// One output tile.  BM x BN pixels x channels, WM x WN waves, NBUF LDS buffers (2: loads one chunk ahead, 3: two chunks ahead).
// Whole trips of NBUF chunks: chunks at or past the end of this split's K range load zeros on both sides (out-of-range offsets)
// and are multiplied like the others - no exit inside a trip, buffers addressed statically.
template <int BM, int BN, int WM, int WN, int NBUF>
__device__ __forceinline__ void fsv_hconv_body(const HConvP& p, const int bx, const int by, const int bz) {
  constexpr int BK = FSV_HBK;
  constexpr int NT = 64 * WM * WN;
  constexpr int RPA = NT / 8;          // tile rows per load pass: 8 lanes x 16 B = one 128-byte row
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NPA = BM / RPA, NPB = BN / RPA;
  constexpr int A_ST = BM * BK, B_ST = BN * BK;      // halves per buffer
  static_assert(TM >= 1 && TN >= 1 && NPA >= 1 && NPB >= 1 && NPA * RPA == BM && NPB * RPA == BN, "tile / thread-count mismatch");
  static_assert(NBUF == 2 || NBUF == 3, "two or three LDS buffers");
  static_assert((RPA / 2) % 8 == 0, "rows of one thread must share a slot swizzle");
  __shared__ __attribute__((aligned(16))) fsv_h16 smem[NBUF * (A_ST + B_ST)];
  fsv_h16* const As = smem;
  fsv_h16* const Bs = smem + NBUF * A_ST;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bm0 = bx * BM, bn0 = by * BN;
  const fsv_h16* wt = p.wt + (long long)zs * p.w_bstride;

  // the lane's LDS slot is fixed by its lane id; the 8 k it fetches are the ones that belong there
  const int r0 = tid >> 3;
  const int ls = (tid & 7) ^ ((r0 >> 1) & 7);          // logical slot (8 k) of this thread's loads
  int a_iy0[NPA], a_ix0[NPA], a_pix[NPA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int m = bm0 + r0 + i * RPA;
    if (m < p.Mz) {
      int n, rem;
      if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.sy; a_ix0[i] = ox * p.sx;
      a_pix[i] = ((n * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.Cin * 2;      // byte offset of tap (0, 0), channel 0
    } else {
      a_iy0[i] = -(1 << 28); a_ix0[i] = 0; a_pix[i] = 0;
    }
  }
  const fsv_rawbuf araw = fsv_make_rawbuf(p.in, (long long)p.N * p.H * p.W * p.Cin * 2);
  const fsv_rawbuf braw = fsv_make_rawbuf(wt, (long long)p.nrows * p.Kpad * 2);

  const int cps = (p.nchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.nchunks) ? (c_begin + cps) : p.nchunks;

  unsigned aoff[NPA], boff[NPB];
  int cur_k = c_begin * BK + ls * 8;
  int cur_t = cur_k / p.Cin;
  int cur_ci = cur_k - cur_t * p.Cin;
  const int q64 = BK / p.Cin, r64 = BK - q64 * p.Cin;
  const int k_lim = c_end * BK < p.K ? c_end * BK : p.K;      // another split's share of K must not be multiplied here
  unsigned b_row[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int n = bn0 + r0 + i * RPA;
    b_row[i] = (n < p.nrows) ? (unsigned)(n * p.Kpad * 2) : FSV_BUF_OOB;
  }
  // byte offsets of one chunk's loads, computed one chunk ahead of their loads (branch-free: see csrc/conv_igemm.hip)
  auto calc_offsets = [&]() {
    const bool kok = cur_k < k_lim;
    int ty, tx;
    fsv_htap(p, cur_t, ty, tx);
    const int toff = ((ty * p.W + tx) * p.Cin + cur_ci) * 2;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
      const bool ok = kok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      aoff[i] = (unsigned)(a_pix[i] + toff) | (ok ? 0u : FSV_BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) boff[i] = (b_row[i] + (unsigned)(cur_k * 2)) | (kok ? 0u : FSV_BUF_OOB);
    cur_k += BK;
    cur_t += q64;
    cur_ci += r64;
    const bool wrap = cur_ci >= p.Cin;
    cur_ci = wrap ? cur_ci - p.Cin : cur_ci;
    cur_t = wrap ? cur_t + 1 : cur_t;
  };
  // a wave's 64 lanes cover 8 consecutive rows x 8 slots: exactly the 1 KB one load instruction writes
  auto issue_loads = [&](fsv_h16* a_buf, fsv_h16* b_buf) {
    fsv_h16* a_dst = a_buf + wave * (8 * BK);
    fsv_h16* b_dst = b_buf + wave * (8 * BK);
#pragma unroll
    for (int i = 0; i < NPA; ++i) fsv_hbuf_load_lds(araw, aoff[i], a_dst + i * (RPA * BK));
#pragma unroll
    for (int i = 0; i < NPB; ++i) fsv_hbuf_load_lds(braw, boff[i], b_dst + i * (RPA * BK));
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lk = lane >> 5;
  int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (TM * 32) + i * 32 + lrow;
    a_off[i] = r * BK; a_swz[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * (TN * 32) + j * 32 + lrow;
    b_off[j] = r * BK; b_swz[j] = (r >> 1) & 7;
  }
  // fragments of MFMA step s (16 k): lanes 0-31 the first 8 k, lanes 32-63 the second 8 - logical slot 2 s + lk of the row
  auto read_step = [&](const fsv_h16* a_src, const fsv_h16* b_src, int s, fsv_h16x8 (&fa)[TM], fsv_h16x8 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const fsv_h16x8*>(&a_src[a_off[i] + (((2 * s + lk) ^ a_swz[i]) << 3)]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const fsv_h16x8*>(&b_src[b_off[j] + (((2 * s + lk) ^ b_swz[j]) << 3)]);
  };
  auto mma_step = [&](const fsv_h16x8 (&fa)[TM], const fsv_h16x8 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  // one chunk out of (a_src, b_src) while the loads of a later chunk fill (a_dma, b_dma); fragment reads one step ahead of their
  // MFMAs, pinned with scheduling fences; before the barrier only the loads of the NEXT chunk are waited for
  auto chunk = [&](fsv_h16* a_dma, fsv_h16* b_dma, const fsv_h16* a_src, const fsv_h16* b_src) {
    issue_loads(a_dma, b_dma);
    fsv_h16x8 fa[2][TM], fb[2][TN];
    read_step(a_src, b_src, 0, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    calc_offsets();
    read_step(a_src, b_src, 1, fa[1], fb[1]);
    FSV_SCHED_FENCE();
    mma_step(fa[0], fb[0]);
    FSV_SCHED_FENCE();
    read_step(a_src, b_src, 2, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    mma_step(fa[1], fb[1]);
    FSV_SCHED_FENCE();
    read_step(a_src, b_src, 3, fa[1], fb[1]);
    FSV_SCHED_FENCE();
    mma_step(fa[0], fb[0]);
    FSV_SCHED_FENCE();
    mma_step(fa[1], fb[1]);
    FSV_WAIT_VMCNT((NBUF - 2) * (NPA + NPB));
    __syncthreads();
  };
  if (c_begin < c_end) {
    fsv_h16* const A0 = As, * const A1 = As + A_ST;
    fsv_h16* const B0 = Bs, * const B1 = Bs + B_ST;
    if constexpr (NBUF == 3) {
      fsv_h16* const A2 = As + 2 * A_ST;
      fsv_h16* const B2 = Bs + 2 * B_ST;
      calc_offsets();
      issue_loads(A0, B0);
      calc_offsets();
      issue_loads(A1, B1);
      calc_offsets();
      FSV_WAIT_VMCNT(NPA + NPB);
      __syncthreads();
#pragma unroll 1
      for (int kc = c_begin; kc < c_end; kc += 3) {
        chunk(A2, B2, A0, B0);
        chunk(A0, B0, A1, B1);
        chunk(A1, B1, A2, B2);
      }
    } else {
      calc_offsets();
      issue_loads(A0, B0);
      calc_offsets();
      FSV_WAIT_VMCNT(0);
      __syncthreads();
#pragma unroll 1
      for (int kc = c_begin; kc < c_end; kc += 2) {
        chunk(A1, B1, A0, B0);
        chunk(A0, B0, A1, B1);
      }
    }
  }
  FSV_WAIT_VMCNT(0);          // loads of chunks past the end are still landing in LDS: they must not outlive the workgroup's allocation

  // ---- epilogue ---------------------------------------------------------------------------------------------------------------
  const int st_g0 = p.stats ? bm0 / p.stats_ohw : 0;
  const int st_split = (st_g0 + 1) * (p.stats ? p.stats_ohw : 0);
  fsv_hconv_epilogue<TM, TN>(p, acc, zs, bn0, wm, wn, lane, bx, st_g0, st_split, bm0 + BM > st_split && st_split < p.Mz,
                             [&](int rt) { const int m = bm0 + rt; return m < p.Mz ? m : -1; });
}

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_hconv_kernel(HConvP p) {
  int bx, by;
  if (!fsv_h_xcd_tile((p.Mz + BM - 1) / BM, (p.Cout + BN - 1) / BN, bx, by)) return;
  fsv_hconv_body<BM, BN, WM, WN, NBUF>(p, bx, by, (int)blockIdx.z);
}

// (A patch-resident form of the stride-1 3x3 convolutions - the (TH + 2) x (TW + 2) input patch of a pixel tile in LDS once per 32
// channels, the taps as steps over it: 2 - 4x fewer bytes from L2 to LDS - was built and measured in round 4 (profiles/r04_notes.md
// section 2, r04_h_ab_patch.jsonl): correct on hardware and SLOWER on every layer shape (372 against 625 TFLOP/s on M32768 N128 K2304;
// street --amp step 22.03 against 21.64 ms).  The fill volume does not bound the gather form; the kernel was removed in round 5.)

// grouped launch (see fsv_conv_igemm_group_kernel): up to FSV_GROUP_MAX independent problems in one 1-D grid
struct HConvGroup {
  int nprob;
  int tile_end[FSV_GROUP_MAX];
  HConvP p[FSV_GROUP_MAX];
};

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(64 * WM * WN) void fsv_hconv_group_kernel(HConvGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.nprob && b >= g.tile_end[i]) ++i;
  const int t = b - (i ? g.tile_end[i - 1] : 0);
  const HConvP& p = g.p[i];
  const int gx = (p.Mz + BM - 1) / BM, gy = (p.Cout + BN - 1) / BN;
  const int r = t / gx;
  fsv_hconv_body<BM, BN, WM, WN, NBUF>(p, t - r * gx, r % gy, r / gy);
}

// ---- finishing pass of split-K launches: out = act((ws + bias) * scale) + res, out half or fp32 ----------------------------------
__global__ __launch_bounds__(256) void fsv_hconv_finish_kernel(const float* ws, void* out, const float* bias, const void* res,
                                                               long long total, int C, long long pix_per_sample,
                                                               long long b_bstride, int act, float scale, int out_h, int res_h) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    float v = ws[i];
    if (bias) {
      const long long n = b_bstride ? pix / pix_per_sample : 0;
      v += bias[n * b_bstride + c];
    }
    v *= scale;
    if (act == FSV_ACT_DLRELU) {
      const float aux = res_h ? (float)reinterpret_cast<const fsv_h16*>(res)[i] : reinterpret_cast<const float*>(res)[i];
      v = aux > 0.f ? v : 0.2f * v;
    } else {
      v = fsv_act(v, act);
      if (res) v += res_h ? (float)reinterpret_cast<const fsv_h16*>(res)[i] : reinterpret_cast<const float*>(res)[i];
    }
    if (out_h) reinterpret_cast<fsv_h16*>(out)[i] = (fsv_h16)v; else reinterpret_cast<float*>(out)[i] = v;
  }
}

// ---- weight gradient: dwt[z][t*Cin+ci][co] (+)= sum_pixels in[n, oy*sy+ty, ox*sx+tx, ci] * dout[n,oy,ox,co] (fp32 out) ------
// Both operands are pixel-major in HBM (channels contiguous) and the MFMA wants 8 consecutive PIXELS of one channel per lane, so both
// tiles are transposed on the way into LDS through registers: a work-item owns 8 channels of two consecutive pixels (two 16-byte
// loads), packs the eight (pixel, pixel + 1) pairs and writes them as 4-byte stores into the [channel][64 pixels] images - the 32
// pixel pairs of a chunk sit on consecutive lanes, so a store instruction covers the 32 dwords of one 128-byte row (conflict free)
// and the slot swizzle of the forward kernel keeps the b128 fragment reads conflict free as well.  Two LDS buffers, loads one
// chunk ahead in registers (issued at the top of a chunk, packed and stored behind its MFMAs), one barrier per 64-pixel chunk.
// The loop is bound by the LDS store path (32 KB of 4-byte stores per 128x128 chunk at 64 B/clk against 512 cycles of MFMA);
// the loads are 16 bytes per lane at a pixel stride (each lane its own line: four times the tag work of a coalesced load).
struct HWgradP {
  const fsv_h16* in;
  const fsv_h16* dout;
  float* dwt;              // [K_pad][ldw] fp32 per z-sample (the layout grad_finalize.GradFinalizer consumes)
  int N, H, W, Cin;
  int OH, OW, Cout;
  int K, ldw;
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;
  long long w_bstride;
  int per_sample, nsplit;
  int Mz;
  int pchunks;             // ceil(Mz/64)
};

__device__ __forceinline__ void fsv_h_xcd_range(int& kt, int& nt, int& z) {
  const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const int G = gx * gy;
  const int b = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  int w;
  if (b < (gz >> 3) * 8 * G) {
    const int xcd = b & 7, j = b >> 3;
    z = xcd + 8 * (j / G);
    w = j % G;
  } else {
    z = b / G;
    w = b - z * G;
  }
  kt = w % gx;
  nt = w / gx;
}

template <int BMK, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fsv_hconv_wgrad_kernel(HWgradP p) {
  constexpr int BP = FSV_HBK;          // pixels per chunk
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BMK / (WM * 32), TN = BN / (WN * 32);
  constexpr int PP = BP / 2;           // pixel pairs per chunk: the fast work-item index
  constexpr int OPP = NT / PP;         // channel octets per pass
  constexpr int NPA = BMK / (8 * OPP), NPB = (BN + 8 * OPP - 1) / (8 * OPP);
  constexpr int A_ST = BMK * BP, B_ST = BN * BP;
  // (a B tile narrower than one pass - 128x32 - is loaded and stored by the work-items whose octet lies inside it: whole waves)
  static_assert(TM >= 1 && TN >= 1 && NPA >= 1 && NPA * 8 * OPP == BMK && (NPB * 8 * OPP == BN || (NPB == 1 && BN % 8 == 0)), "tile / thread-count mismatch");
  __shared__ __attribute__((aligned(16))) fsv_h16 smem[2 * (A_ST + B_ST)];
  fsv_h16* const As = smem;
  fsv_h16* const Bs = smem + 2 * A_ST;
  int kt, nt, bz;
  fsv_h_xcd_range(kt, nt, bz);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bi0 = kt * BMK, bn0 = nt * BN;
  float* dwt = p.dwt + (long long)zs * p.w_bstride;
  const int ohw = p.OH * p.OW;

  const int pp = tid % PP, o0 = tid / PP;          // pixel pair of the chunk, first channel octet
  // A: the (tap, channel octet)s of this work-item are fixed for the whole reduction
  bool kok[NPA];
  int a_ci[NPA], a_ty[NPA], a_tx[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int kcol = bi0 + (o0 + i * OPP) * 8;
    kok[i] = kcol < p.K;
    const int t = kok[i] ? kcol / p.Cin : 0;
    a_ci[i] = kok[i] ? (kcol - t * p.Cin) : 0;
    unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
    const int sh = (t & 7) * 8;
    a_ty[i] = (int)((code >> sh) & 15ull) - 8;
    a_tx[i] = (int)((code >> (sh + 4)) & 15ull) - 8;
  }
  const long long dout_base = (long long)zs * (p.per_sample ? p.Mz : 0);
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * p.H * p.W * p.Cin * 2);
  const fsv_buf bbuf = fsv_make_buf(p.dout + dout_base * p.Cout, (long long)p.Mz * p.Cout * 2);

  const int cps = (p.pchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.pchunks) ? (c_begin + cps) : p.pchunks;

  float4 areg[NPA][2], breg[NPB][2];
  // the two pixels of this work-item's pair: (n, oy, ox) advanced by 64 pixels per chunk with one conditional subtract per level
  // (the host sends geometries with 64 / OW + 1 > OH to the fp32 path)
  int m_pix[2], p_n[2], p_oy[2], p_ox[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int m = c_begin * BP + 2 * pp + h;
    int n, rem;
    if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
    m_pix[h] = m; p_n[h] = n; p_oy[h] = rem / p.OW; p_ox[h] = rem - p_oy[h] * p.OW;
  }
  const int qw = BP / p.OW, rw = BP - qw * p.OW;
  auto load_chunk = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool mok = m_pix[h] < p.Mz;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int iy = p_oy[h] * p.sy + a_ty[i], ix = p_ox[h] * p.sx + a_tx[i];
        const bool ok = mok & kok[i] & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        areg[i][h] = fsv_buf_load4(abuf, ok ? (unsigned)((((p_n[h] * p.H + iy) * p.W + ix) * p.Cin + a_ci[i]) * 2) : FSV_BUF_OOB);
      }
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const int bcol = bn0 + (o0 + i * OPP) * 8;
        breg[i][h] = fsv_buf_load4(bbuf, (mok & (bcol < p.Cout) & ((o0 + i * OPP) * 8 < BN)) ? (unsigned)((m_pix[h] * p.Cout + bcol) * 2) : FSV_BUF_OOB);
      }
      m_pix[h] += BP;
      int ox = p_ox[h] + rw, oy = p_oy[h] + qw;
      const bool cx = ox >= p.OW;
      ox = cx ? ox - p.OW : ox;
      oy = cx ? oy + 1 : oy;
      const bool cy = oy >= p.OH;
      p_oy[h] = cy ? oy - p.OH : oy;
      p_n[h] = cy ? p_n[h] + 1 : p_n[h];
      p_ox[h] = ox;
    }
  };
  // (v0, v1) = 8 channels of two consecutive pixels -> eight packed (pixel, pixel + 1) pairs into rows row0 .. row0 + 7
  auto store_pairs = [&](fsv_h16* img, const float4& v0, const float4& v1, int row0) {
    const unsigned a[4] = {__builtin_bit_cast(unsigned, v0.x), __builtin_bit_cast(unsigned, v0.y), __builtin_bit_cast(unsigned, v0.z),
                           __builtin_bit_cast(unsigned, v0.w)};
    const unsigned b[4] = {__builtin_bit_cast(unsigned, v1.x), __builtin_bit_cast(unsigned, v1.y), __builtin_bit_cast(unsigned, v1.z),
                           __builtin_bit_cast(unsigned, v1.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned lo = (a[e] & 0xffffu) | (b[e] << 16);         // channel 2e:     (pixel, pixel + 1)
      const unsigned hi = (a[e] >> 16) | (b[e] & 0xffff0000u);     // channel 2e + 1
      const int ra = row0 + 2 * e, rb = ra + 1;
      // column 2 pp of the 64-pixel row: slot (pp >> 2) swizzled by the row, dword (pp & 3) inside it
      *reinterpret_cast<unsigned*>(&img[ra * BP + ((((pp >> 2) ^ (ra >> 1)) & 7) << 3) + ((pp & 3) << 1)]) = lo;
      *reinterpret_cast<unsigned*>(&img[rb * BP + ((((pp >> 2) ^ (rb >> 1)) & 7) << 3) + ((pp & 3) << 1)]) = hi;
    }
  };
  auto store_chunk = [&](int buf) {
    fsv_h16* a_dst = As + buf * A_ST;
    fsv_h16* b_dst = Bs + buf * B_ST;
#pragma unroll
    for (int i = 0; i < NPA; ++i) store_pairs(a_dst, areg[i][0], areg[i][1], (o0 + i * OPP) * 8);
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      if ((o0 + i * OPP) * 8 < BN) store_pairs(b_dst, breg[i][0], breg[i][1], (o0 + i * OPP) * 8);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lk = lane >> 5;
  int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (TM * 32) + i * 32 + lrow;
    a_off[i] = r * BP; a_swz[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * (TN * 32) + j * 32 + lrow;
    b_off[j] = r * BP; b_swz[j] = (r >> 1) & 7;
  }
  auto read_step = [&](const fsv_h16* a_src, const fsv_h16* b_src, int s, fsv_h16x8 (&fa)[TM], fsv_h16x8 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const fsv_h16x8*>(&a_src[a_off[i] + (((2 * s + lk) ^ a_swz[i]) << 3)]);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const fsv_h16x8*>(&b_src[b_off[j] + (((2 * s + lk) ^ b_swz[j]) << 3)]);
  };
  auto mma_step = [&](const fsv_h16x8 (&fa)[TM], const fsv_h16x8 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  if (c_begin < c_end) {
    load_chunk();
    store_chunk(0);
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int pc = c_begin; pc < c_end; ++pc) {
      // the next chunk's loads first; they land in registers under this chunk's MFMAs and are stored into the OTHER buffer (nobody
      // reads it during this iteration); loads past the end of this split's range read another split's pixels or zeros and what
      // the last iteration stores is never used
      load_chunk();
      const fsv_h16* a_src = As + buf * A_ST;
      const fsv_h16* b_src = Bs + buf * B_ST;
      fsv_h16x8 fa[2][TM], fb[2][TN];
      read_step(a_src, b_src, 0, fa[0], fb[0]);
      FSV_SCHED_FENCE();
      read_step(a_src, b_src, 1, fa[1], fb[1]);
      FSV_SCHED_FENCE();
      mma_step(fa[0], fb[0]);
      FSV_SCHED_FENCE();
      read_step(a_src, b_src, 2, fa[0], fb[0]);
      FSV_SCHED_FENCE();
      mma_step(fa[1], fb[1]);
      FSV_SCHED_FENCE();
      read_step(a_src, b_src, 3, fa[1], fb[1]);
      FSV_SCHED_FENCE();
      mma_step(fa[0], fb[0]);
      FSV_SCHED_FENCE();
      store_chunk(buf ^ 1);
      FSV_SCHED_FENCE();
      mma_step(fa[1], fb[1]);
      __syncthreads();
      buf ^= 1;
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

// ---- operand preparation --------------------------------------------------------------------------------------------------------
// K-major fp32 weights wt[z][Kpad32][ldw] (fsv_prep_weight / layout_cache: forward and data-gradient layouts alike) -> N-major half
// wh[z][nrows][Kpad64] = half(s * wt[k][n]), zero beyond (K32 rows, ldw columns).  Table-driven so that an optimiser's whole layout
// cache is converted by ONE launch right behind fsv_prep_weight_grouped: jobs[j] = {src, dst, Kpad32, ldw, nrows, Kpad64, nbatch,
// reserved} as 64-bit words, tmap[b] = (job, k tile of 64, n tile of 64, z).
__device__ __forceinline__ void fsv_hconv_prep_tile(const float* src, fsv_h16* dst, int K32, int ldw, int nrows, int K64, int kt,
                                                    int ntile, float (*t)[65]) {
  const int tid = threadIdx.x;
  const int k0 = kt * 64, n0 = ntile * 64;
  // read [64 k][64 n] with n fastest (coalesced rows of the K-major source)
  for (int e = tid; e < 64 * 64; e += 256) {
    const int kk = e >> 6, nn = e & 63;
    const int k = k0 + kk, n = n0 + nn;
    t[kk][nn] = (k < K32 && n < ldw) ? src[(long long)k * ldw + n] : 0.f;
  }
  __syncthreads();
  // write [64 n][64 k] with k fastest, two halves per work-item
  for (int e = tid; e < 64 * 32; e += 256) {
    const int nn = e >> 5, kp = (e & 31) * 2;
    const int n = n0 + nn, k = k0 + kp;
    if (n < nrows && k < K64) {
      fsv_h16x2 v;
      v[0] = (fsv_h16)t[kp][nn];
      v[1] = (fsv_h16)t[kp + 1][nn];
      *reinterpret_cast<fsv_h16x2*>(&dst[(long long)n * K64 + k]) = v;
    }
  }
}

__global__ __launch_bounds__(256) void fsv_hconv_prep_kernel(const long long* jobs, const int* tmap) {
  __shared__ float t[64][65];
  const int j = tmap[blockIdx.x * 4], kt = tmap[blockIdx.x * 4 + 1], ntile = tmap[blockIdx.x * 4 + 2], z = tmap[blockIdx.x * 4 + 3];
  const long long* jb = jobs + (long long)j * 8;
  const int K32 = (int)jb[2], ldw = (int)jb[3], nrows = (int)jb[4], K64 = (int)jb[5];
  fsv_hconv_prep_tile(reinterpret_cast<const float*>(jb[0]) + (long long)z * K32 * ldw,
                      reinterpret_cast<fsv_h16*>(jb[1]) + (long long)z * nrows * K64, K32, ldw, nrows, K64, kt, ntile, t);
}

// one layout, geometry in the kernel arguments (no device table: legal inside a graph capture - per-call operands of generated
// weights); grid (k tiles of 64, n tiles of 64, batch)
__global__ __launch_bounds__(256) void fsv_hconv_prep_one_kernel(const float* src, fsv_h16* dst, int K32, int ldw, int nrows, int K64) {
  __shared__ float t[64][65];
  const int z = blockIdx.z;
  fsv_hconv_prep_tile(src + (long long)z * K32 * ldw, dst + (long long)z * nrows * K64, K32, ldw, nrows, K64, (int)blockIdx.x,
                      (int)blockIdx.y, t);
}

// fp32 -> half / half -> fp32 element conversion of a dense tensor (n % 4 == 0 handled as float4 / 8-byte vectors, tail per element)
__global__ __launch_bounds__(256) void fsv_cast_f2h_kernel(const float* x, fsv_h16* y, long long n) {
  const long long n4 = n >> 2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (long long q = i; q < n4; q += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[q];
    fsv_h16x2 a, b;
    a[0] = (fsv_h16)v.x; a[1] = (fsv_h16)v.y; b[0] = (fsv_h16)v.z; b[1] = (fsv_h16)v.w;
    reinterpret_cast<fsv_h16x2*>(y)[2 * q] = a;
    reinterpret_cast<fsv_h16x2*>(y)[2 * q + 1] = b;
  }
  for (long long e = (n4 << 2) + i; e < n; e += stride) y[e] = (fsv_h16)x[e];
}
__global__ __launch_bounds__(256) void fsv_cast_h2f_kernel(const fsv_h16* x, float* y, long long n) {
  const long long n4 = n >> 2;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (long long q = i; q < n4; q += stride) {
    const fsv_h16x2 a = reinterpret_cast<const fsv_h16x2*>(x)[2 * q], b = reinterpret_cast<const fsv_h16x2*>(x)[2 * q + 1];
    reinterpret_cast<float4*>(y)[q] = make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
  }
  for (long long e = (n4 << 2) + i; e < n; e += stride) y[e] = (float)x[e];
}

// =============================================== host side ===================================================
static inline void fsv_h_pack_taps(const int* ty, const int* tx, int n, unsigned long long& lo, unsigned long long& hi) {
  lo = 0; hi = 0;
  for (int t = 0; t < n; ++t) {
    unsigned long long c = (unsigned long long)((ty[t] + 8) & 15) | ((unsigned long long)((tx[t] + 8) & 15) << 4);
    if (t < 8) lo |= c << (t * 8); else hi |= c << ((t - 8) * 8);
  }
}

// One problem at the C ABI (include/fsv2v.h fsv_hconv_desc; ctypes mirror in few-shot-vid2vid_amd/conv.py) - MUST match both.
struct fsv_hconv_desc {
  const void* in; const void* wt; const float* bias; const void* res; void* out; const float* wscale;
  float* ws;               // fp32 workspace of N*outH*outW*Cout elements for split-K launches with a half output (or NULL: never split)
  double* stats;           // statistics partials (or NULL)
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, outH, outW, osy, osx, ooy, oox;
  int Kpad, nrows;
  int per_sample, act, accumulate;
  int out_h, res_h;
  int force_tile, force_split;
  int stats_groups, stats_slots, stats_prezeroed;
  float scale;
  long long w_bstride, b_bstride;
};

// tile ids: 0 = 128x128, 1 = 128x64, 2 = 128x32, 4 = 64x64, 9 = 64x128 (pixels x output channels) as 4-wave workgroups,
// 3 = 128x128 and 5 = 256x128 as 8-wave workgroups;
// + 16: the same tile with two LDS buffers (loads one chunk ahead) instead of three
static inline int fsv_h_tile_dims(int tile, int& bm, int& bn) {
  switch (tile & 15) {
    case 0: bm = 128; bn = 128; return 0;
    case 1: bm = 128; bn = 64; return 0;
    case 2: bm = 128; bn = 32; return 0;
    case 3: bm = 128; bn = 128; return 0;      // 8 waves
    case 4: bm = 64; bn = 64; return 0;
    case 5: bm = 256; bn = 128; return 0;      // 8 waves
    case 9: bm = 64; bn = 128; return 0;
    default: return -1;
  }
}

// FSV_HCONV_NBUF = 2 | 3: force the LDS buffer count of the tiles the plan picks (A/B runs; 0 / unset: the plan's own choice)
static inline int fsv_h_nbuf() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("FSV_HCONV_NBUF"); v = e ? atoi(e) : 0; if (v != 2 && v != 3) v = 0; }
  return v;
}

static inline long long fsv_h_wgs(int Mz, int Cout, int nsamp, int tile) {
  int bm, bn;
  fsv_h_tile_dims(tile, bm, bn);
  return (long long)fsv_cdiv(Mz, bm) * fsv_cdiv(Cout, bn) * nsamp;
}

// Tile of a launch over `Mz` pixels x `Cout` channels (in-box A/B of round 4, tools/h_ab.py, profiles/r04_notes.md): the kernel is
// bound by the rate at which the memory pipe fills LDS (~10 - 13 TB/s over the chip whatever the tile), so the widest tile wins
// as long as every CU still gets work, and below that OCCUPANCY wins - the 64x64 tile with two LDS buffers (32 KB, five workgroups
// per CU) beats the 64x128 / 128x64 forms on every shape where they leave fewer than two workgroups per CU.  128x128 runs as the
// 8-wave workgroup (two waves per SIMD: 605 against 397 TFLOP/s for the 4-wave form on M32768 N128 K2304).  Three LDS buffers
// (loads two chunks ahead) only where a CU holds a single workgroup; short K (two chunks) always two.
static inline int fsv_h_pick_tile(int Mz, int Cout, int nchunks, int nsamp, bool grouped) {
  int tile;
  if (Cout <= 32) tile = 2;
  else if (Cout <= 64) tile = 4;
  else if (!grouped && fsv_h_wgs(Mz, Cout, nsamp, 3) >= 256) tile = 3;
  else tile = fsv_h_wgs(Mz, Cout, nsamp, 9) >= 512 ? 9 : 4;
  int nbuf = (tile == 3 || grouped || nchunks <= 2 || fsv_h_wgs(Mz, Cout, nsamp, tile) > 320) ? 2 : 3;
  if (fsv_h_nbuf()) nbuf = fsv_h_nbuf();
  return nbuf == 2 ? tile + 16 : tile;
}

// Plan: tile as above; split-K (zeroed fp32 accumulation + finishing pass) when the launch covers less than three quarters of the chip
// and K is long.
extern "C" int fsv_hconv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split, int can_split,
                              int* tile_out, int* nsplit_out) {
  int tile = force_tile, nsplit = force_split > 0 ? force_split : 1;
  if (tile < 0) tile = fsv_h_pick_tile(Mz, Cout, nchunks, nsamp, false);
  int bm, bn;
  if (fsv_h_tile_dims(tile, bm, bn)) return -1;
  if (force_split <= 0) {
    const long long wgs = (long long)fsv_cdiv(Mz, bm) * fsv_cdiv(Cout, bn) * nsamp;
    const char* det = getenv("FSV_DETERMINISTIC");
    if (can_split && wgs < 192 && nchunks >= 16 && !(det && det[0] == '1')) {
      nsplit = (int)((512 + wgs - 1) / wgs);
      if (nsplit > 8) nsplit = 8;
      if (nsplit > nchunks / 6) nsplit = nchunks / 6;
      if (nsplit < 1) nsplit = 1;
    }
  }
  if (nsplit > nchunks) nsplit = nchunks;
  if (nsplit < 1) nsplit = 1;
  *tile_out = tile; *nsplit_out = nsplit;
  return 0;
}

static int fsv_h_fill(HConvP& p, const fsv_hconv_desc& d) {
  if (!d.in || !d.wt || !d.out || d.ntaps < 1 || d.ntaps > 16 || d.N < 1 || d.Cin < 1 || d.Cout < 1) return FSV_ERR_BAD_ARG;
  if ((d.Cin & 7) != 0 || (d.Kpad & 63) != 0 || d.Kpad < d.ntaps * d.Cin || d.nrows < d.Cout) return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < d.ntaps; ++t)
    if (d.ty[t] < -8 || d.ty[t] > 7 || d.tx[t] < -8 || d.tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((long long)d.N * d.H * d.W * d.Cin * 2 > FSV_BUF_MAX_BYTES || (long long)d.nrows * d.Kpad * 2 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  if (d.accumulate && (d.bias || d.res || d.act != FSV_ACT_NONE || d.scale != 1.f || d.out_h)) return FSV_ERR_BAD_ARG;
  if (d.act == FSV_ACT_DLRELU && !d.res) return FSV_ERR_BAD_ARG;
  {
    // the epilogue addresses the output with 32-bit element indices and the residual through one descriptor
    const long long oelems = (long long)d.N * d.outH * d.outW * d.Cout;
    if (oelems >= 0x7fffffffll || (d.res && oelems * (d.res_h ? 2 : 4) > FSV_BUF_MAX_BYTES)) return FSV_ERR_UNSUPPORTED;
    p.res_bytes = d.res ? oelems * (d.res_h ? 2 : 4) : 0;
  }
  p.in = reinterpret_cast<const fsv_h16*>(d.in); p.wt = reinterpret_cast<const fsv_h16*>(d.wt);
  p.bias = d.bias; p.res = d.res; p.wscale = d.wscale; p.out = d.out;
  p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.OH = d.OH; p.OW = d.OW; p.Cout = d.Cout;
  p.K = d.ntaps * d.Cin; p.nchunks = fsv_cdiv(p.K, FSV_HBK); p.Kpad = d.Kpad; p.nrows = d.nrows;
  p.sy = d.sy; p.sx = d.sx; p.ntaps = d.ntaps;
  fsv_h_pack_taps(d.ty, d.tx, d.ntaps, p.taps_lo, p.taps_hi);
  p.outH = d.outH; p.outW = d.outW; p.osy = d.osy; p.osx = d.osx; p.ooy = d.ooy; p.oox = d.oox;
  p.dense_out = (d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.outH == d.OH && d.outW == d.OW) ? 1 : 0;
  p.w_bstride = d.w_bstride; p.b_bstride = d.b_bstride; p.per_sample = d.per_sample ? 1 : 0;
  p.act = d.act; p.scale = d.scale;
  p.Mz = d.per_sample ? d.OH * d.OW : d.N * d.OH * d.OW;
  p.nsplit = 1;
  p.out_h = d.out_h ? 1 : 0; p.res_h = d.res_h ? 1 : 0;
  p.stats = nullptr; p.stats_slots = 1; p.stats_ohw = 1;
  return FSV_OK;
}

#define FSV_H_CASES(KERNEL, G, P)                                                                                     \
  switch (tile) {                                                                                                     \
    case 0: FSV_LAUNCH((KERNEL<128, 128, 2, 2, 3>), G, dim3(256), stream, P); break;                                  \
    case 1: FSV_LAUNCH((KERNEL<128, 64, 2, 2, 3>), G, dim3(256), stream, P); break;                                   \
    case 2: FSV_LAUNCH((KERNEL<128, 32, 4, 1, 3>), G, dim3(256), stream, P); break;                                   \
    case 3: FSV_LAUNCH((KERNEL<128, 128, 2, 4, 3>), G, dim3(512), stream, P); break;                                  \
    case 5: FSV_LAUNCH((KERNEL<256, 128, 4, 2, 3>), G, dim3(512), stream, P); break;                                  \
    case 4: FSV_LAUNCH((KERNEL<64, 64, 2, 2, 3>), G, dim3(256), stream, P); break;                                    \
    case 9: FSV_LAUNCH((KERNEL<64, 128, 2, 2, 3>), G, dim3(256), stream, P); break;                                   \
    case 16: FSV_LAUNCH((KERNEL<128, 128, 2, 2, 2>), G, dim3(256), stream, P); break;                                 \
    case 17: FSV_LAUNCH((KERNEL<128, 64, 2, 2, 2>), G, dim3(256), stream, P); break;                                  \
    case 18: FSV_LAUNCH((KERNEL<128, 32, 4, 1, 2>), G, dim3(256), stream, P); break;                                  \
    case 19: FSV_LAUNCH((KERNEL<128, 128, 2, 4, 2>), G, dim3(512), stream, P); break;                                 \
    case 21: FSV_LAUNCH((KERNEL<256, 128, 4, 2, 2>), G, dim3(512), stream, P); break;                                 \
    case 20: FSV_LAUNCH((KERNEL<64, 64, 2, 2, 2>), G, dim3(256), stream, P); break;                                   \
    case 25: FSV_LAUNCH((KERNEL<64, 128, 2, 2, 2>), G, dim3(256), stream, P); break;                                  \
    default: return FSV_ERR_BAD_ARG;                                                                                  \
  }

extern "C" {

// n == 1: one launch (split-K allowed when d->ws is given or the output is fp32); n > 1: ONE grouped launch of independent problems
// (no K splits; every problem as the tile the group's widest member takes).  Returns *produced = 1 when the statistics partials
// were written (single launches only).
int fsv_hconv_gather(const fsv_hconv_desc* d, int n, int* produced, hipStream_t stream) {
  if (!d || n < 1 || n > 64) return FSV_ERR_BAD_ARG;
  if (produced) *produced = 0;
  if (n == 1) {
    HConvP p;
    int rc = fsv_h_fill(p, d[0]);
    if (rc) return rc;
    const int nsamp = d->per_sample ? d->N : 1;
    const long long total = (long long)d->N * d->outH * d->outW * d->Cout;
    // a K split adds partial sums into a zeroed fp32 buffer: the output itself when it is fp32 and dense, else the workspace
    const bool can_split = !d->accumulate && p.dense_out && d->act != FSV_ACT_DLRELU && (!d->out_h || d->ws);
    int tile = 0, nsplit = 1;
    if (fsv_hconv_plan(p.Mz, d->Cout, p.nchunks, nsamp, d->force_tile, d->force_split, can_split ? 1 : 0, &tile, &nsplit)) return FSV_ERR_BAD_ARG;
    if (nsplit > 1 && !can_split) nsplit = 1;
    p.nsplit = nsplit;
    void* final_out = p.out;
    if (nsplit > 1) {
      float* acc = d->out_h ? d->ws : reinterpret_cast<float*>(p.out);
      (void)hipMemsetAsync(acc, 0, (size_t)total * sizeof(float), stream);
      p.out = acc;
    }
    if (d->stats && nsplit == 1 && !d->accumulate && !d->per_sample && p.dense_out && d->stats_groups >= 1 && d->stats_slots >= 1 &&
        p.Mz % d->stats_groups == 0 && (d->stats_groups == 1 || p.Mz / d->stats_groups >= 128)) {
      p.stats = d->stats; p.stats_slots = d->stats_slots; p.stats_ohw = p.Mz / d->stats_groups;
      if (!d->stats_prezeroed)
        (void)hipMemsetAsync(d->stats, 0, (size_t)d->stats_groups * d->stats_slots * d->Cout * 2 * sizeof(double), stream);
      if (produced) *produced = 1;
    }
    int bm, bn;
    if (fsv_h_tile_dims(tile, bm, bn)) return FSV_ERR_BAD_ARG;
    const int tiles_xy = fsv_cdiv(p.Mz, bm) * fsv_cdiv(d->Cout, bn);
    const dim3 g(8 * fsv_cdiv(tiles_xy, 8), 1, nsamp * nsplit);          // 1-D, padded: see fsv_h_xcd_tile
    FSV_H_CASES(fsv_hconv_kernel, g, p)
    rc = fsv_check_launch();
    if (rc) return rc;
    if (nsplit > 1 && (d->out_h || d->bias || d->res || d->act != FSV_ACT_NONE || d->scale != 1.f)) {
      int grid = (int)((total + 256 * 8 - 1) / (256 * 8));
      if (grid > 4096) grid = 4096;
      if (grid < 1) grid = 1;
      FSV_LAUNCH(fsv_hconv_finish_kernel, dim3(grid), dim3(256), stream, (const float*)p.out, final_out, d->bias, d->res, total,
                 d->Cout, (long long)d->outH * d->outW, d->per_sample ? d->b_bstride : 0ll, d->act, d->scale, d->out_h, d->res_h);
      rc = fsv_check_launch();
    }
    return rc;
  }
  // ---- grouped ---------------------------------------------------------------------------------------------------------------
  HConvP ps[64];
  int nsamp[64];
  long long weight[64];
  int max_cout = 0;
  long long wgs128 = 0;
  for (int i = 0; i < n; ++i) {
    int rc = fsv_h_fill(ps[i], d[i]);
    if (rc) return rc;
    nsamp[i] = d[i].per_sample ? d[i].N : 1;
    if (d[i].Cout > max_cout) max_cout = d[i].Cout;
    weight[i] = ps[i].nchunks;
  }
  int tile = d[0].force_tile;
  if (tile < 0) {
    // one tile shape for the whole group: the rule of the single launches applied to the group's total pixel count
    long long mz_total = 0;
    int min_chunks = 1 << 30;
    for (int i = 0; i < n; ++i) {
      mz_total += (long long)ps[i].Mz * nsamp[i] * fsv_cdiv(ps[i].Cout, max_cout > 0 ? max_cout : 1);
      if (ps[i].nchunks < min_chunks) min_chunks = ps[i].nchunks;
    }
    tile = fsv_h_pick_tile((int)(mz_total > (1ll << 30) ? (1ll << 30) : mz_total), max_cout, min_chunks, 1, true);
  }
  (void)wgs128;
  int bm, bn;
  if (fsv_h_tile_dims(tile, bm, bn)) return FSV_ERR_BAD_ARG;
  int order[64];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i) {            // longest problems first
    const int v = order[i];
    int j = i - 1;
    while (j >= 0 && weight[order[j]] < weight[v]) { order[j + 1] = order[j]; --j; }
    order[j + 1] = v;
  }
  for (int b0 = 0; b0 < n; b0 += FSV_GROUP_MAX) {
    HConvGroup g;
    g.nprob = (n - b0 < FSV_GROUP_MAX) ? (n - b0) : FSV_GROUP_MAX;
    int tiles = 0;
    for (int j = 0; j < FSV_GROUP_MAX; ++j) {
      if (j < g.nprob) {
        const int i = order[b0 + j];
        g.p[j] = ps[i];
        tiles += fsv_cdiv(ps[i].Mz, bm) * fsv_cdiv(ps[i].Cout, bn) * nsamp[i];
      } else {
        g.p[j] = ps[order[b0]];
      }
      g.tile_end[j] = tiles;
    }
    const dim3 grid(tiles);
    FSV_H_CASES(fsv_hconv_group_kernel, grid, g)
  }
  return fsv_check_launch();
}

// weight gradient from half activations / half output gradients into the fp32 K-major layout dwt[Kpad][ldw]
int fsv_hconv_wgrad(const void* in, const void* dout, float* dwt,
                    int N, int H, int W, int Cin, int OH, int OW, int Cout,
                    int ntaps, const int* ty, const int* tx, int sy, int sx,
                    int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                    int force_tile, hipStream_t stream) {
  if (!in || !dout || !dwt || ntaps < 1 || ntaps > 16) return FSV_ERR_BAD_ARG;
  if ((Cin & 7) != 0 || (Cout & 7) != 0) return FSV_ERR_UNSUPPORTED;
  // the pixel walk advances (n, oy, ox) by 64 pixels with ONE carry per level: 64 / OW rows (+ 1 when the column wraps) may not exceed OH
  if (FSV_HBK / OW + ((FSV_HBK % OW) ? 1 : 0) > OH) return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((long long)N * H * W * Cin * 2 > FSV_BUF_MAX_BYTES || (long long)N * OH * OW * Cout * 2 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  HWgradP p;
  p.in = reinterpret_cast<const fsv_h16*>(in); p.dout = reinterpret_cast<const fsv_h16*>(dout); p.dwt = dwt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.ldw = ldw; p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  fsv_h_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi);
  p.w_bstride = w_bstride; p.per_sample = per_sample ? 1 : 0;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  p.pchunks = fsv_cdiv(p.Mz, FSV_HBK);
  const int nsamp = per_sample ? N : 1;
  // tiles (rows = taps * Cin, columns = Cout): force_tile 1 = 64x64, 2 = 128x64, 3 = 64x128, 4 = 128x128, 5 = 128x32, 6 = 128x64 as an
  // 8-wave workgroup.  Plan (in-box A/B of round 4, tools/h_ab.py): 128x64 beats 128x128 wherever the tile count is small (the
  // deep layers) under the fp32 kernel's 1024-workgroup split rule, but with ONE wave of workgroups
  // over the chip (~288) with as many 64-pixel chunks per workgroup as that leaves beats both the 1024-workgroup target of the fp32
  // kernel (short reductions drown in their atomic epilogues: 142 -> 194 on pix8192 N256 K2304) and too few workgroups.
  int bmk = 128, bn = Cout > 64 ? 128 : (Cout > 32 ? 64 : 32);
  if (p.K <= 64 && bn >= 64) bmk = 64;
  int w8 = 0;
  if (force_tile == 1) { bmk = 64; bn = 64; }
  else if (force_tile == 2) { bmk = 128; bn = 64; }
  else if (force_tile == 3) { bmk = 64; bn = 128; }
  else if (force_tile == 4) { bmk = 128; bn = 128; }
  else if (force_tile == 5) { bmk = 128; bn = 32; }
  else if (force_tile == 6) { bmk = 128; bn = 64; w8 = 1; }
  const long long blocks = (long long)fsv_cdiv(p.K, bmk) * fsv_cdiv(Cout, bn) * nsamp;
  int nsplit = 1;
  const char* det = getenv("FSV_DETERMINISTIC");
  if (force_split > 0) nsplit = force_split;
  else if (!(det && det[0] == '1')) {
    // (the 128x32 tile - 40 KB of LDS, four workgroups per CU - wants them all: 138 against 92 TFLOP/s on pix524288 N32 K288)
    const long long target = bn == 32 ? 1024 : 288;
    nsplit = (int)((target + blocks - 1) / blocks);
    const int maxs = p.pchunks / 4;                 // at least 4 chunks (256 pixels) per split
    if (nsplit > maxs) nsplit = maxs;
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > p.pchunks) nsplit = p.pchunks;
  p.nsplit = nsplit;
  if (nsplit > 1 && !prezeroed)
    (void)hipMemsetAsync(dwt, 0, (size_t)((per_sample ? (long long)N * w_bstride : (long long)Kpad * ldw)) * sizeof(float), stream);
  const dim3 g(fsv_cdiv(p.K, bmk), fsv_cdiv(Cout, bn), nsamp * nsplit), block(256);
  if (w8) FSV_LAUNCH((fsv_hconv_wgrad_kernel<128, 64, 4, 2>), g, dim3(512), stream, p);
  else if (bmk == 128 && bn == 128) FSV_LAUNCH((fsv_hconv_wgrad_kernel<128, 128, 2, 2>), g, block, stream, p);
  else if (bmk == 128 && bn == 64) FSV_LAUNCH((fsv_hconv_wgrad_kernel<128, 64, 2, 2>), g, block, stream, p);
  else if (bmk == 64 && bn == 128) FSV_LAUNCH((fsv_hconv_wgrad_kernel<64, 128, 2, 2>), g, block, stream, p);
  else if (bmk == 64 && bn == 64) FSV_LAUNCH((fsv_hconv_wgrad_kernel<64, 64, 2, 2>), g, block, stream, p);
  else if (bmk == 128 && bn == 32) FSV_LAUNCH((fsv_hconv_wgrad_kernel<128, 32, 4, 1>), g, block, stream, p);
  else return FSV_ERR_BAD_ARG;
  return fsv_check_launch();
}

// table-driven conversion of K-major fp32 layouts into the N-major half operands (see fsv_hconv_prep_kernel)
int fsv_hconv_prep_weight(const long long* jobs, const int* tmap, int nblocks, hipStream_t stream) {
  if (!jobs || !tmap || nblocks < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_hconv_prep_kernel, dim3(nblocks), dim3(256), stream, jobs, tmap);
  return fsv_check_launch();
}

int fsv_hconv_prep_weight_one(const float* src, void* dst, int Kpad32, int ldw, int nrows, int Kpad64, int nbatch, hipStream_t stream) {
  if (!src || !dst || Kpad32 < 1 || ldw < 1 || nrows < 1 || Kpad64 < 1 || nbatch < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_hconv_prep_one_kernel, dim3(fsv_cdiv(Kpad64, 64), fsv_cdiv(nrows, 64), nbatch), dim3(256), stream, src,
             reinterpret_cast<fsv_h16*>(dst), Kpad32, ldw, nrows, Kpad64);
  return fsv_check_launch();
}

// dense element conversion: dir 0 = fp32 -> half, 1 = half -> fp32
int fsv_cast_half(const void* x, void* y, long long n, int dir, hipStream_t stream) {
  if (!x || !y || n < 0) return FSV_ERR_BAD_ARG;
  if (n == 0) return FSV_OK;
  long long g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  if (dir == 0) FSV_LAUNCH(fsv_cast_f2h_kernel, dim3((unsigned)g), dim3(256), stream, (const float*)x, (fsv_h16*)y, n);
  else FSV_LAUNCH(fsv_cast_h2f_kernel, dim3((unsigned)g), dim3(256), stream, (const fsv_h16*)x, (float*)y, n);
  return fsv_check_launch();
}

}  // extern "C"
