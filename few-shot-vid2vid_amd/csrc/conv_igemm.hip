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
// 1/sigma).  Tiles are staged through two LDS buffers (one barrier per 32-wide K chunk): A as [row][8 quads of 4 k] with the
// quads XOR-swizzled by the row (ds_write_b128 stores; fragments read with ds_read_b128 + a per-MFMA select, or - the AF
// variants every tile shape runs as since round 3 - as a re-ordered quad read in place with ds_read_b64), B as [k][n] as it
// lies in HBM; the scalar-gather twin (Cin % 4 != 0) keeps round 1's transposed [k][m] image.  The 64x128 tile - the template most of
// the step's time is spent in - loads straight into LDS (buffer_load_dwordx4 ... lds, three buffers, no register staging: the LD
// form below).  Details at the kernels below.
// The MFMA result is bitwise an fp32 fma chain (guide section 3), which is what lets the parity tests use a
// 1e-3 relative tolerance against the fp32 CPU oracle with a wide margin.
#include <stdlib.h>
#include "conv_igemm.h"

// Byte offsets inside one tensor are 32-bit (checked on the host: tensor < 2 GiB); rows / taps outside the image, K-tail
// columns and weight columns beyond ldw are loaded at the out-of-range offset FSV_BUF_OOB (hardware zero fill, conv_igemm.h)
// instead of being zeroed after the load.  That matters: a select on the LOADED value makes the compiler wait for the load
// right where it was issued (s_waitcnt vmcnt(0) at the top of the MFMA block - the whole L2 / HBM latency exposed once per
// K chunk, measured round 2: an fp16-operand twin of the old loop ran only 1.3x faster than fp32 although its MFMA work
// is 1/16); now the loaded registers are first touched by the LDS stores behind the chunk's MFMAs.

// XCD-aware tile order (MI355X_MICROARCH.md "Workgroup dispatch": linear workgroup b runs on XCD b % 8, each XCD has its
// own L2).  The workgroups that share the activation tile of one pixel tile (all `ny` channel tiles) are given ids that
// are congruent mod 8 and adjacent in dispatch order, so the gathered activations are fetched into ONE L2.
__device__ __forceinline__ void fsv_xcd_tile(int nx, int ny, int& bx, int& by) {
  const int b = blockIdx.x + blockIdx.y * nx;
  if ((nx & 7) != 0 || ny == 1) { bx = blockIdx.x; by = blockIdx.y; return; }
  const int xcd = b & 7, j = b >> 3;
  by = j % ny;
  bx = xcd + 8 * (j / ny);
}

// V4 kernel (Cin % 4 == 0).  LDS images, two buffers each (one barrier per 32-wide K chunk):
//   A [BM rows][8 quads of 4 k]: quad q of row r sits in slot q ^ ((r >> 1) & 7) - written with ds_write_b128 (8 lanes = one
//     row = 8 distinct slots) and read back as MFMA fragments with ds_read_b128 (16-lane service groups see 16 distinct
//     (row parity, slot) pairs): both conflict free, and 4x fewer LDS instructions than a transposed [k][m] image;
//   B [32 k][BN] as it lies in HBM: ds_write_b128 rows, ds_read_b32 fragments (32 consecutive columns).
// Two b128 reads of A feed the four MFMA steps of a k-group; the k order of every output's fma chain is ascending.
// The LDS fragment reads of k-group g + 1 are issued (into a second register set) BEFORE the MFMAs of group g and pinned
// there with scheduling fences - left alone the compiler sinks every read next to its MFMAs and waits for it (read,
// s_waitcnt lgkmcnt(0), two MFMAs, read, ...), which exposes the LDS latency whenever a SIMD holds a single wave.
// PF: prefetch distance of the global loads in chunks.  PF = 1 (every tile of the launch plan): the loads of chunk k + 1 are
// issued at the top of chunk k and stored to LDS behind its 12th MFMA.  PF = 2 (tile ids 10 - 12, reachable through force_tile
// or FSV_CONV_PF2=1 - built at the end of round 2 from the ISA reading in profiles/r02_notes.md section 15; measured there:
// section 16):
// two register sets, the loads of chunk k + 2 are issued at the top of chunk k and the set stored behind the 12th MFMA was
// loaded a whole chunk earlier, for grids with one workgroup per CU where nothing else covers the HBM latency.
// DBG (only with -DFSV_DIAG, tools/knockout.py): bit mask of the loop's parts that are left out to see what each costs -
// 1 barrier, 2 LDS stores, 4 global loads, 8 fragment reads, 16 offset arithmetic, 32 epilogue stores (PF = 2 loop only)
// LD (experimental, 8-wave tiles, PF = 2 + AF only): the global loads write LDS directly (buffer_load_dwordx4 ... lds: a wave's 64
// quads land in 1 KB of consecutive LDS, so lanes are mapped to the LDS image and the slot swizzle of A moves to the GLOBAL side - lane
// (row r, slot s) fetches quad s ^ swz(r)); no register staging, no ds_write, three LDS buffers instead of two register sets.  The
// quad of a row then lies in memory order (k0 k1 k2 k3), so in-place fragments pair the k of an MFMA step as (k, k + 2): lanes 0-31
// read (k0 k1), lanes 32-63 (k2 k3) - every output's fp32 chain sums the same products in the order k0 k2 k1 k3 instead of ascending.
// MODE 1 (PF = 2): whole trips - the loop always runs its two chunks per trip; chunks past the end of the K range (or of this
// split's share of it) load zeros and are multiplied like the others.  The exit between the two chunks of a trip made the compiler
// keep the accumulators in two register sets and copy one into the other after every first chunk (s_nop 16 + 8 v_mov_b64 + s_nop:
// the wave waits for its last MFMA).  MODE 2 = LD, whole trips of three.
// UP (round 5): the input tensor is stored at half the resolution and read through the nearest x2 up-sampling index - p.H / p.W are
// the logical (up-sampled) size, tap validity is tested against them, the address is that of source pixel (iy >> 1, ix >> 1).  The
// offset is no longer (pixel base + tap offset): two shifts, a multiply-add and a multiply per row and chunk, computed one chunk
// ahead like the others; its own instantiations, so that the instruction streams of the plain kernels stay what was validated.
template <int BM, int BN, int WM, int WN, int PF, bool AF, int DBG = 0, int MODE = 0, bool UP = false>
__device__ __forceinline__ void fsv_conv_igemm_body(const ConvP& p, const int bx, const int by, const int bz) {
  constexpr bool LD = MODE == 2, WT = MODE >= 1;
  constexpr int BK = FSV_BK;
  constexpr int NT = 64 * WM * WN;    // 4 or 8 waves
  constexpr int RPA = NT / 8;         // A rows per pass (8 threads per row)
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  constexpr int NPA = BM / RPA;       // A rows per thread and chunk
  constexpr int QB = BN / 4;          // B float4 per k row
  constexpr int RPB = NT / QB;        // B rows per pass
  constexpr int NPB = BK / RPB;       // B passes
  constexpr int A_ST = BM * BK, B_ST = BK * BN;
  static_assert(TM >= 1 && TN >= 1 && NPA >= 1 && NPA * RPA == BM && NPB >= 1 && NPB * RPB == BK, "tile / thread-count mismatch");
  constexpr int NBUF = LD ? 3 : 2;
  static_assert(!LD || (PF == 2 && AF), "LD form: prefetch distance 2, in-place fragments");
  __shared__ __attribute__((aligned(16))) float smem[NBUF * (A_ST + B_ST)];
  float* const As = smem;
  float* const Bs = smem + NBUF * A_ST;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bm0 = bx * BM, bn0 = by * BN;
  const float* wt = p.wt + (long long)zs * p.w_bstride;

  // ---- per-thread A row bookkeeping ------------------------------------------------------------------
  // LD: the lane's LDS slot is fixed by its lane id; the quad it fetches is the one that belongs there (rows of one thread are
  // RPA = 64 apart: the same swizzle for all of them)
  const int ar0 = tid >> 3;
  const int kq = LD ? ((tid & 7) ^ ((ar0 >> 1) & 7)) : (tid & 7);
  int a_iy0[NPA], a_ix0[NPA], a_pix[NPA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    int m = bm0 + ar0 + i * RPA;
    if (m < p.Mz) {
      int n, rem;
      if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
      int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy0[i] = oy * p.sy; a_ix0[i] = ox * p.sx;
      a_pix[i] = UP ? n * (p.H >> 1)                                       // UP: first source row of the sample
                    : ((n * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.Cin * 4;      // byte offset of tap (0, 0), channel 0
    } else {
      a_iy0[i] = -(1 << 28); a_ix0[i] = 0; a_pix[i] = 0;
    }
  }
  const int bq = tid % QB, br0 = tid / QB;
  const int bcol = bn0 + bq * 4;
  const bool bcol_ok = bcol < p.ldw;
  const long long in_bytes = UP ? (long long)p.N * (p.H >> 1) * (p.W >> 1) * p.Cin * 4 : (long long)p.N * p.H * p.W * p.Cin * 4;
  const fsv_buf abuf = fsv_make_buf(p.in, in_bytes);
  const fsv_buf bbuf = fsv_make_buf(wt, (long long)p.nchunks * BK * p.ldw * 4);
  const fsv_rawbuf araw = fsv_make_rawbuf(p.in, in_bytes);          // LD form only
  const fsv_rawbuf braw = fsv_make_rawbuf(wt, (long long)p.nchunks * BK * p.ldw * 4);

  // chunk range of this K split
  const int cps = (p.nchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.nchunks) ? (c_begin + cps) : p.nchunks;

  constexpr int NSET = LD ? 1 : (PF >= 2 ? 2 : 1);        // register sets of global loads in flight (LD: unused)
  float4 areg[NSET][NPA], breg[NSET][NPB];
  unsigned aoff[NPA], boff[NPB];
  // Byte offsets of one chunk's loads (FSV_BUF_OOB = "absent": hardware zero fill), computed one chunk ahead of their loads
  // so that this VALU work sits between the MFMAs instead of in front of them.  Branch-free on purpose (a guarded integer
  // division becomes a basic block of its own, which the scheduler cannot interleave with the MFMAs): the thread's K
  // position (tap t, channel ci) is advanced by 32 per chunk with precomputed 32 / Cin and 32 % Cin.  Chunks past c_end
  // are harmless: k >= K and weight rows >= nchunks * 32 are out of range for their descriptors.
  int cur_k = c_begin * BK + kq * 4;
  int cur_t = cur_k / p.Cin;
  int cur_ci = cur_k - cur_t * p.Cin;
  const int q32 = BK / p.Cin, r32 = BK - q32 * p.Cin;
  int cur_b = (c_begin * BK + br0) * p.ldw + bcol;      // element offset of this thread's first weight row
  // whole trips: chunks at or past c_end load zeros on both sides (another split's share of K must not be multiplied here)
  const int k_lim = WT ? (c_end * BK < p.K ? c_end * BK : p.K) : p.K;
  const int kb_lim = c_end * BK + kq * 4;
  auto calc_offsets = [&]() {
    const bool kok = cur_k < k_lim;
    const bool b_live = WT ? (bcol_ok & (cur_k < kb_lim)) : bcol_ok;
    int ty, tx;
    fsv_tap(p, cur_t, ty, tx);
    const int toff = ((ty * p.W + tx) * p.Cin + cur_ci) * 4;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int iy = a_iy0[i] + ty, ix = a_ix0[i] + tx;
      // `&`, not `&&`: a short-circuit here turns the whole offset computation into a guarded basic block
      const bool ok = kok & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      if constexpr (UP) {
        const unsigned off = (unsigned)((((a_pix[i] + (iy >> 1)) * (p.W >> 1) + (ix >> 1)) * p.Cin + cur_ci) * 4);
        aoff[i] = ok ? off : FSV_BUF_OOB;
      } else if constexpr (WT) aoff[i] = (unsigned)(a_pix[i] + toff) | (ok ? 0u : FSV_BUF_OOB);      // any offset >= 2^31 is out of range
      else aoff[i] = ok ? (unsigned)(a_pix[i] + toff) : FSV_BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) boff[i] = b_live ? (unsigned)((cur_b + i * RPB * p.ldw) * 4) : FSV_BUF_OOB;
    cur_k += BK;
    cur_t += q32;
    cur_ci += r32;
    const bool wrap = cur_ci >= p.Cin;
    cur_ci = wrap ? cur_ci - p.Cin : cur_ci;
    cur_t = wrap ? cur_t + 1 : cur_t;
    cur_b += BK * p.ldw;
  };
  auto issue_loads = [&](float4 (&ar)[NPA], float4 (&br)[NPB]) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) ar[i] = fsv_buf_load4(abuf, aoff[i]);
#pragma unroll
    for (int i = 0; i < NPB; ++i) br[i] = fsv_buf_load4(bbuf, boff[i]);
  };
  // LD: the same loads straight into LDS buffer nb; a wave's lanes cover 8 consecutive A rows (8 slots each) resp. 64 / QB
  // consecutive B rows - exactly the 1 KB the instruction writes
  auto issue_loads_lds = [&](float* a_buf, float* b_buf) {
    float* a_dst = a_buf + wave * (8 * BK);
    float* b_dst = b_buf + wave * ((64 / QB) * BN);
#pragma unroll
    for (int i = 0; i < NPA; ++i) fsv_buf_load4_lds(araw, aoff[i], a_dst + i * (RPA * BK));
#pragma unroll
    for (int i = 0; i < NPB; ++i) fsv_buf_load4_lds(braw, boff[i], b_dst + i * (RPB * BN));
  };
  // AF (round 3): the quad (k0 k1 k2 k3) of a row is stored as (k0 k2 | k1 k3) - rows with bit 4 set as
  // (k1 k3 | k0 k2) - so that a lane reads exactly the two values its MFMA steps multiply with ONE ds_read_b64 (lanes 0-31 take
  // the even k of a quad, lanes 32-63 the odd ones; the bit-4 swap spreads the 32 lanes of a ds_read_b64 service group over
  // all 32 bank pairs).  The b128 form makes every MFMA wait for a v_cndmask + s_nop that picks its operand out of the quad -
  // an issue slot between two MFMAs on the SAME accumulator, which costs the matrix pipe ~40 cycles each time
  // (MI355X_MICROARCH.md, instruction timing); with the operands read in place the four MFMAs of a k-group issue back to back.
  auto store_chunk = [&](int buf, const float4 (&ar)[NPA], const float4 (&br)[NPB]) {
    float* a_dst = As + buf * A_ST;
    float* b_dst = Bs + buf * B_ST;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int r = ar0 + i * RPA;
      float4 v = ar[i];
      if constexpr (AF) {
        const bool hi = (r >> 4) & 1;
        v.x = hi ? ar[i].y : ar[i].x; v.y = hi ? ar[i].w : ar[i].z;
        v.z = hi ? ar[i].x : ar[i].y; v.w = hi ? ar[i].z : ar[i].w;
      }
      *reinterpret_cast<float4*>(&a_dst[r * BK + ((kq ^ ((r >> 1) & 7)) << 2)]) = v;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int kr = br0 + i * RPB;
      *reinterpret_cast<float4*>(&b_dst[kr * BN + bq * 4]) = br[i];
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
  // fragment addresses inside a buffer: A row (wm, i, lrow), quads 2g and 2g + 1; B row 8g + 2s + lk, column (wn, j, lrow)
  int a_off[TM], a_swz[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * (TM * 32) + i * 32 + lrow;
    a_off[i] = r * BK + (LD ? 2 * lk : (AF ? 2 * (lk ^ ((r >> 4) & 1)) : 0));       // AF: this lane's half of every quad
    a_swz[i] = (r >> 1) & 7;
  }
  // LD: lanes 32-63 multiply k + 2 of a quad (rows 2 apart), not k + 1
  const int b_off = (LD ? 2 * lk : lk) * BN + wn * (TN * 32) + lrow;

  // fragments of one k-group (8 k): the two quads of A per row tile, four B values per column tile.  MFMA step s of the
  // group multiplies k = 8g + 2s (lanes 0-31) and k = 8g + 2s + 1 (lanes 32-63): the sum over k stays ONE ascending fp32 fma
  // chain per output, the order of the reference's own arithmetic (a permuted order moved LeakyReLU kinks of near-zero
  // activations in the reference's finetune fixture and its gradients by 1 %).
  auto read_group = [&](const float* a_src, const float* b_src, int g, float4 (&a4)[2][TM], float (&b)[4][TN]) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (AF) {
          const float2 v = *reinterpret_cast<const float2*>(&a_src[a_off[i] + (((2 * g + q) ^ a_swz[i]) << 2)]);
          a4[q][i].x = v.x; a4[q][i].y = v.y;
        } else {
          a4[q][i] = *reinterpret_cast<const float4*>(&a_src[a_off[i] + (((2 * g + q) ^ a_swz[i]) << 2)]);
        }
      }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j) b[t][j] = b_src[b_off + (LD ? 8 * g + 4 * (t >> 1) + (t & 1) : 8 * g + 2 * t) * BN + j * 32];
  };
  auto mma_group = [&](const float4 (&a4)[2][TM], const float (&b)[4][TN]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 v = a4[t >> 1][i];
        const float a = AF ? ((t & 1) ? v.y : v.x) : ((t & 1) ? (lk ? v.w : v.z) : (lk ? v.y : v.x));
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t][j], acc[i][j], 0, 0, 0);
      }
  };

  // one K chunk: the loads issued at its top go into (lar, lbr); (sar, sbr) is the set stored into the other LDS buffer behind
  // three quarters of its MFMAs - the same set for PF = 1, the one loaded a chunk earlier for PF = 2
  auto chunk = [&](int buf, float4 (&lar)[NPA], float4 (&lbr)[NPB], const float4 (&sar)[NPA], const float4 (&sbr)[NPB]) {
    if constexpr (!(DBG & 4)) issue_loads(lar, lbr);
    const float* a_src = As + buf * A_ST;
    const float* b_src = Bs + buf * B_ST;
    float4 fa[2][2][TM];
    float fb[2][4][TN];
    if constexpr (DBG & 8) {
      // one read per chunk instead of four: the operands stay defined, the LDS traffic of the fragments goes
      read_group(a_src, b_src, 0, fa[0], fb[0]);
      read_group(a_src, b_src, 1, fa[1], fb[1]);
      FSV_SCHED_FENCE();
      if constexpr (!(DBG & 16)) calc_offsets();
      mma_group(fa[0], fb[0]);
      FSV_SCHED_FENCE();
      mma_group(fa[1], fb[1]);
      FSV_SCHED_FENCE();
      mma_group(fa[0], fb[0]);
      FSV_SCHED_FENCE();
      if constexpr (!(DBG & 2)) store_chunk(buf ^ 1, sar, sbr);
      FSV_SCHED_FENCE();
      mma_group(fa[1], fb[1]);
    } else {
    read_group(a_src, b_src, 0, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    if constexpr (!(DBG & 16)) calc_offsets();
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
    if constexpr (!(DBG & 2)) store_chunk(buf ^ 1, sar, sbr);
    FSV_SCHED_FENCE();
    mma_group(fa[1], fb[1]);
    }
    if constexpr (!(DBG & 1)) __syncthreads();
  };

  // LD: chunk c is multiplied out of LDS buffer c % 3 while the loads of chunk c + 2 fill buffer (c + 2) % 3 - the one chunk c - 1 was
  // read from, free since the barrier that closed it; before this chunk's barrier only the loads of chunk c + 1 are waited for
  // one chunk out of (a_src, b_src) while the loads of the chunk after next fill (a_dma, b_dma)
  auto chunk_lds = [&](float* a_dma, float* b_dma, const float* a_src, const float* b_src) {
    issue_loads_lds(a_dma, b_dma);
    float4 fa[2][2][TM];
    float fb[2][4][TN];
    read_group(a_src, b_src, 0, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    calc_offsets();
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
    mma_group(fa[1], fb[1]);
    FSV_WAIT_VMCNT(NPA + NPB);
    __syncthreads();
  };
  // the loop over three LDS buffers (a fourth - loads three chunks ahead, 96 KB - costs 0.6 ... 1.1 ms per step in occupancy); the loads of a chunk are waited for explicitly (FSV_WAIT_VMCNT) in front of the barrier that
  // publishes them - issued through inline assembly, they are invisible to the compiler's own wait-count pass (conv_igemm.h)
  auto loop_lds = [&]() {
    float* const A0 = As, * const A1 = As + A_ST, * const A2 = As + 2 * A_ST;
    float* const B0 = Bs, * const B1 = Bs + B_ST, * const B2 = Bs + 2 * B_ST;
    calc_offsets();
    issue_loads_lds(A0, B0);
    calc_offsets();
    issue_loads_lds(A1, B1);
    calc_offsets();
    FSV_WAIT_VMCNT(NPA + NPB);
    __syncthreads();
#pragma unroll 1
    for (int kc = c_begin; kc < c_end; kc += 3) {
      chunk_lds(A2, B2, A0, B0);
      chunk_lds(A0, B0, A1, B1);
      chunk_lds(A1, B1, A2, B2);
    }
  };
  if (c_begin < c_end) {
    if constexpr (LD) {
      loop_lds();
    } else if constexpr (PF == 1) {
      calc_offsets();
      issue_loads(areg[0], breg[0]);
      calc_offsets();
      store_chunk(0, areg[0], breg[0]);
      __syncthreads();
      int buf = 0;
#pragma unroll 1
      for (int kc = c_begin; kc < c_end; ++kc) {
        // the next chunk's loads are issued first (offsets were computed during the previous iteration) and land in
        // registers under the first three quarters of this chunk's MFMAs; they are stored into the OTHER buffer (nobody
        // reads it during this iteration), so one barrier per chunk suffices.  The copy stored by the last iteration is never
        // used.  (Written out instead of calling chunk(): the plan's kernels keep the exact instruction stream that was
        // validated on hardware.)
        issue_loads(areg[0], breg[0]);
        const float* a_src = As + buf * A_ST;
        const float* b_src = Bs + buf * B_ST;
        float4 fa[2][2][TM];
        float fb[2][4][TN];
        read_group(a_src, b_src, 0, fa[0], fb[0]);
        FSV_SCHED_FENCE();
        calc_offsets();
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
        store_chunk(buf ^ 1, areg[0], breg[0]);
        FSV_SCHED_FENCE();
        mma_group(fa[1], fb[1]);
        __syncthreads();
        buf ^= 1;
      }
    } else {
      // chunk c_begin -> LDS buffer 0, chunk c_begin + 1 -> register set 1, offsets of c_begin + 2 ready; then two chunks per
      // trip so that the register sets are addressed statically: an even step loads c + 2 into set 0 and stores set 1 (c + 1),
      // an odd step the other way round.  Chunks past c_end read zeros (see calc_offsets) and are never multiplied.
      calc_offsets();
      issue_loads(areg[0], breg[0]);
      calc_offsets();
      store_chunk(0, areg[0], breg[0]);
      issue_loads(areg[1], breg[1]);
      calc_offsets();
      __syncthreads();
#pragma unroll 1
      for (int kc = c_begin; kc < c_end; kc += 2) {
        chunk(0, areg[0], breg[0], areg[1], breg[1]);
        if constexpr (!WT) { if (kc + 1 >= c_end) break; }
        chunk(1, areg[1], breg[1], areg[0], breg[0]);
      }
    }
  }

  // ---- epilogue: D layout col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) -----------
  const float* bias = p.bias ? (p.bias + (long long)zs * p.b_bstride) : nullptr;
  const float ws = p.wscale ? p.wscale[0] : 1.f;
  // statistics for the normalisation that follows (p.stats, host side: only with nsplit == 1): rows below `gsplit` belong to the
  // tile's first group, the rest (InstanceNorm, a tile that straddles two samples) to the next one
  const int st_g0 = p.stats ? bm0 / p.stats_ohw : 0;
  const int st_split = (st_g0 + 1) * (p.stats ? p.stats_ohw : 0);
  auto out_pixel = [&](int m) -> long long {
    if (p.dense_out) return (long long)zs * (p.per_sample ? p.Mz : 0) + m;
    int n, rem;
    if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    return ((long long)n * p.outH + (oy * p.osy + p.ooy)) * p.outW + (ox * p.osx + p.oox);
  };
  // residual / LeakyReLU-mask operand: all of a lane's values are loaded (through a descriptor, absent ones as zero fill) BEFORE
  // its first store - the stores may alias p.res as far as the compiler can tell, so a load inside the store loop waits for its
  // full memory latency once per element, 16 - 32 times per tile (round 4)
  const bool pre = p.res_bytes > 0;
  const fsv_buf rbuf = fsv_make_buf(p.res, pre ? p.res_bytes : 0);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = bn0 + wn * (TN * 32) + j * 32 + lrow;
    const bool cok = co < p.Cout;
    const float bv = (bias && p.nsplit == 1 && cok) ? bias[co] : 0.f;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    float aux[TM][16];
    if (pre) {                // uniform
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int m = bm0 + wm * (TM * 32) + i * 32 + row;
          const bool ok = cok & (m < p.Mz);
          aux[i][r] = fsv_buf_load1(rbuf, ok ? (unsigned)((out_pixel(m) * p.Cout + co) * 4) : FSV_BUF_OOB);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = bm0 + wm * (TM * 32) + i * 32 + row;
        if (m >= p.Mz || !cok) continue;
        const long long opix = out_pixel(m);
        float* dst = p.out + opix * p.Cout + co;
        float v = acc[i][j][r] * ws;
        if (p.nsplit > 1) {
          if (p.part) p.part[(long long)zk * p.part_stride + opix * p.Cout + co] = v;      // uniform: ordered split
          else atomicAdd(dst, v);
        } else {
          v = (v + bv) * p.scale;
          if (p.act == FSV_ACT_DLRELU) {        // data gradient handed straight to the layer below: times LeakyReLU'(its output)
            const float a = pre ? aux[i][r] : p.res[opix * p.Cout + co];
            v = a > 0.f ? v : 0.2f * v;
          } else {
            v = fsv_act(v, p.act);
            if (p.res) v += pre ? aux[i][r] : p.res[opix * p.Cout + co];
          }
          if constexpr (!(DBG & 32)) *dst = v; else if (v == 1.2345e30f) *dst = v;      // keeps the arithmetic alive
          if (p.stats) {
            if (m < st_split) { s0 += v; q0 += v * v; } else { s1 += v; q1 += v * v; }
          }
        }
      }
    }
    if (p.stats) {            // uniform
      // the two half-waves hold the same 32 channels (rows 4 lk + ...): fold them, then one fp64 atomic pair per channel
      s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32);
      s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
      if (lk == 0 && cok) {
        const int slot = bx % p.stats_slots;
        double* d = p.stats + (((long long)st_g0 * p.stats_slots + slot) * p.Cout + co) * 2;
        atomicAdd(d, (double)s0); atomicAdd(d + 1, (double)q0);
        if (bm0 + BM > st_split && st_split < p.Mz) {       // uniform: the tile reaches into the next group
          d += (long long)p.stats_slots * p.Cout * 2;
          atomicAdd(d, (double)s1); atomicAdd(d + 1, (double)q1);
        }
      }
    }
  }
}

// (Round 5 measured a finishing-pass-free form of the ordered split - the workgroup that takes a tile's last ticket sums the copies
// and applies the epilogue: device-scope fence, ticket, fence, as in the fused reductions of norm.hip.  Bit-identical, and +4.4 ...
// +5.1 ms per step in two forms (per-lane dependent loads; cooperative walk with 4 x nsplit loads in flight): isolated, M512 N1024
// K4608 split 8 ran at 31 TFLOP/s against 65 with the finishing launch - the device-scope release every split workgroup pays (an L2
// write-back on this part) costs more than the 16 us launch it saves.  profiles/r05_notes.md section 4; removed.)

// XCD bands (round 4; the half-precision kernel's order, conv_h.hip): XCD x - the workgroups with linear id b % 8 == x - owns the
// CONTIGUOUS run of tiles [start(x), start(x + 1)) in (pixel tile, channel tile) order with the channel tile fastest, so the pixel
// tiles resident on one XCD are neighbours: the rows above and below that the 3x3 / 4x4 taps reach into are in the SAME L2.  With
// the interleaved order above an XCD holds every eighth pixel tile and shares nothing between its resident tiles (PMC, round 3:
// 213 - 265 MB fetched per launch for 67 MB of activations at the Cout <= 64 full-resolution layers).  No padding workgroups:
// the first (total % 8) XCDs own one tile more.
__device__ __forceinline__ void fsv_xcd_band(int nx, int ny, int& bx, int& by) {
  const int total = nx * ny;
  const int b = blockIdx.x + blockIdx.y * nx;
  const int q = total >> 3, r = total & 7;
  const int x = b & 7;
  const int t = x * q + (x < r ? x : r) + (b >> 3);
  by = t % ny;
  bx = t / ny;
}

template <int BM, int BN, int WM, int WN, int PF = 1, bool AF = false, int DBG = 0, int MODE = 0, bool UP = false>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_kernel(ConvP p) {
  int bx, by;
  if (p.band) fsv_xcd_band(gridDim.x, gridDim.y, bx, by);
  else fsv_xcd_tile(gridDim.x, gridDim.y, bx, by);
  fsv_conv_igemm_body<BM, BN, WM, WN, PF, AF, DBG, MODE, UP>(p, bx, by, (int)blockIdx.z);
}

// Grouped launch: up to FSV_GROUP_MAX INDEPENDENT gather-GEMM problems in one 1-D grid (the problem table travels in the
// kernel arguments: no device-side table, nothing to keep alive, legal inside a graph capture).  What it is for: the step
// holds ~250 launches whose grids cover a fraction of the chip - the 16 weight-generator MLPs (generator.py:103-110,
// 245-273: three small GEMMs each), the four output-parity classes of every stride-2 data gradient - and a launch lasts as
// long as its longest workgroup (prologue + K loop + epilogue, ~10 us even for two K chunks), so 16 of them back to back
// cost 16 critical paths where one grid with all their tiles costs one.  Workgroup b belongs to problem i with
// tile_end[i - 1] <= b < tile_end[i]; inside a problem the tiles are ordered pixel tile fastest, then channel tile, then
// (sample, K split).  Problems with nsplit > 1 add their partial sums atomically into a zeroed output (no epilogue).
struct ConvGroup {
  int nprob;
  int tile_end[FSV_GROUP_MAX];
  ConvP p[FSV_GROUP_MAX];
};

template <int BM, int BN, int WM, int WN, int PF = 1, bool AF = false, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_group_kernel(ConvGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.nprob && b >= g.tile_end[i]) ++i;            // wave-uniform: scalar loads from the kernel arguments
  const int t = b - (i ? g.tile_end[i - 1] : 0);
  const ConvP& p = g.p[i];
  const int gx = (p.Mz + BM - 1) / BM, gy = (p.Cout + BN - 1) / BN;
  const int r = t / gx;
  fsv_conv_igemm_body<BM, BN, WM, WN, PF, AF, 0, MODE>(p, t - r * gx, r % gy, r / gy);
}

// Scalar-gather twin for Cin % 4 != 0 (image / label inputs that were not channel-padded): single LDS buffer, A transposed
// [k][m] with row stride BM + 1.  A handful of small launches per step.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void fsv_conv_igemm_v1_body(const ConvP& p, const int bx, const int by, const int bz) {
  constexpr int V = 1;
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
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bm0 = bx * BM, bn0 = by * BN;
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
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * p.H * p.W * p.Cin * 4);
  const fsv_buf bbuf = fsv_make_buf(wt, (long long)p.nchunks * BK * p.ldw * 4);

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
      areg[i][0] = fsv_buf_load1(abuf, ok ? (unsigned)(((a_base[i] + (long long)iy * p.W + ix) * p.Cin + ci) * 4) : FSV_BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int kr = kc * BK + br0 + i * RPB;
      breg[i] = fsv_buf_load4(bbuf, bcol_ok ? (unsigned)((kr * p.ldw + bcol) * 4) : FSV_BUF_OOB);
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
          atomicAdd(dst, v);                 // (the scalar-gather form keeps the atomic split: the host never arms it)
        } else {
          v = (v + bv) * p.scale;
          if (p.act == FSV_ACT_DLRELU) {
            v = p.res[opix * p.Cout + co] > 0.f ? v : 0.2f * v;
          } else {
            v = fsv_act(v, p.act);
            if (p.res) v += p.res[opix * p.Cout + co];
          }
          *dst = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_v1_kernel(ConvP p) {
  fsv_conv_igemm_v1_body<BM, BN, WM, WN>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// grouped form of the scalar-gather twin (a group that holds a problem with Cin % 4 != 0: the data gradients of the weight
// generators' last layers, whose "input" is the gradient of a 2 * (c + 1)-wide FC output)
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_igemm_v1_group_kernel(ConvGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.nprob && b >= g.tile_end[i]) ++i;
  const int t = b - (i ? g.tile_end[i - 1] : 0);
  const ConvP& p = g.p[i];
  const int gx = (p.Mz + BM - 1) / BM, gy = (p.Cout + BN - 1) / BN;
  const int r = t / gx;
  fsv_conv_igemm_v1_body<BM, BN, WM, WN>(p, t - r * gx, r % gy, r / gy);
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

// ---- finishing pass of an ORDERED split-K launch: out = act((sum_k part[k] + bias) * scale) + res, k ascending -----------------
__global__ __launch_bounds__(256) void fsv_split_finish_kernel(const float* part, long long part_stride, int nsplit, float* out,
                                                               const float* bias, const float* res, long long total, int C,
                                                               long long pix_per_sample, long long b_bstride, int act, float scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    float v = part[i];
    for (int k = 1; k < nsplit; ++k) v += part[(long long)k * part_stride + i];
    if (bias) {
      const int c = (int)(i % C);
      const long long n = b_bstride ? (i / C) / pix_per_sample : 0;
      v += bias[n * b_bstride + c];
    }
    v = fsv_act(v * scale, act);
    if (res) v += res[i];
    out[i] = v;
  }
}

// the same, four consecutive channels per work-item (C % 4 == 0, 16-byte aligned tensors): round 6 - the scalar form walks eight
// elements per work-item one dependent round trip after the other (16 - 21 us for 0.5 M outputs x 8 splits: 0.9 TB/s); here every
// work-item has its nsplit 16-byte loads in flight at once and the launch is one pass (the same sums in the same order)
__global__ __launch_bounds__(256) void fsv_split_finish4_kernel(const float4* part, long long part_stride4, int nsplit, float4* out,
                                                                const float* bias, const float4* res, long long total4, int C,
                                                                long long pix_per_sample, long long b_bstride, int act, float scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total4; i += stride) {
    float4 v = part[i];
#pragma unroll 4
    for (int k = 1; k < nsplit; ++k) {
      const float4 t = part[(long long)k * part_stride4 + i];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (bias) {
      const long long e = i * 4;
      const int c = (int)(e % C);
      const long long n = b_bstride ? (e / C) / pix_per_sample : 0;
      const float* b = bias + n * b_bstride + c;
      v.x += b[0]; v.y += b[1]; v.z += b[2]; v.w += b[3];
    }
    v.x = fsv_act(v.x * scale, act); v.y = fsv_act(v.y * scale, act); v.z = fsv_act(v.z * scale, act); v.w = fsv_act(v.w * scale, act);
    if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    out[i] = v;
  }
}

// the finishing pass that ALSO leaves the per-channel (sum, sum of squares) of the finished output for the normalisation that
// follows (second session of round 6).  A K-split launch had no statistics epilogue - the splits only hold partial outputs - so its
// consumer ran a reduction pass of its own (fsv_red2_kernel<0>: 60 launches of 5 - 6 us per step, one in every conv -> norm chain of
// the 16x16 ... 64x64 encoder levels).  This pass has the finished values in registers anyway: a workgroup takes a 32-pixel x
// 32-channel tile (eight work-items of four channels per pixel row: 128-byte runs), reduces its 32 rows through LDS and adds
// 32 x 2 doubles into the slotted partials of ConvP::stats (slot = pixel tile % slots; same layout, same consumer).  Same output
// bits as fsv_split_finish4_kernel (the same sums in the same order).  Needs C % 32 == 0 and group sizes that are multiples of 32.
__global__ __launch_bounds__(256) void fsv_split_finish4_stats_kernel(const float4* part, long long part_stride4, int nsplit,
                                                                      float4* out, const float* bias, const float4* res,
                                                                      long long npix, int C, int act, float scale, double* stats,
                                                                      int stats_slots, long long stats_ohw) {
  __shared__ float red[32 * 8 * 8];
  const int tid = threadIdx.x, prow = tid >> 3, cq = tid & 7;
  const long long pix = (long long)blockIdx.x * 32 + prow;
  const int c0 = (int)blockIdx.y * 32 + 4 * cq;
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  if (pix < npix) {
    const long long i = (pix * C + c0) >> 2;
    float4 v = part[i];
#pragma unroll 4
    for (int k = 1; k < nsplit; ++k) {
      const float4 t = part[(long long)k * part_stride4 + i];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (bias) { const float* b = bias + c0; v.x += b[0]; v.y += b[1]; v.z += b[2]; v.w += b[3]; }
    v.x = fsv_act(v.x * scale, act); v.y = fsv_act(v.y * scale, act); v.z = fsv_act(v.z * scale, act); v.w = fsv_act(v.w * scale, act);
    if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    out[i] = v;
    s[0] = v.x; s[1] = v.y; s[2] = v.z; s[3] = v.w;
    q[0] = v.x * v.x; q[1] = v.y * v.y; q[2] = v.z * v.z; q[3] = v.w * v.w;
  }
  float* mine = red + (prow * 8 + cq) * 8;
#pragma unroll
  for (int e = 0; e < 4; ++e) { mine[e] = s[e]; mine[4 + e] = q[e]; }
  __syncthreads();
  if (tid < 64) {                       // (channel tid & 31 of the tile, component tid >> 5): the 32 rows in ascending order
    const int c = tid & 31, comp = tid >> 5;
    float a = 0.f;
    for (int r = 0; r < 32; ++r) a += red[(r * 8 + (c >> 2)) * 8 + 4 * comp + (c & 3)];
    const long long g = ((long long)blockIdx.x * 32) / stats_ohw;
    const int slot = (int)(blockIdx.x % (unsigned)stats_slots);
    atomicAdd(stats + ((g * stats_slots + slot) * C + (int)blockIdx.y * 32 + c) * 2 + comp, (double)a);
  }
}

// ---- weight gradient: dwt[z][t*Cin+ci][co] (+)= sum_pixels in[n, oy*sy+ty, ox*sx+tx, ci] * dout[n,oy,ox,co] ----
// All workgroups of one pixel range (blockIdx.z) read the same x pixels (shifted by their taps) and the same dout rows:
// they are mapped onto ONE XCD so that x and dout are fetched from HBM once per range instead of once per L2
// (round 1 PMC: 2.2x the algorithmic bytes on the Cout <= 64 full-resolution layers, which are HBM-bound).
__device__ __forceinline__ void fsv_xcd_range(int& kt, int& nt, int& z) {
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

// V4 kernel: both operands are pixel-major in HBM and in LDS ([32 pixels][columns], ds_write_b128 / ds_read_b32, conflict
// free); two LDS buffers, one barrier per 32-pixel chunk, absent rows / columns are loaded at FSV_BUF_OOB.
// PF: prefetch distance of the global loads in chunks of 32 pixels (see the forward kernel): 1 = one register set, the loads of
// chunk c + 1 issued at the top of chunk c; 2 = two sets, the set stored behind the 12th MFMA was loaded a whole chunk earlier.
// (A form with the loads written straight into LDS - the forward kernel's LD - was built and measured in round 3: 2 ... 15 % SLOWER per
// shape, +0.45 ms on the step: its whole trips of three chunks pad reductions that are split into pieces of 8 - 16 chunks.  Removed.)
template <int BMK, int BN, int WM, int WN, bool COUT4, int PF = 1>
__device__ __forceinline__ void fsv_conv_wgrad_body(const WgradP& p, const int kt, const int nt, const int bz) {
  constexpr int BK = FSV_BK;   // pixels per chunk
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BMK / (WM * 32), TN = BN / (WN * 32);
  constexpr int QA = BMK / 4, RPA = NT / QA, NPA = BK / RPA;
  constexpr int QB = BN / 4, RPB = NT / QB, NPB = BK / RPB;
  constexpr int A_ST = BK * BMK, B_ST = BK * BN;
  static_assert(TM >= 1 && TN >= 1, "tile");
  static_assert(NPA >= 1 && NPB >= 1 && RPA >= 1 && NPA * RPA == BK && NPB * RPB == BK, "tile / thread-count mismatch");
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_ST + B_ST)];
  float* const As = smem;
  float* const Bs = smem + 2 * A_ST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int zs = bz / p.nsplit, zk = bz % p.nsplit;
  const int bi0 = kt * BMK, bn0 = nt * BN;
  float* dwt = p.dwt + (long long)zs * p.w_bstride;
  const int ohw = p.OH * p.OW;

  // A: column (t,ci) handled by this thread is fixed for the whole reduction
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
  const long long dout_base = (long long)zs * (p.per_sample ? p.Mz : 0);
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * (p.H >> p.up) * (p.W >> p.up) * p.Cin * 4);
  const fsv_buf bbuf = fsv_make_buf(p.dout + dout_base * p.Cout, (long long)p.Mz * p.Cout * 4);

  const int cps = (p.pchunks + p.nsplit - 1) / p.nsplit;
  const int c_begin = zk * cps;
  const int c_end = (c_begin + cps < p.pchunks) ? (c_begin + cps) : p.pchunks;

  constexpr int NSET = PF >= 2 ? 2 : 1;
  float4 areg[NSET][NPA], breg[NSET][NPB];
  unsigned aoff[NPA], boff[NPB];
  // Branch-free, incremental addressing (see the forward kernel): every A row of this thread walks the output pixels in
  // steps of 32; (n, oy, ox) are advanced with 32 / OW and 32 % OW and one conditional subtract per level - the host
  // sends geometries with 32 / OW + 1 > OH to the scalar twin, where a step could cross two images.  Offsets are
  // computed one chunk ahead of their loads.
  int a_m[NPA], a_n[NPA], a_oy[NPA], a_ox[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int m = c_begin * BK + apr0 + i * RPA;
    int n, rem;
    if (p.per_sample) { n = zs; rem = m; } else { n = m / ohw; rem = m - n * ohw; }
    a_m[i] = m; a_n[i] = n; a_oy[i] = rem / p.OW; a_ox[i] = rem - a_oy[i] * p.OW;
  }
  const int qw = BK / p.OW, rw = BK - qw * p.OW;
  int b_m = c_begin * BK + bpr0;
  auto calc_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int iy = a_oy[i] * p.sy + ty, ix = a_ox[i] * p.sx + tx;
      const bool ok = kok & (a_m[i] < p.Mz) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      // (p.up = 1: the tensor lies at half the resolution behind a folded nearest x2 up-sampling - two shifts, see ConvP::up)
      aoff[i] = ok ? (unsigned)((((a_n[i] * (p.H >> p.up) + (iy >> p.up)) * (p.W >> p.up) + (ix >> p.up)) * p.Cin + ci) * 4) : FSV_BUF_OOB;
      a_m[i] += BK;
      int ox = a_ox[i] + rw, oy = a_oy[i] + qw;
      const bool cx = ox >= p.OW;
      ox = cx ? ox - p.OW : ox;
      oy = cx ? oy + 1 : oy;
      const bool cy = oy >= p.OH;
      a_oy[i] = cy ? oy - p.OH : oy;
      a_n[i] = cy ? a_n[i] + 1 : a_n[i];
      a_ox[i] = ox;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const int m = b_m + i * RPB;
      boff[i] = ((m < p.Mz) & (bcol < p.Cout)) ? (unsigned)((m * p.Cout + bcol) * 4) : FSV_BUF_OOB;
    }
    b_m += BK;
  };
  auto issue_loads = [&](float4 (&areg)[NPA], float4 (&breg)[NPB]) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) areg[i] = fsv_buf_load4(abuf, aoff[i]);
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      if constexpr (COUT4) {
        breg[i] = fsv_buf_load4(bbuf, boff[i]);
      } else {       // Cout = 1 ... 3 (image / flow / mask heads): element loads, each column range-checked on its own
        const unsigned e = boff[i];
        const bool rok = e != FSV_BUF_OOB;
        breg[i] = make_float4(fsv_buf_load1(bbuf, e),
                              fsv_buf_load1(bbuf, (rok & (bcol + 1 < p.Cout)) ? e + 4 : FSV_BUF_OOB),
                              fsv_buf_load1(bbuf, (rok & (bcol + 2 < p.Cout)) ? e + 8 : FSV_BUF_OOB),
                              fsv_buf_load1(bbuf, (rok & (bcol + 3 < p.Cout)) ? e + 12 : FSV_BUF_OOB));
      }
    }
  };
  auto store_chunk = [&](int buf, const float4 (&areg)[NPA], const float4 (&breg)[NPB]) {
    float* a_dst = As + buf * A_ST;
    float* b_dst = Bs + buf * B_ST;
#pragma unroll
    for (int i = 0; i < NPA; ++i) *reinterpret_cast<float4*>(&a_dst[(apr0 + i * RPA) * BMK + aq * 4]) = areg[i];
#pragma unroll
    for (int i = 0; i < NPB; ++i) *reinterpret_cast<float4*>(&b_dst[(bpr0 + i * RPB) * BN + bq * 4]) = breg[i];
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
  // fragments of four k steps (8 pixels); read one group ahead of its MFMAs, pinned by scheduling fences (see the forward kernel)
  auto read_group = [&](const float* a_src, const float* b_src, int g, float (&fa)[4][TM], float (&fb)[4][TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[s][i] = a_src[a_off + (4 * g + s) * 2 * BMK + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[s][j] = b_src[b_off + (4 * g + s) * 2 * BN + j * 32];
    }
  };
  auto mma_group = [&](const float (&fa)[4][TM], const float (&fb)[4][TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
  };
  // one chunk: the loads issued at its top go into (lar, lbr); (sar, sbr) is the set stored into the other LDS buffer behind three
  // quarters of its MFMAs - the same set for PF = 1, the one loaded a chunk earlier for PF = 2.  One barrier per chunk.  Chunks past
  // c_end are past the descriptors (zeros) or another split's pixels; what the last iterations store is never used.
  auto chunk = [&](int buf, float4 (&lar)[NPA], float4 (&lbr)[NPB], const float4 (&sar)[NPA], const float4 (&sbr)[NPB]) {
    issue_loads(lar, lbr);
    const float* a_src = As + buf * A_ST;
    const float* b_src = Bs + buf * B_ST;
    float fa[2][4][TM], fb[2][4][TN];
    read_group(a_src, b_src, 0, fa[0], fb[0]);
    FSV_SCHED_FENCE();
    calc_offsets();
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
    store_chunk(buf ^ 1, sar, sbr);
    FSV_SCHED_FENCE();
    mma_group(fa[1], fb[1]);
    __syncthreads();
  };
  if (c_begin < c_end) {
    if constexpr (PF == 1) {
      calc_offsets();
      issue_loads(areg[0], breg[0]);
      calc_offsets();
      store_chunk(0, areg[0], breg[0]);
      __syncthreads();
      int buf = 0;
#pragma unroll 1
      for (int pc = c_begin; pc < c_end; ++pc) {
        // (written out instead of calling chunk(): the instruction stream that was validated on hardware, 80 registers)
        issue_loads(areg[0], breg[0]);
        const float* a_src = As + buf * A_ST;
        const float* b_src = Bs + buf * B_ST;
        float fa[2][4][TM], fb[2][4][TN];
        read_group(a_src, b_src, 0, fa[0], fb[0]);
        FSV_SCHED_FENCE();
        calc_offsets();
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
        store_chunk(buf ^ 1, areg[0], breg[0]);
        FSV_SCHED_FENCE();
        mma_group(fa[1], fb[1]);
        __syncthreads();
        buf ^= 1;
      }
    } else {
      calc_offsets();
      issue_loads(areg[0], breg[0]);
      calc_offsets();
      store_chunk(0, areg[0], breg[0]);
      issue_loads(areg[1], breg[1]);
      calc_offsets();
      __syncthreads();
#pragma unroll 1
      for (int pc = c_begin; pc < c_end; pc += 2) {
        chunk(0, areg[0], breg[0], areg[1], breg[1]);
        if (pc + 1 >= c_end) break;
        chunk(1, areg[1], breg[1], areg[0], breg[0]);
      }
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

template <int BMK, int BN, int WM, int WN, bool COUT4, int PF = 1>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_wgrad_kernel(WgradP p) {
  int kt, nt, bz;
  fsv_xcd_range(kt, nt, bz);
  fsv_conv_wgrad_body<BMK, BN, WM, WN, COUT4, PF>(p, kt, nt, bz);
}

// Grouped weight gradients (see fsv_conv_igemm_group_kernel): the problems' tiles in one 1-D grid, inside a problem ordered
// (row tile, column tile) fastest and pixel range slowest, so that consecutive workgroups share a pixel range; every problem
// adds into its own zeroed dwt when its reduction is split.
struct WgradGroup {
  int nprob;
  int tile_end[FSV_GROUP_MAX];
  WgradP p[FSV_GROUP_MAX];
};

template <int BMK, int BN, int WM, int WN, bool COUT4, int PF = 1>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_wgrad_group_kernel(WgradGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.nprob && b >= g.tile_end[i]) ++i;
  const int t = b - (i ? g.tile_end[i - 1] : 0);
  const WgradP& p = g.p[i];
  const int gx = (p.K + BMK - 1) / BMK, gy = (p.Cout + BN - 1) / BN;
  const int r = t / gx;
  fsv_conv_wgrad_body<BMK, BN, WM, WN, COUT4, PF>(p, t - r * gx, r % gy, r / gy);
}

// scalar-gather twin (Cin % 4 != 0), single LDS buffer
template <int BMK, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void fsv_conv_wgrad_v1_kernel(WgradP p) {
  constexpr int V = 1;
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
  const long long dout_base = (long long)zs * (p.per_sample ? p.Mz : 0);
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * p.H * p.W * p.Cin * 4);
  const fsv_buf bbuf = fsv_make_buf(p.dout + dout_base * p.Cout, (long long)p.Mz * p.Cout * 4);
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
      areg[i][0] = fsv_buf_load1(abuf, ok ? (unsigned)(((((long long)n * p.H + iy) * p.W + ix) * p.Cin + ci) * 4) : FSV_BUF_OOB);
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      int m = pc * BK + bpr0 + i * RPB;
      bool rok = m < p.Mz;
      if (cout4) {
        breg[i] = fsv_buf_load4(bbuf, (rok && bcol < p.Cout) ? (unsigned)((m * p.Cout + bcol) * 4) : FSV_BUF_OOB);
      } else {
        const unsigned e = (unsigned)((m * p.Cout + bcol) * 4);
        breg[i] = make_float4(fsv_buf_load1(bbuf, (rok && bcol + 0 < p.Cout) ? e : FSV_BUF_OOB),
                              fsv_buf_load1(bbuf, (rok && bcol + 1 < p.Cout) ? e + 4 : FSV_BUF_OOB),
                              fsv_buf_load1(bbuf, (rok && bcol + 2 < p.Cout) ? e + 8 : FSV_BUF_OOB),
                              fsv_buf_load1(bbuf, (rok && bcol + 3 < p.Cout) ? e + 12 : FSV_BUF_OOB));
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

// ---- thin-output convolutions (Cout <= 4: the image / flow / mask heads, generator.py:126-131, 479-496) ------------------------
// On the matrix cores a Cout = 3 layer pays for a 32-wide tile: 29 of 32 MFMA columns multiply padding (M524288 N3 K288 took 130 us
// forward, 112 us for its weight gradient - 9.7 GFLOP of MFMA work for 0.9 GFLOP of arithmetic).  These two kernels do the 0.9 GFLOP
// on the vector ALUs, memory-bound: one work-item per output pixel, the K dimension walked tap by tap and four channels per 16-byte
// load; taps outside the image contribute exact zeros through the buffer descriptor's out-of-range zero fill.
// Forward: Cin / 4 adjacent lanes share one output pixel (lane = (pixel of the wave, channel quad): a load instruction covers
// whole 128-byte lines - the first version, one work-item per pixel walking its K alone, touched 64 different lines per load and
// was no faster than the padded MFMA tile, profiles/r03_notes.md), the tap weights of a lane's four channels come from an LDS copy
// of the K-major weight columns, and the per-lane partial sums are folded over the pixel's lanes with xor-shuffles (a fixed
// order, but not the ascending-k chain of the MFMA path: these heads feed tanh / sigmoid / a scale, no LeakyReLU kink).
// Host: Cin / 4 a power of two <= 64, K <= FSV_THIN_MAXK.
#define FSV_THIN_MAXK 1152
// T9 (round 6; the three heads of the step are 3x3): the lane's 9 x 4 x CO weights live in REGISTERS for the whole launch (they
// depend on the lane's channel quad only - the loop used to re-read them from LDS for every pixel: 12 ds_reads per 12 fmas) and
// the nine tap loads of a pixel are issued together; the fma chain of every output is the one of the generic loop (bit-equal).
template <int CO, bool T9 = false>
__global__ __launch_bounds__(256) void fsv_conv_thin_fwd_kernel(ConvP p, int iters) {
  __shared__ float wl[FSV_THIN_MAXK * CO];        // [k][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < p.K * CO; i += 256) {
    const int k = i / CO, c = i - k * CO;
    wl[i] = (c < p.Cout) ? p.wt[(long long)k * p.ldw + c] : 0.f;
  }
  __syncthreads();
  const int QC = p.Cin >> 2;
  const int PPW = 64 / QC;                         // pixels per wave
  const int cq = lane % QC, pl = lane / QC;
  const int ohw = p.OH * p.OW;
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * p.H * p.W * p.Cin * 4);
  const float ws = p.wscale ? p.wscale[0] : 1.f;
  const int m_base = blockIdx.x * (4 * PPW * iters);
  float wr[T9 ? 9 : 1][4][CO];
  int tyr[T9 ? 9 : 1], txr[T9 ? 9 : 1];
  if constexpr (T9) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      fsv_tap(p, t, tyr[t], txr[t]);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CO; ++c) wr[t][e][c] = wl[(t * p.Cin + cq * 4 + e) * CO + c];
    }
  }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int m = m_base + (it * 4 + wave) * PPW + pl;
    const bool live = m < p.Mz;
    const int mm = live ? m : 0;
    const int n = mm / ohw, rem = mm - n * ohw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    if constexpr (T9) {
      float4 v9[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy * p.sy + tyr[t], ix = ox * p.sx + txr[t];
        const bool ok = live & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        v9[t] = fsv_buf_load4(abuf, ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : FSV_BUF_OOB);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
          acc[c] = fmaf(v9[t].x, wr[t][0][c], acc[c]);
          acc[c] = fmaf(v9[t].y, wr[t][1][c], acc[c]);
          acc[c] = fmaf(v9[t].z, wr[t][2][c], acc[c]);
          acc[c] = fmaf(v9[t].w, wr[t][3][c], acc[c]);
        }
      }
    } else
#pragma unroll 3
    for (int t = 0; t < p.ntaps; ++t) {
      int ty, tx;
      fsv_tap(p, t, ty, tx);
      const int iy = oy * p.sy + ty, ix = ox * p.sx + tx;
      const bool ok = live & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const float4 v = fsv_buf_load4(abuf, ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.Cin + cq * 4) * 4) : FSV_BUF_OOB);
      const float* w0 = wl + (t * p.Cin + cq * 4) * CO;
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        acc[c] = fmaf(v.x, w0[c], acc[c]);
        acc[c] = fmaf(v.y, w0[CO + c], acc[c]);
        acc[c] = fmaf(v.z, w0[2 * CO + c], acc[c]);
        acc[c] = fmaf(v.w, w0[3 * CO + c], acc[c]);
      }
    }
    // fold the channel quads of a pixel (adjacent lanes)
    for (int o = 1; o < QC; o <<= 1) {
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] += __shfl_xor(acc[c], o);
    }
    if (live && cq == 0) {
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        if (c >= p.Cout) break;
        float v = acc[c] * ws;
        v = (v + (p.bias ? p.bias[c] : 0.f)) * p.scale;
        v = fsv_act(v, p.act);
        if (p.res) v += p.res[(long long)m * p.Cout + c];
        p.out[(long long)m * p.Cout + c] = v;
      }
    }
  }
}

// (A second vector-ALU kernel for SHORT-K layers with a wide output - the first convolutions on 4 ... 16-channel inputs, the data
// gradients of these heads - was built and measured in round 3: bit-equal, but 0.1 ... 0.3 ms slower on the step than the padded
// MFMA tiles at every K limit tried (one LDS weight read per four fmas); removed.  profiles/r03_notes.md.)
// Weight gradient of a thin-output convolution into the K-major layout dwt[(tap, ci)][co] (ZEROED by the caller / the launcher):
// a workgroup owns a run of `chunk` pixels; work-item = (group of 8 * ntaps lanes -> one (tap, ci quad) each ... ) see below.
// Lanes are laid out as tid = g * QK + q: q = tap * (Cin / 4) + ci / 4 walks the K quads (the 8 quads of a pixel-tap are adjacent
// lanes: one 128-byte line), g = 0 .. NG - 1 are pixel sub-streams.  Every work-item keeps 4 * CO partial sums over its pixels;
// the NG sub-streams are folded through LDS and one atomic add per (k, co) and workgroup lands in dwt.
template <int CO>
__global__ __launch_bounds__(256) void fsv_conv_thin_wgrad_kernel(WgradP p, int chunk) {
  __shared__ float red[256 * 4 * CO];
  const int QK = p.ntaps * (p.Cin >> 2);          // K quads (host: QK <= 256)
  const int NG = 256 / QK;                        // pixel sub-streams per workgroup
  const int tid = threadIdx.x;
  const int q = tid % QK, g = tid / QK;
  const bool lane_on = g < NG;
  const int t = q / (p.Cin >> 2), ci = (q - t * (p.Cin >> 2)) * 4;
  int ty, tx;
  {
    unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
    int sh = (t & 7) * 8;
    ty = (int)((code >> sh) & 15ull) - 8;
    tx = (int)((code >> (sh + 4)) & 15ull) - 8;
  }
  const fsv_buf abuf = fsv_make_buf(p.in, (long long)p.N * p.H * p.W * p.Cin * 4);
  const fsv_buf dbuf = fsv_make_buf(p.dout, (long long)p.Mz * p.Cout * 4);
  const int ohw = p.OH * p.OW;
  const int m0 = blockIdx.x * chunk;
  const int m1 = (m0 + chunk < p.Mz) ? m0 + chunk : p.Mz;
  float acc[4][CO];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[e][c] = 0.f;
  if (lane_on) {
    // (round 6: four pixels of loads in flight per work-item instead of two - the loop is one 16-byte load + CO 4-byte loads per
    // pixel and latency-bound; the per-(k, co) sums keep their order)
#pragma unroll 4
    for (int m = m0 + g; m < m1; m += NG) {
      const int n = m / ohw, rem = m - n * ohw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      const int iy = oy * p.sy + ty, ix = ox * p.sx + tx;
      const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
      const float4 v = fsv_buf_load4(abuf, ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.Cin + ci) * 4) : FSV_BUF_OOB);
      float d[CO];
#pragma unroll
      for (int c = 0; c < CO; ++c) d[c] = fsv_buf_load1(dbuf, (c < p.Cout) ? (unsigned)((m * p.Cout + c) * 4) : FSV_BUF_OOB);
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        acc[0][c] = fmaf(v.x, d[c], acc[0][c]);
        acc[1][c] = fmaf(v.y, d[c], acc[1][c]);
        acc[2][c] = fmaf(v.z, d[c], acc[2][c]);
        acc[3][c] = fmaf(v.w, d[c], acc[3][c]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < CO; ++c) red[(e * CO + c) * 256 + tid] = acc[e][c];
  __syncthreads();
  if (g == 0) {            // fold the sub-streams of this K quad in a fixed order, then one atomic per (k, co)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        float s = 0.f;
        for (int gg = 0; gg < NG; ++gg) s += red[(e * CO + c) * 256 + gg * QK + q];
        if (c < p.Cout) atomicAdd(p.dwt + (long long)(t * p.Cin + ci + e) * p.ldw + c, s);
      }
  }
}

// Worth it only where the layer is wide in pixels and short in K (one work-item walks the whole K of its pixel: a 512 -> 1
// discriminator head with K = 8192 over 4900 pixels would be a handful of latency-bound work-items).  FSV_CONV_THIN (read at every
// call: tests switch it): 0 = never, 2 = whenever the layer is eligible (unit tests on small maps), default = the size rule.
static inline bool fsv_conv_thin(int Mz, int K) {
  const char* e = getenv("FSV_CONV_THIN");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '2') return true;
  return K <= 1152 && (long long)Mz >= 64ll * K;
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
// tmap[b] = (first layout of the layer, layouts of the layer, 32-wide co tile, ci tile): the OIHW source tile [32 co][CI_T ci][KH*KW] is read in contiguous runs of
// CI_T * KK floats per output channel, staged in LDS and written out as rows of the K-major layout (CI_T = 32 for <= 8 source
// taps, else 16).  Only the valid region is written: the padding rows / columns of a layout are zero from allocation on.
// The kernel is a pure HBM stream (4 B read + 4 B written per element and layout), so what matters is the number of loads a
// work-item has in flight and the width of its stores: the tile is walked as a flat element sequence with four loads issued
// before the first LDS write, index arithmetic is by compile-time constants for the three source tap counts that hold all but a
// few KB of the parameters (1x1 / linear, 3x3, 4x4; KKT = 0: any other, run-time divisions), and the layouts are written as
// float4 (round 3: 805 -> see profiles/r03_notes.md).
// Round 6: a workgroup stages its OIHW source tile ONCE and writes every layout of the layer from it (forward + data gradient, the
// four parity classes of a stride-2 layer, the summed-tap layouts below): tmap[b] = (first job of the layer, number of jobs, co tile,
// ci tile); the jobs of a layer are consecutive in the tables and share source pointer and source geometry (dims[0..4]).  Before, every
// job re-read and re-staged the source (16 B per weight of HBM traffic and two staging passes for a stride-1 layer: now 12 B and one).
// mode & 4 (summed taps - conv3x3(nearest_x2(x)) without its redundant products): the two nibbles of a tap code are
// MASKS over kh resp. kw and the written value is the sum of the selected source taps, kh ascending outside, kw ascending inside
// (sums of at most four weights: V = R W R^T with 0 / 1 rows R - ops._SUBPIXEL_ROWS / _DGRAD_ROWS).
template <int KKT>
__device__ __forceinline__ void fsv_prep_stage(const PrepGroup& g, const int layer, const int cot, const int cit, float* t) {
  const int* d = g.dims + layer * 9;
  const int Cout = d[0], CinR = d[2], KW = d[4];
  const int KK = KKT ? KKT : d[3] * KW;
  const int CI_T = KK <= 8 ? 32 : 16;
  const int run = CI_T * KK, lds = run + 1;
  const float* w = reinterpret_cast<const float*>(g.src[layer]);
  const int co0 = cot * 32, ci0 = cit * CI_T;
  const int tid = threadIdx.x;
  const int total = 32 * run;
  constexpr int U = 8;          // loads in flight per work-item (4 until round 6: 2.3 TB/s on a pure stream)
  for (int e0 = tid; e0 < total; e0 += 256 * U) {
    float val[U];
    int dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      const int col = e / run, idx = e - col * run;
      const int co = co0 + col, ci = ci0 + idx / KK;
      const bool ok = (e < total) & (co < Cout) & (ci < CinR);
      val[u] = ok ? w[((long long)co * CinR + ci0) * KK + idx] : 0.f;
      dst[u] = (e < total) ? col * lds + idx : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u] >= 0) t[dst[u]] = val[u];
  }
}

// one weight of a layout out of the staged tile: a single source tap, or (masked) the sum of the taps the two masks select
__device__ __forceinline__ float fsv_prep_pick(const float* src, const int code, const int KW, const bool masked) {
  if (!masked) return src[(code & 15) * KW + (code >> 4)];
  const int mh = code & 15, mw = code >> 4;
  float s = 0.f;
  bool first = true;
  for (int kh = 0; kh < 4; ++kh) {
    if (!((mh >> kh) & 1)) continue;
    for (int kw = 0; kw < 4; ++kw) {
      if (!((mw >> kw) & 1)) continue;
      const float v = src[kh * KW + kw];
      s = first ? v : s + v;
      first = false;
    }
  }
  return s;
}

template <int KKT>
__device__ __forceinline__ void fsv_prep_emit(const PrepGroup& g, const int layer, const int cot, const int cit, const float* t) {
  const int* d = g.dims + layer * 9;
  const int Cout = d[0], CinP = d[1], KW = d[4], ntaps = d[5], ldw = d[7];
  const int mode = d[8] & 1;
  const bool masked = (d[8] & 4) != 0;
  const int KK = KKT ? KKT : d[3] * KW;
  const int CI_T = KK <= 8 ? 32 : 16;
  const int run = CI_T * KK, lds = run + 1;
  float* wt = reinterpret_cast<float*>(g.dst[layer]);
  const unsigned long long lo = g.taps[layer * 2], hi = g.taps[layer * 2 + 1];
  const int co0 = cot * 32, ci0 = cit * CI_T;
  const int tid = threadIdx.x;
  if (mode == 0) {          // rows (tap j, ci), 32 consecutive output channels each: 8 work-items x float4 per row
    const int q = tid & 7, r0 = tid >> 3;
    const int nrows = ntaps * CI_T;
    for (int r = r0; r < nrows; r += 32) {
      const int j = r / CI_T, cil = r - j * CI_T;
      const int ci = ci0 + cil;
      if (ci >= CinP) continue;
      const unsigned long long code8 = (j < 8) ? lo : hi;
      const int code = (int)((code8 >> ((j & 7) * 8)) & 255ull);
      const float* src = t + (4 * q) * lds + cil * KK;
      // columns at or beyond Cout hold zeros in LDS and land in the layout's zero padding (ldw is a multiple of 32)
      *reinterpret_cast<float4*>(&wt[((long long)j * CinP + ci) * ldw + co0 + 4 * q]) =
          make_float4(fsv_prep_pick(src, code, KW, masked), fsv_prep_pick(src + lds, code, KW, masked),
                      fsv_prep_pick(src + 2 * lds, code, KW, masked), fsv_prep_pick(src + 3 * lds, code, KW, masked));
    }
  } else {                  // rows (tap j, co), CI_T consecutive input channels each: CI_T / 4 work-items x float4 per row
    const int QN = CI_T / 4;
    const int q = tid % QN, r0 = tid / QN, rstep = 256 / QN;
    const int nrows = ntaps * 32;
    for (int r = r0; r < nrows; r += rstep) {
      const int j = r >> 5, col = r & 31;
      const int co = co0 + col, ci = ci0 + 4 * q;
      if (co >= Cout || ci >= CinP) continue;
      const unsigned long long code8 = (j < 8) ? lo : hi;
      const int code = (int)((code8 >> ((j & 7) * 8)) & 255ull);
      const float* src = t + col * lds + (4 * q) * KK;
      float* dstp = &wt[((long long)j * Cout + co) * ldw + ci];
      if (ci + 3 < CinP) {
        *reinterpret_cast<float4*>(dstp) =
            make_float4(fsv_prep_pick(src, code, KW, masked), fsv_prep_pick(src + KK, code, KW, masked),
                        fsv_prep_pick(src + 2 * KK, code, KW, masked), fsv_prep_pick(src + 3 * KK, code, KW, masked));
      } else {
        for (int k = 0; ci + k < CinP; ++k) dstp[k] = fsv_prep_pick(src + k * KK, code, KW, masked);
      }
    }
  }
}

template <int KKT>
__device__ __forceinline__ void fsv_prep_layer(const PrepGroup& g, const int first, const int njobs, const int cot, const int cit,
                                               float* t) {
  fsv_prep_stage<KKT>(g, first, cot, cit, t);
  __syncthreads();
  for (int j = 0; j < njobs; ++j) fsv_prep_emit<KKT>(g, first + j, cot, cit, t);
}

__global__ __launch_bounds__(256) void fsv_prep_group_kernel(PrepGroup g, const int* tmap) {
  __shared__ float t[32 * 257];
  const int first = tmap[blockIdx.x * 4], njobs = tmap[blockIdx.x * 4 + 1];
  const int cot = tmap[blockIdx.x * 4 + 2], cit = tmap[blockIdx.x * 4 + 3];
  const int KK = g.dims[first * 9 + 3] * g.dims[first * 9 + 4];
  if (KK == 9) fsv_prep_layer<9>(g, first, njobs, cot, cit, t);
  else if (KK == 1) fsv_prep_layer<1>(g, first, njobs, cot, cit, t);
  else if (KK == 16) fsv_prep_layer<16>(g, first, njobs, cot, cit, t);
  else fsv_prep_layer<0>(g, first, njobs, cot, cit, t);
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

// tile ids: 0 = 128x128, 1 = 128x64, 2 = 128x32, 4 = 64x64, 9 = 64x128 (BM pixels x BN output channels).  0 / 1 / 9 are
// 8-wave workgroups (two waves per SIMD cover each other's LDS latency and barrier: +8 ... +14 % in-box over the same
// tiles with 4 waves, profiles/r02_tile_ab.jsonl), 2 / 4 are 4-wave workgroups.
static inline int fsv_tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case 0: bm = 128; bn = 128; return 0;
    case 1: bm = 128; bn = 64; return 0;
    case 2: bm = 128; bn = 32; return 0;
    case 4: bm = 64; bn = 64; return 0;
    case 9: bm = 64; bn = 128; return 0;
    // experimental, force_tile only (never chosen by fsv_conv_plan): the 8-wave tiles with a prefetch distance of two chunks
    case 10: bm = 64; bn = 128; return 0;
    case 11: bm = 128; bn = 128; return 0;
    case 12: bm = 128; bn = 64; return 0;
    // 10 - 12 with the A fragments read in place (ds_read_b64 of a re-ordered quad: no select between the MFMAs; PF = 3)
    case 13: bm = 64; bn = 128; return 0;
    case 14: bm = 128; bn = 128; return 0;
    case 15: bm = 128; bn = 64; return 0;
    // the tiles that keep their global loads one chunk ahead, with in-place A fragments
    case 16: bm = 128; bn = 128; return 0;
    case 17: bm = 64; bn = 64; return 0;
    case 18: bm = 128; bn = 32; return 0;
    // 64x64 with a prefetch distance of two chunks + in-place A fragments
    case 20: bm = 64; bn = 64; return 0;
    // global loads straight into LDS (three buffers); the same form of the 128x128 and 128x32 tiles lost (96 / 60 KB of LDS) and was removed
    case 21: bm = 64; bn = 128; return 0;
    case 22: bm = 128; bn = 64; return 0;
    case 27: bm = 64; bn = 64; return 0;
#ifdef FSV_DIAG
    case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: bm = 64; bn = 128; return 0;
#endif
    default: return -1;
  }
}

// Kernel variant a tile SHAPE of the plan (0 = 128x128, 1 = 128x64, 2 = 128x32, 4 = 64x64, 9 = 64x128) runs as.  Variants differ in
// the prefetch distance of the global loads (PF: 1 or 2 chunks) and the A-fragment format (AF: b128 quad + per-MFMA select, or
// re-ordered quad read in place with ds_read_b64); all are bit-equal.  Defaults = the in-box A/B of round 3
// (profiles/r03_notes.md, tools/tile_ab.py): 64x128 and 128x64 -> PF 2 + AF (ids 13 / 15: +6 ... +8 % over PF 2 alone, which
// round 2 measured at +5 % over the base tile); 128x128 -> PF 1 + AF (id 16: its PF 2 forms lose occupancy - 168 registers - on
// the multi-workgroup grids it is picked for); 64x64 -> PF 2 + AF (id 20): per shape with warm caches it equals PF 1 + AF (id 17:
// 78.9 vs 81.4 ... 86.4 vs 85.7 TFLOP/s), inside the step - every layer's weights cold - it is worth 0.3 ms (49.19 vs 49.48, two
// in-box pairs); the same form of the 128x32 tile loses 0.2 ms there and was removed.  64x128 -> loads straight into LDS (id 21, LD in
// the kernel's header comment): +3 ... +6 % per shape over id 13 (111 - 121 TFLOP/s), -0.17 ... -0.25 ms on the step (three in-box
// triples); the LD forms of 128x64 (22) and 64x64 (27) gain per shape (+4 %, +17 % on M2048 N512 K2304) and nothing inside the step:
// reachable, not default.  FSV_CONV_V<shape>=<id> overrides one shape (A/B runs).
static inline int fsv_conv_variant(int shape) {
  static int map[10] = {-2, -2, -2, -2, -2, -2, -2, -2, -2, -2};
  if (shape < 0 || shape > 9) return shape;
  if (map[shape] == -2) {
    static const int dflt[10] = {16, 15, 18, -1, 20, -1, -1, -1, -1, 21};
    char name[16];
    snprintf(name, sizeof(name), "FSV_CONV_V%d", shape);
    const char* e = getenv(name);
    map[shape] = e ? atoi(e) : dflt[shape];
    if (map[shape] < 0) map[shape] = shape;
  }
  return map[shape];
}

static int fsv_launch_conv(const ConvP& p, bool vec4, int nz, hipStream_t stream, int tile) {
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return FSV_ERR_BAD_ARG;
  dim3 g(fsv_cdiv(p.Mz, bm), fsv_cdiv(p.Cout, bn), nz);
  if (p.up) {
    // the folded up-sampling exists for the float4 gather, as the variants the plan's five tile shapes run as
    if (!vec4) return FSV_ERR_UNSUPPORTED;
    switch (tile) {
      case 0: case 11: case 14: case 16: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 4, 1, true, 0, 0, true>), g, dim3(512), stream, p); break;
      case 1: case 12: case 15: case 22: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 4, 2, 2, true, 0, 0, true>), g, dim3(512), stream, p); break;
      case 2: case 18: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 32, 4, 1, 1, true, 0, 0, true>), g, dim3(256), stream, p); break;
      case 4: case 17: case 20: case 27: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2, 2, true, 0, 0, true>), g, dim3(256), stream, p); break;
      case 9: case 10: case 13: case 21: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 0, 2, true>), g, dim3(512), stream, p); break;
      default: return FSV_ERR_BAD_ARG;
    }
    return fsv_check_launch();
  }
  if (vec4) {
    switch (tile) {
      case 0: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 4>), g, dim3(512), stream, p); break;
      case 1: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 4, 2>), g, dim3(512), stream, p); break;
      case 2: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 32, 4, 1>), g, dim3(256), stream, p); break;
      case 4: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2>), g, dim3(256), stream, p); break;
      case 10: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2>), g, dim3(512), stream, p); break;
      case 11: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 4, 2>), g, dim3(512), stream, p); break;
      case 12: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 4, 2, 2>), g, dim3(512), stream, p); break;
      case 13: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true>), g, dim3(512), stream, p); break;
      case 14: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 4, 2, true>), g, dim3(512), stream, p); break;
      case 15: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 4, 2, 2, true>), g, dim3(512), stream, p); break;
      case 16: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 128, 2, 4, 1, true>), g, dim3(512), stream, p); break;
      case 17: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2, 1, true>), g, dim3(256), stream, p); break;
      case 18: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 32, 4, 1, 1, true>), g, dim3(256), stream, p); break;
      case 20: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2, 2, true>), g, dim3(256), stream, p); break;
      case 21: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 0, 2>), g, dim3(512), stream, p); break;
      case 22: FSV_LAUNCH((fsv_conv_igemm_kernel<128, 64, 4, 2, 2, true, 0, 2>), g, dim3(512), stream, p); break;
      case 27: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 64, 2, 2, 2, true, 0, 2>), g, dim3(256), stream, p); break;
#ifdef FSV_DIAG
      // knock-out forms of the dominant kernel (tools/knockout.py; results are wrong by construction, only the time is read)
      case 30: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 1>), g, dim3(512), stream, p); break;
      case 31: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 2>), g, dim3(512), stream, p); break;
      case 32: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 4>), g, dim3(512), stream, p); break;
      case 33: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 8>), g, dim3(512), stream, p); break;
      case 34: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 16>), g, dim3(512), stream, p); break;
      case 35: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 30>), g, dim3(512), stream, p); break;
      case 36: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 31>), g, dim3(512), stream, p); break;
      case 37: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4, 2, true, 32>), g, dim3(512), stream, p); break;
#endif
      default: FSV_LAUNCH((fsv_conv_igemm_kernel<64, 128, 2, 4>), g, dim3(512), stream, p); break;
    }
  } else {
    if (tile == 10 || tile == 13 || tile == 21) tile = 9; else if (tile == 11 || tile == 14 || tile == 16) tile = 0; else if (tile == 12 || tile == 15 || tile == 22) tile = 1;     // scalar gather: no variants
    else if (tile == 17 || tile == 20 || tile == 27) tile = 4; else if (tile == 18) tile = 2;
    switch (tile) {
      case 0: FSV_LAUNCH((fsv_conv_igemm_v1_kernel<128, 128, 2, 2>), g, dim3(256), stream, p); break;
      case 1: FSV_LAUNCH((fsv_conv_igemm_v1_kernel<128, 64, 2, 2>), g, dim3(256), stream, p); break;
      case 2: FSV_LAUNCH((fsv_conv_igemm_v1_kernel<128, 32, 4, 1>), g, dim3(256), stream, p); break;
      case 4: FSV_LAUNCH((fsv_conv_igemm_v1_kernel<64, 64, 2, 2>), g, dim3(256), stream, p); break;
      default: FSV_LAUNCH((fsv_conv_igemm_v1_kernel<64, 128, 2, 2>), g, dim3(256), stream, p); break;
    }
  }
  return fsv_check_launch();
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

// Tile / split-K plan shared by the launcher and (through the C ABI) by the host-side profiler labels.
// tile ids: 0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 256x32, 4 = 64x64 (BM x BN, pixels x output channels).
// Predicted duration (seconds) of one gather-GEMM launch with tile `tile` and `nsplit` K splits - a small cost model
// calibrated on the in-box A/B tables (tools/tile_ab.py, profiles/r02_tile_ab.jsonl: mean error 8 %, picks within 4 % of
// the best measured configuration on every shape):
//   * the launch takes as long as its busiest CU: L = ceil(workgroups / 256) workgroups share one CU's matrix pipes - a
//     launch of 276 workgroups costs as much as one of 512, which is what made the discriminator's 33x33 layers run at
//     half speed under a rule that only looked at "at least 256 workgroups";
//   * one K chunk of a BM x BN tile is BM * BN / 4 matrix-pipe cycles; a lone 4-wave workgroup reaches ~78 % of that
//     rate (LDS latency and the barrier are exposed), two waves per SIMD ~90 %, several workgroups per CU ~95 %;
//   * prologue + epilogue per workgroup (partly hidden when other workgroups are co-resident), a launch floor, and for
//     split-K the zero fill, the atomics and the finishing pass.
static inline double fsv_conv_cost(int Mz, int Cout, int nchunks, int nsamp, int tile, int nsplit) {
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return 1e30;
  const bool w8 = (tile == 0 || tile == 1 || tile == 9 || tile >= 10);
  const double wgs = (double)fsv_cdiv(Mz, bm) * fsv_cdiv(Cout, bn) * nsamp * nsplit;
  const double L = (double)((long long)((wgs + 255.0) / 256.0));
  const double cps = (double)fsv_cdiv(nchunks, nsplit);
  const double cyc = (double)bm * bn / 4.0;
  const double eff = w8 ? (L <= 1.0 ? 0.91 : 0.97) : (L <= 1.0 ? 0.78 : (L <= 2.0 ? 0.88 : 0.95));
  const double ovh = (5000.0 + bm * bn / 8.0) * (L <= 1.0 ? 1.0 : 0.45);
  double t = L * (cps * cyc / eff + ovh) / 1.95e9 + 4e-6;
  // (FSV_SPLIT_FIX_US / FSV_SPLIT_BW_TBS: in-box A/B of the split's fixed cost and of the rate of its copies + finishing pass)
  static double split_fix = -1.0, split_bw = 0.0;
  if (split_fix < 0.0) {
    const char* a = getenv("FSV_SPLIT_FIX_US");
    const char* b = getenv("FSV_SPLIT_BW_TBS");
    // round 6: 3 us + 4.5 TB/s (6 us + 3.0 TB/s until the finishing pass became one vectorised pass: 15.6 -> 5.8 us per launch);
    // in-box: 42.86 -> 42.77 ms per step, the M2048 N512 K2304 layers now split in two (79.6 -> 94.0 TFLOP/s in isolation)
    split_bw = (b ? atof(b) : 4.5) * 1e12;
    split_fix = (a ? atof(a) : 3.0) * 1e-6;
  }
  if (nsplit > 1) t += split_fix + 0.5e-6 * nsplit + (double)Mz * Cout * nsamp * 4.0 * (2.0 + 0.25 * nsplit) / split_bw;
  return t;
}

// FSV_DETERMINISTIC=1 (read at every call: tests switch it at run time): no reduction is split across workgroups, so no sum
// depends on the arrival order of atomic adds - every gradient is a fixed-order fp32 sum.  For parity runs: exact cancellations
// that sit ON a LeakyReLU kink otherwise take the sign the atomics' order gives them (profiles/r02_notes.md section 11: one
// weight gradient of the C1 step 3.6e-2 off in one run of eight).  Slower (small grids), never the measured configuration.
static inline bool fsv_deterministic() {
  const char* e = getenv("FSV_DETERMINISTIC");
  return e && e[0] == '1';
}

// size rule of the thin-output (vector-ALU) kernels, exported so that host-side profilers label those launches as what they
// are; returns 1 / 0, not a status
extern "C" int fsv_conv_thin_rule(int Mz, int K) { return fsv_conv_thin(Mz, K) ? 1 : 0; }

extern "C" int fsv_conv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split,
                             int* tile_out, int* nsplit_out) {
  if (force_split <= 0 && fsv_deterministic()) force_split = 1;
  int tile = force_tile, nsplit = force_split > 0 ? force_split : 1;
  if (tile < 0 || force_split <= 0) {
    // candidates: every tile that is not wider than the layer needs (a forced tile: only that one) x split factors that
    // leave at least 8 chunks (256 K-elements) per split; ties go to the larger tile (less operand traffic per FLOP)
    static const int tiles[5] = {0, 9, 1, 4, 2};
    static const int splits[6] = {1, 2, 3, 4, 6, 8};
    double best = 1e30;
    int bt = -1, bs = 1;
    for (int ti = 0; ti < 5; ++ti) {
      const int t = tiles[ti];
      if (force_tile >= 0 && t != force_tile) continue;
      if (force_tile < 0) {
        if ((t == 0 || t == 9) && Cout <= 64) continue;
        if (t == 1 && Cout <= 32) continue;
        if (t == 2 && Cout > 32) continue;
      }
      for (int si = 0; si < 6; ++si) {
        const int sp = splits[si];
        if (force_split > 0 && sp != 1) break;
        const int use = force_split > 0 ? force_split : sp;
        if (use > 1 && nchunks / use < 8) break;
        const double c = fsv_conv_cost(Mz, Cout, nchunks, nsamp, t, use) * (1.0 + 0.01 * ti);
        if (c < best) { best = c; bt = t; bs = use; }
      }
    }
    if (bt < 0) { bt = force_tile >= 0 ? force_tile : (Cout <= 32 ? 2 : 4); bs = force_split > 0 ? force_split : 1; }
    tile = bt; nsplit = bs;
  }
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return -1;
  if (nsplit > nchunks) nsplit = nchunks;
  if (nsplit < 1) nsplit = 1;
  *tile_out = tile; *nsplit_out = nsplit;
  return 0;
}

// XCD order of the single-problem launches: 1 = bands (fsv_xcd_band), 0 = interleaved (fsv_xcd_tile); FSV_CONV_BAND: in-box A/B
static inline int fsv_conv_band() {
  static int v = -1;
  // (measured in-box, profiles/r04_notes.md section 9: neutral on both bench steps - the interleaved order stays the default)
  if (v < 0) { const char* e = getenv("FSV_CONV_BAND"); v = e ? (atoi(e) != 0) : 0; }
  return v;
}

static inline void fsv_fill_convp(ConvP& p, const float* in, const float* wt, const float* bias, const float* res, float* out,
                                  const float* wscale, int N, int H, int W, int Cin, int OH, int OW, int Cout, int ntaps,
                                  const int* ty, const int* tx, int sy, int sx, int outH, int outW, int osy, int osx, int ooy,
                                  int oox, int ldw, long long w_bstride, long long b_bstride, int per_sample, int act,
                                  float scale) {
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
  p.nsplit = 1;
  p.stats = nullptr; p.stats_slots = 1; p.stats_ohw = 1;
  p.part = nullptr; p.part_stride = 0;
  p.band = fsv_conv_band(); p.up = 0;
  {
    const long long obytes = (long long)N * outH * outW * Cout * 4;
    p.res_bytes = (res && obytes <= FSV_BUF_MAX_BYTES) ? obytes : 0;
  }
}

static inline bool vec4_ok(int cin) { return (cin & 3) == 0; }

// Generic gather-GEMM (see header comment and include/fsv2v.h: fsv_conv_gather_fwd).  stats / stats_groups / stats_slots:
// optional statistics partials for the normalisation that follows (ConvP::stats); *produced tells whether this launch wrote
// them (not when the plan splits K or the layer takes the scalar-gather kernel: the caller then runs its reduction pass).
static int fsv_conv_gather_impl(const float* in, const float* wt, const float* bias, const float* res, float* out,
                        int N, int H, int W, int Cin, int OH, int OW, int Cout,
                        int ntaps, const int* ty, const int* tx, int sy, int sx,
                        int outH, int outW, int osy, int osx, int ooy, int oox,
                        int ldw, long long w_bstride, long long b_bstride, int per_sample,
                        int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                        double* stats, int stats_groups, int stats_slots, int stats_prezeroed, int* produced,
                        float* split_ws, long long split_cap, int in_up, hipStream_t stream) {
  if (!in || !wt || !out || ntaps < 1 || ntaps > 16 || N < 1 || Cin < 1 || Cout < 1) return FSV_ERR_BAD_ARG;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((ldw & 3) != 0 || ldw < Cout) return FSV_ERR_BAD_ARG;
  // in_up: the input is stored at H / 2 x W / 2 and read through the nearest x2 index (ConvP::up); float4 gather, MFMA tiles only
  if (in_up && ((H & 1) || (W & 1) || (Cin % 4 != 0) || per_sample || accumulate || Cout <= 4)) return FSV_ERR_UNSUPPORTED;
  // the V4 kernels index one tensor / one weight matrix with 32-bit element offsets
  if ((long long)N * H * W * Cin * 4 > FSV_BUF_MAX_BYTES || (long long)(ntaps * Cin + 32) * ldw * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;
  ConvP p;
  fsv_fill_convp(p, in, wt, bias, res, out, wscale, N, H, W, Cin, OH, OW, Cout, ntaps, ty, tx, sy, sx, outH, outW, osy, osx, ooy,
                 oox, ldw, w_bstride, b_bstride, per_sample, act, scale);
  p.up = in_up ? 1 : 0;
  const int nsamp = per_sample ? N : 1;
  int tile = 0, nsplit = 1;
  // thin-output layers (image / flow / mask heads) run on the vector ALUs: see fsv_conv_thin_fwd_kernel
  if (Cout <= 4 && (Cin % 4 == 0) && Cin <= 256 && ((Cin >> 2) & ((Cin >> 2) - 1)) == 0 && p.K <= FSV_THIN_MAXK && !per_sample &&
      p.dense_out && !accumulate && force_tile < 0 && force_split <= 0 && !stats &&
      act != FSV_ACT_DLRELU && fsv_conv_thin(p.Mz, p.K)) {
    p.nsplit = 1;
    if (produced) *produced = 0;
    const int ppb = 4 * (64 / (Cin >> 2));                  // pixels per workgroup and iteration
    int iters = 16;
    while (iters > 1 && fsv_cdiv(p.Mz, ppb * iters) < 1024) iters >>= 1;
    const dim3 g(fsv_cdiv(p.Mz, ppb * iters));
    // measured neutral on the step (profiles/r06_step_ab_kernel_tweaks.txt: 42.81 ms with, 42.74 without - the heads are bound by
    // their L2 gathers, not by the LDS weight reads): opt-in, FSV_THIN_T9=1
    const char* t9e = getenv("FSV_THIN_T9");
    const bool t9 = p.ntaps == 9 && t9e && t9e[0] == '1';
    switch (Cout) {
      case 1: if (t9) FSV_LAUNCH((fsv_conv_thin_fwd_kernel<1, true>), g, dim3(256), stream, p, iters);
              else FSV_LAUNCH((fsv_conv_thin_fwd_kernel<1>), g, dim3(256), stream, p, iters); break;
      case 2: if (t9) FSV_LAUNCH((fsv_conv_thin_fwd_kernel<2, true>), g, dim3(256), stream, p, iters);
              else FSV_LAUNCH((fsv_conv_thin_fwd_kernel<2>), g, dim3(256), stream, p, iters); break;
      case 3: if (t9) FSV_LAUNCH((fsv_conv_thin_fwd_kernel<3, true>), g, dim3(256), stream, p, iters);
              else FSV_LAUNCH((fsv_conv_thin_fwd_kernel<3>), g, dim3(256), stream, p, iters); break;
      default: if (t9) FSV_LAUNCH((fsv_conv_thin_fwd_kernel<4, true>), g, dim3(256), stream, p, iters);
               else FSV_LAUNCH((fsv_conv_thin_fwd_kernel<4>), g, dim3(256), stream, p, iters); break;
    }
    return fsv_check_launch();
  }
  if (act == FSV_ACT_DLRELU) {       // epilogue-only form: the finishing pass of a split launch does not know it
    if (!res || accumulate) return FSV_ERR_BAD_ARG;
    force_split = 1;
  }
  if (fsv_conv_plan(p.Mz, Cout, p.nchunks, nsamp, force_tile, force_split, &tile, &nsplit)) return FSV_ERR_BAD_ARG;
  p.nsplit = nsplit;
  const long long total = (long long)N * outH * outW * Cout;
  // accumulate != 0: `out` was zeroed by the caller and partial results are added atomically (used by the
  // four parity-class launches of a stride-2 data gradient); bias/act/res are not applied in that mode.
  if (accumulate) {
    if (bias || res || act != FSV_ACT_NONE || scale != 1.f) return FSV_ERR_BAD_ARG;
    // nsplit == 1: every output pixel belongs to exactly one parity class and one tile -> plain stores
  } else if (nsplit > 1) {
    if (!p.dense_out) return FSV_ERR_UNSUPPORTED;
    if (split_ws && (Cin % 4 == 0) && split_cap >= (long long)nsplit * total) {
      p.part = split_ws; p.part_stride = total;         // ordered: one copy of the output per split, summed by the finishing pass
    } else {
      (void)hipMemsetAsync(out, 0, (size_t)total * sizeof(float), stream);
    }
  }
  const bool vec4 = (Cin % 4 == 0);
  if (produced) *produced = 0;
  if (stats && vec4 && nsplit == 1 && !accumulate && !per_sample && p.dense_out && stats_groups >= 1 && stats_slots >= 1 &&
      p.Mz % stats_groups == 0 && (stats_groups == 1 || p.Mz / stats_groups >= 128)) {      // a pixel tile (<= 128 rows) touches
                                                                                              // at most two groups
    p.stats = stats; p.stats_slots = stats_slots; p.stats_ohw = p.Mz / stats_groups;
    if (!stats_prezeroed)
      (void)hipMemsetAsync(stats, 0, (size_t)stats_groups * stats_slots * Cout * 2 * sizeof(double), stream);
    if (produced) *produced = 1;
  }
  // ... or from the finishing pass of an ordered K-split launch (fsv_split_finish4_stats_kernel)
  double* fin_stats = nullptr;
  {
    const char* fse = getenv("FSV_SPLIT_FIN_STATS");       // =0: the consumer's own reduction pass (in-box A/B; read at every call)
    const char* f4e = getenv("FSV_SPLIT_FIN4");
    if (stats && !(fse && fse[0] == '0') && !(f4e && f4e[0] == '0') && nsplit > 1 && p.part && !accumulate && !per_sample && p.dense_out &&
        stats_groups >= 1 && stats_slots >= 1 && (Cout % 32) == 0 && p.Mz % stats_groups == 0 && ((p.Mz / stats_groups) % 32) == 0 &&
        act != FSV_ACT_DLRELU && !fsv_deterministic() &&
        (((unsigned long long)p.part | (unsigned long long)out | (unsigned long long)res) & 15ull) == 0) {
      fin_stats = stats;
      if (!stats_prezeroed)
        (void)hipMemsetAsync(stats, 0, (size_t)stats_groups * stats_slots * Cout * 2 * sizeof(double), stream);
      if (produced) *produced = 1;
    }
  }
  // the plan's 8-wave tiles run as their prefetch-distance-2 variants (a forced tile id is taken literally)
  if (force_tile < 0 && vec4) tile = fsv_conv_variant(tile);
  int rc = fsv_launch_conv(p, vec4, nsamp * nsplit, stream, tile);
  if (rc) return rc;
  if (!accumulate && nsplit > 1 && (p.part || bias || res || act != FSV_ACT_NONE || scale != 1.f)) {
    int grid = (int)((total + 256 * 8 - 1) / (256 * 8));
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    const char* f4e = getenv("FSV_SPLIT_FIN4");          // =0: the scalar finishing pass (A/B, bit-equality test)
    const bool fin4 = !(f4e && f4e[0] == '0') && p.part && (Cout % 4 == 0) && (total % 4 == 0) &&
                      (((unsigned long long)p.part | (unsigned long long)out | (unsigned long long)res) & 15ull) == 0;
    if (fin_stats) {
      const long long npix = (long long)N * outH * outW;
      FSV_LAUNCH(fsv_split_finish4_stats_kernel, dim3((unsigned)((npix + 31) / 32), (unsigned)(Cout / 32)), dim3(256), stream,
                 (const float4*)p.part, p.part_stride / 4, nsplit, (float4*)out, bias, (const float4*)res, npix, Cout, act, scale,
                 fin_stats, stats_slots, (long long)(p.Mz / stats_groups));
    } else if (fin4) {
      int g4 = (int)((total / 4 + 255) / 256);
      if (g4 > 8192) g4 = 8192;
      FSV_LAUNCH(fsv_split_finish4_kernel, dim3(g4), dim3(256), stream, (const float4*)p.part, p.part_stride / 4, nsplit, (float4*)out,
                 bias, (const float4*)res, total / 4, Cout, (long long)outH * outW, per_sample ? b_bstride : 0ll, act, scale);
    } else if (p.part)
      FSV_LAUNCH(fsv_split_finish_kernel, dim3(grid), dim3(256), stream, (const float*)p.part, p.part_stride, nsplit, out, bias, res,
                 total, Cout, (long long)outH * outW, per_sample ? b_bstride : 0ll, act, scale);
    else
      FSV_LAUNCH(fsv_bias_act_kernel, dim3(grid), dim3(256), stream, out, bias, res, total, Cout,
                 (long long)outH * outW, per_sample ? b_bstride : 0ll, act, scale);
    rc = fsv_check_launch();
  }
  return rc;
}

extern "C" {

// split_ws / split_ws_floats (nullable): ordered split-K - when the plan splits K, split k stores its partial output into the k-th
// copy inside split_ws (nsplit x N*outH*outW*Cout floats) and a finishing pass sums the copies in ascending order (the same bits on
// every run, no zero fill, no atomics); a call that does not split, or whose copies do not fit, ignores the workspace.
int fsv_conv_gather_fwd(const float* in, const float* wt, const float* bias, const float* res, float* out,
                        int N, int H, int W, int Cin, int OH, int OW, int Cout,
                        int ntaps, const int* ty, const int* tx, int sy, int sx,
                        int outH, int outW, int osy, int osx, int ooy, int oox,
                        int ldw, long long w_bstride, long long b_bstride, int per_sample,
                        int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                        float* split_ws, long long split_ws_floats, int in_up, hipStream_t stream) {
  return fsv_conv_gather_impl(in, wt, bias, res, out, N, H, W, Cin, OH, OW, Cout, ntaps, ty, tx, sy, sx, outH, outW, osy, osx,
                              ooy, oox, ldw, w_bstride, b_bstride, per_sample, act, scale, force_tile, force_split, accumulate,
                              wscale, nullptr, 0, 0, 0, nullptr, split_ws, split_ws ? split_ws_floats : 0, in_up, stream);
}

int fsv_conv_gather_fwd_stats(const float* in, const float* wt, const float* bias, const float* res, float* out,
                              int N, int H, int W, int Cin, int OH, int OW, int Cout,
                              int ntaps, const int* ty, const int* tx, int sy, int sx,
                              int ldw, int act, float scale, const float* wscale,
                              double* stats, int stats_groups, int stats_slots, int stats_prezeroed, int* produced,
                              float* split_ws, long long split_ws_floats, int in_up, hipStream_t stream) {
  if (!stats || !produced) return FSV_ERR_BAD_ARG;
  return fsv_conv_gather_impl(in, wt, bias, res, out, N, H, W, Cin, OH, OW, Cout, ntaps, ty, tx, sy, sx, OH, OW, 1, 1, 0, 0, ldw,
                              0, 0, 0, act, scale, -1, 0, 0, wscale, stats, stats_groups, stats_slots, stats_prezeroed, produced,
                              split_ws, split_ws ? split_ws_floats : 0, in_up, stream);
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

// FSV_WGRAD_PF = 1 | 2: prefetch distance of the 64x64 and 128x32 weight-gradient kernels.  2 is opt-in: 128 / 144 registers
// instead of 80 / 92 (3 instead of 5 - 6 waves per SIMD) for 49.12 / 49.30 vs 49.31 / 49.34 ms per step in-box (round 3, pass r3w)
// - inside the noise of the pair.
static inline int fsv_wgrad_pf() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("FSV_WGRAD_PF"); v = e ? atoi(e) : 1; if (v != 2) v = 1; }
  return v;
}

int fsv_conv_wgrad(const float* in, const float* dout, float* dwt,
                   int N, int H, int W, int Cin, int OH, int OW, int Cout,
                   int ntaps, const int* ty, const int* tx, int sy, int sx,
                   int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                   int force_tile, int in_up, hipStream_t stream) {
  if (!in || !dout || !dwt || ntaps < 1 || ntaps > 16) return FSV_ERR_BAD_ARG;
  // in_up: `in` at H / 2 x W / 2 behind a folded nearest x2 up-sampling (ConvP::up): the float4 MFMA kernels only
  if (in_up && ((H & 1) || (W & 1) || (Cin % 4 != 0) || (Cout & 3) != 0 || Cout <= 4 || per_sample || FSV_BK / OW + 1 > OH))
    return FSV_ERR_UNSUPPORTED;
  for (int t = 0; t < ntaps; ++t)
    if (ty[t] < -8 || ty[t] > 7 || tx[t] < -8 || tx[t] > 7) return FSV_ERR_UNSUPPORTED;
  if ((long long)N * H * W * Cin * 4 > FSV_BUF_MAX_BYTES || (long long)N * OH * OW * Cout * 4 > FSV_BUF_MAX_BYTES) return FSV_ERR_UNSUPPORTED;      // 32-bit byte offsets
  WgradP p;
  p.in = in; p.dout = dout; p.dwt = dwt;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout;
  p.K = ntaps * Cin; p.ldw = ldw; p.sy = sy; p.sx = sx; p.ntaps = ntaps;
  fsv_pack_taps(ty, tx, ntaps, p.taps_lo, p.taps_hi, 8);
  p.w_bstride = w_bstride; p.per_sample = per_sample ? 1 : 0;
  p.Mz = per_sample ? OH * OW : N * OH * OW;
  p.pchunks = fsv_cdiv(p.Mz, FSV_BK);
  p.up = in_up ? 1 : 0;
  const int nsamp = per_sample ? N : 1;
  if (Cout <= 4 && vec4_ok(Cin) && !per_sample && force_tile == 0 && force_split <= 0 && ntaps * (Cin >> 2) <= 256 &&
      fsv_conv_thin(p.Mz, p.K)) {
    // thin-output layers: vector-ALU reduction (fsv_conv_thin_wgrad_kernel), ~512 workgroups, atomics into the zeroed matrix
    p.nsplit = 1;
    if (!prezeroed) (void)hipMemsetAsync(dwt, 0, (size_t)Kpad * ldw * sizeof(float), stream);
    int chunk = fsv_cdiv(p.Mz, 512);
    if (chunk < 64) chunk = 64;
    const dim3 g(fsv_cdiv(p.Mz, chunk));
    switch (Cout) {
      case 1: FSV_LAUNCH((fsv_conv_thin_wgrad_kernel<1>), g, dim3(256), stream, p, chunk); break;
      case 2: FSV_LAUNCH((fsv_conv_thin_wgrad_kernel<2>), g, dim3(256), stream, p, chunk); break;
      case 3: FSV_LAUNCH((fsv_conv_thin_wgrad_kernel<3>), g, dim3(256), stream, p, chunk); break;
      default: FSV_LAUNCH((fsv_conv_thin_wgrad_kernel<4>), g, dim3(256), stream, p, chunk); break;
    }
    return fsv_check_launch();
  }
  int bn = (Cout <= 32) ? 32 : (Cout <= 64 ? 64 : 128);
  // rows of the weight-gradient tile = taps * Cin; the 1x1 SPADE / embedding layers have only 32 or 64 of them and
  // would waste 3/4 or 1/2 of a 128-row tile's MFMA work
  int bmk = 128;
  if (vec4_ok(Cin) && bn >= 64) bmk = (p.K <= 32) ? 32 : (p.K <= 64 ? 64 : 128);
  // force_tile (A/B runs, tools/wgrad_ab.py): 1 = 64x64, 2 = 128x64, 3 = 64x128 (rows x columns), vec4 layers only
  if (force_tile == 1 && vec4_ok(Cin) && Cout > 32) { bmk = 64; bn = 64; }
  else if (force_tile == 2 && vec4_ok(Cin) && Cout > 32) { bmk = 128; bn = 64; }
  else if (force_tile == 3 && vec4_ok(Cin) && Cout > 64) { bmk = 64; bn = 128; }
  else if (force_tile == 4 && vec4_ok(Cin) && Cout > 64) { bmk = 128; bn = 128; }
  long long target = fsv_tune(2);
  static int wplan_v = -1;
  if (wplan_v < 0) { const char* e = getenv("FSV_WGRAD_PLAN"); wplan_v = e ? atoi(e) : 1; }
  if (force_tile == 0 && wplan_v == 1 && vec4_ok(Cin) && Cout >= 64 && p.K > 64) {
    // in-box A/B on the step's layer shapes (tools/wgrad_ab.py, profiles/r02_wgrad_ab.jsonl): the 64x64 tile with ~2048
    // workgroups is best or within 2 % of the best on every shape (83 ... 111 TFLOP/s); 128-row tiles lose 10 ... 25 %
    bmk = 64; bn = 64; target = 2048;
  }
  long long blocks = (long long)fsv_cdiv(p.K, bmk) * fsv_cdiv(Cout, bn) * nsamp;
  int nsplit = 1;
  if (force_split > 0) nsplit = force_split;
  else if (fsv_deterministic()) nsplit = 1;
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
  // the V4 kernel advances (n, oy, ox) by 32 pixels with one conditional subtract per level: needs 32 / OW + 1 <= OH
  const bool vec4 = (Cin % 4 == 0) && (FSV_BK / OW + 1 <= OH) && ((Cout & 3) == 0 || bn == 32);
  dim3 g(fsv_cdiv(p.K, bmk), fsv_cdiv(Cout, bn), nsamp * nsplit);
  if (vec4) {
    if (bn == 128 && bmk == 32) FSV_LAUNCH((fsv_conv_wgrad_kernel<32, 128, 1, 4, true>), g, block, stream, p);
    else if (bn == 128 && bmk == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 128, 2, 2, true>), g, block, stream, p);
    else if (bn == 64 && bmk == 32) FSV_LAUNCH((fsv_conv_wgrad_kernel<32, 64, 1, 2, true>), g, dim3(128), stream, p);
    else if (bn == 64 && bmk == 64 && fsv_wgrad_pf() == 2) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 64, 2, 2, true, 2>), g, block, stream, p);
    else if (bn == 64 && bmk == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<64, 64, 2, 2, true>), g, block, stream, p);
    else if (bn == 128) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 128, 2, 2, true>), g, block, stream, p);
    else if (bn == 64) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 64, 2, 2, true>), g, block, stream, p);
    else if ((Cout & 3) == 0 && fsv_wgrad_pf() == 2) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 32, 4, 1, true, 2>), g, block, stream, p);
    else if ((Cout & 3) == 0) FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 32, 4, 1, true>), g, block, stream, p);
    else FSV_LAUNCH((fsv_conv_wgrad_kernel<128, 32, 4, 1, false>), g, block, stream, p);
  } else {
    if (bn == 128) FSV_LAUNCH((fsv_conv_wgrad_v1_kernel<128, 128, 2, 2>), g, block, stream, p);
    else if (bn == 64) FSV_LAUNCH((fsv_conv_wgrad_v1_kernel<128, 64, 2, 2>), g, block, stream, p);
    else FSV_LAUNCH((fsv_conv_wgrad_v1_kernel<128, 32, 4, 1>), g, block, stream, p);
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


// ---- grouped launches (include/fsv2v.h: fsv_conv_gather_group / fsv_conv_wgrad_group) ------------------------------------
static inline double fsv_group_cost(const ConvP* ps, const int* nsamp, int n, int tile) {
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return 1e30;
  const bool w8 = (tile == 0 || tile == 1 || tile == 9);
  const double cyc = (double)bm * bn / 4.0;
  const double ovh = 5000.0 + bm * bn / 8.0;
  double load = 0.0, crit = 0.0;
  for (int i = 0; i < n; ++i) {
    const double wgs = (double)fsv_cdiv(ps[i].Mz, bm) * fsv_cdiv(ps[i].Cout, bn) * nsamp[i];
    const double one = ps[i].nchunks * cyc;
    load += wgs * (one / 0.95 + 0.45 * ovh);
    const double lone = one / (w8 ? 0.91 : 0.78) + ovh;
    if (lone > crit) crit = lone;
  }
  load /= 256.0;
  return (load > crit ? load : crit) / 1.95e9 + 4e-6;
}

static inline void fsv_group_order(const long long* weight, int n, int* order) {
  // longest problems first: the dispatcher hands workgroups out in grid order, so the long tiles start first and the short
  // ones fill the gaps (LPT)
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i) {
    const int v = order[i];
    int j = i - 1;
    while (j >= 0 && weight[order[j]] < weight[v]) { order[j + 1] = order[j]; --j; }
    order[j + 1] = v;
  }
}

#define FSV_GROUP_LIMIT 64

int fsv_conv_gather_group(const fsv_conv_desc* d, int n, int force_tile, hipStream_t stream) {
  if (!d || n < 1 || n > FSV_GROUP_LIMIT) return FSV_ERR_BAD_ARG;
  ConvP ps[FSV_GROUP_LIMIT];
  int nsamp[FSV_GROUP_LIMIT];
  long long weight[FSV_GROUP_LIMIT];
  bool vec4 = true;
  int max_cout = 0;
  for (int i = 0; i < n; ++i) {
    const fsv_conv_desc& q = d[i];
    if (!q.in || !q.wt || !q.out || q.ntaps < 1 || q.ntaps > 16 || q.N < 1 || q.Cin < 1 || q.Cout < 1) return FSV_ERR_BAD_ARG;
    for (int t = 0; t < q.ntaps; ++t)
      if (q.ty[t] < -8 || q.ty[t] > 7 || q.tx[t] < -8 || q.tx[t] > 7) return FSV_ERR_UNSUPPORTED;
    if ((q.ldw & 3) != 0 || q.ldw < q.Cout) return FSV_ERR_BAD_ARG;
    if ((long long)q.N * q.H * q.W * q.Cin * 4 > FSV_BUF_MAX_BYTES || (long long)(q.ntaps * q.Cin + 32) * q.ldw * 4 > FSV_BUF_MAX_BYTES)
      return FSV_ERR_UNSUPPORTED;
    if (q.accumulate && (q.bias || q.res || q.act != FSV_ACT_NONE || q.scale != 1.f)) return FSV_ERR_BAD_ARG;
    if (q.act == FSV_ACT_DLRELU && !q.res) return FSV_ERR_BAD_ARG;
    fsv_fill_convp(ps[i], q.in, q.wt, q.bias, q.res, q.out, q.wscale, q.N, q.H, q.W, q.Cin, q.OH, q.OW, q.Cout, q.ntaps, q.ty,
                   q.tx, q.sy, q.sx, q.outH, q.outW, q.osy, q.osx, q.ooy, q.oox, q.ldw, q.w_bstride, q.b_bstride, q.per_sample,
                   q.act, q.scale);
    nsamp[i] = q.per_sample ? q.N : 1;
    vec4 = vec4 && vec4_ok(q.Cin);
    if (q.Cout > max_cout) max_cout = q.Cout;
  }
  // one tile shape for the whole group
  int tile = force_tile;
  if (tile < 0) {
    static const int tiles[5] = {0, 9, 1, 4, 2};
    double best = 1e30;
    for (int ti = 0; ti < 5; ++ti) {
      const int t = tiles[ti];
      if ((t == 0 || t == 9) && max_cout <= 64) continue;
      if (t == 1 && max_cout <= 32) continue;
      if (t == 2 && max_cout > 32) continue;
      const double c = fsv_group_cost(ps, nsamp, n, t) * (1.0 + 0.01 * ti);
      if (c < best) { best = c; tile = t; }
    }
    if (tile < 0) tile = 4;
  }
  if (tile >= 10) tile = (tile == 10 || tile == 13 || tile == 21) ? 9 : (tile == 11 || tile == 14 || tile == 16) ? 0 : (tile == 17 || tile == 20 || tile == 27) ? 4 : (tile == 18) ? 2 : 1;
  int bm, bn;
  if (fsv_tile_dims(tile, bm, bn)) return FSV_ERR_BAD_ARG;
  // K splits: only problems that accumulate into a zeroed output, and only when the whole group would leave CUs idle
  long long wgs1 = 0;
  for (int i = 0; i < n; ++i) wgs1 += (long long)fsv_cdiv(ps[i].Mz, bm) * fsv_cdiv(ps[i].Cout, bn) * nsamp[i];
  // (in-box, round 3: the parity classes of a 16x16 / 32x32 stride-2 data gradient as 256 unsplit tiles of up to 128 chunks
  // took 188 us - as long as the four split launches they replaced)
  int want = 1;
  if (wgs1 <= 384 && !fsv_deterministic()) { want = (int)((768 + wgs1 - 1) / (wgs1 > 0 ? wgs1 : 1)); if (want > 8) want = 8; if (want < 1) want = 1; }
  for (int i = 0; i < n; ++i) {
    int sp = 1;
    if (d[i].accumulate && want > 1) {
      sp = want;
      if (sp > ps[i].nchunks / 8) sp = ps[i].nchunks / 8;
      if (sp < 1) sp = 1;
    }
    ps[i].nsplit = sp;
    weight[i] = (long long)fsv_cdiv(ps[i].nchunks, sp);
  }
  int order[FSV_GROUP_LIMIT];
  fsv_group_order(weight, n, order);
  const int variant = vec4 ? fsv_conv_variant(tile) : tile;
  for (int b0 = 0; b0 < n; b0 += FSV_GROUP_MAX) {
    ConvGroup g;
    g.nprob = (n - b0 < FSV_GROUP_MAX) ? (n - b0) : FSV_GROUP_MAX;
    int tiles = 0;
    for (int j = 0; j < FSV_GROUP_MAX; ++j) {
      if (j < g.nprob) {
        const int i = order[b0 + j];
        g.p[j] = ps[i];
        tiles += fsv_cdiv(ps[i].Mz, bm) * fsv_cdiv(ps[i].Cout, bn) * nsamp[i] * ps[i].nsplit;
      } else {
        g.p[j] = ps[order[b0]];
      }
      g.tile_end[j] = tiles;
    }
    const dim3 grid(tiles);
    if (vec4) {
      switch (variant) {
        case 0: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 128, 2, 4>), grid, dim3(512), stream, g); break;
        case 1: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 64, 4, 2>), grid, dim3(512), stream, g); break;
        case 2: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 32, 4, 1>), grid, dim3(256), stream, g); break;
        case 4: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 64, 2, 2>), grid, dim3(256), stream, g); break;
        case 9: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 128, 2, 4>), grid, dim3(512), stream, g); break;
        case 10: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 128, 2, 4, 2>), grid, dim3(512), stream, g); break;
        case 11: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 128, 2, 4, 2>), grid, dim3(512), stream, g); break;
        case 12: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 64, 4, 2, 2>), grid, dim3(512), stream, g); break;
        case 13: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 128, 2, 4, 2, true>), grid, dim3(512), stream, g); break;
        case 14: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 128, 2, 4, 2, true>), grid, dim3(512), stream, g); break;
        case 15: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 64, 4, 2, 2, true>), grid, dim3(512), stream, g); break;
        case 16: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 128, 2, 4, 1, true>), grid, dim3(512), stream, g); break;
        case 17: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 64, 2, 2, 1, true>), grid, dim3(256), stream, g); break;
        case 18: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 32, 4, 1, 1, true>), grid, dim3(256), stream, g); break;
        case 20: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 64, 2, 2, 2, true>), grid, dim3(256), stream, g); break;
        case 21: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 128, 2, 4, 2, true, 2>), grid, dim3(512), stream, g); break;
        case 22: FSV_LAUNCH((fsv_conv_igemm_group_kernel<128, 64, 4, 2, 2, true, 2>), grid, dim3(512), stream, g); break;
        case 27: FSV_LAUNCH((fsv_conv_igemm_group_kernel<64, 64, 2, 2, 2, true, 2>), grid, dim3(256), stream, g); break;
        default: return FSV_ERR_BAD_ARG;
      }
    } else {
      switch (tile) {
        case 0: FSV_LAUNCH((fsv_conv_igemm_v1_group_kernel<128, 128, 2, 2>), grid, dim3(256), stream, g); break;
        case 1: FSV_LAUNCH((fsv_conv_igemm_v1_group_kernel<128, 64, 2, 2>), grid, dim3(256), stream, g); break;
        case 2: FSV_LAUNCH((fsv_conv_igemm_v1_group_kernel<128, 32, 4, 1>), grid, dim3(256), stream, g); break;
        case 4: FSV_LAUNCH((fsv_conv_igemm_v1_group_kernel<64, 64, 2, 2>), grid, dim3(256), stream, g); break;
        default: FSV_LAUNCH((fsv_conv_igemm_v1_group_kernel<64, 128, 2, 2>), grid, dim3(256), stream, g); break;
      }
    }
  }
  return fsv_check_launch();
}

// Tile of the grouped launch the planner would pick (tests / profiler labels)
int fsv_conv_group_plan(const int* Mz, const int* Cout, const int* nchunks, const int* nsamp, int n, int* tile_out) {
  if (!Mz || !Cout || !nchunks || !nsamp || !tile_out || n < 1 || n > FSV_GROUP_LIMIT) return FSV_ERR_BAD_ARG;
  ConvP ps[FSV_GROUP_LIMIT];
  int max_cout = 0;
  for (int i = 0; i < n; ++i) { ps[i].Mz = Mz[i]; ps[i].Cout = Cout[i]; ps[i].nchunks = nchunks[i]; if (Cout[i] > max_cout) max_cout = Cout[i]; }
  static const int tiles[5] = {0, 9, 1, 4, 2};
  double best = 1e30;
  int tile = 4;
  for (int ti = 0; ti < 5; ++ti) {
    const int t = tiles[ti];
    if ((t == 0 || t == 9) && max_cout <= 64) continue;
    if (t == 1 && max_cout <= 32) continue;
    if (t == 2 && max_cout > 32) continue;
    const double c = fsv_group_cost(ps, nsamp, n, t) * (1.0 + 0.01 * ti);
    if (c < best) { best = c; tile = t; }
  }
  *tile_out = tile;
  return FSV_OK;
}

// Grouped weight gradients: every dwt is a ZEROED [Kpad][ldw] matrix (slices of the optimiser's per-pass arena); 64x64 tiles
// (the tile the single launches use wherever it applies, profiles/r02_wgrad_ab.jsonl), pixel ranges split so that the group
// as a whole fills the chip.  FSV_ERR_UNSUPPORTED (nothing launched) when a problem needs the scalar gather: callers then
// issue the problems one by one.
int fsv_conv_wgrad_group(const fsv_wgrad_desc* d, int n, hipStream_t stream) {
  if (!d || n < 1 || n > FSV_GROUP_LIMIT) return FSV_ERR_BAD_ARG;
  WgradP ps[FSV_GROUP_LIMIT];
  int nsamp[FSV_GROUP_LIMIT];
  long long weight[FSV_GROUP_LIMIT];
  bool cout4 = true;
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    const fsv_wgrad_desc& q = d[i];
    if (!q.in || !q.dout || !q.dwt || q.ntaps < 1 || q.ntaps > 16) return FSV_ERR_BAD_ARG;
    for (int t = 0; t < q.ntaps; ++t)
      if (q.ty[t] < -8 || q.ty[t] > 7 || q.tx[t] < -8 || q.tx[t] > 7) return FSV_ERR_UNSUPPORTED;
    if ((long long)q.N * q.H * q.W * q.Cin * 4 > FSV_BUF_MAX_BYTES || (long long)q.N * q.OH * q.OW * q.Cout * 4 > FSV_BUF_MAX_BYTES)
      return FSV_ERR_UNSUPPORTED;
    if (!vec4_ok(q.Cin) || !(FSV_BK / q.OW + 1 <= q.OH)) return FSV_ERR_UNSUPPORTED;
    WgradP& p = ps[i];
    p.in = q.in; p.dout = q.dout; p.dwt = q.dwt;
    p.N = q.N; p.H = q.H; p.W = q.W; p.Cin = q.Cin; p.OH = q.OH; p.OW = q.OW; p.Cout = q.Cout;
    p.K = q.ntaps * q.Cin; p.ldw = q.ldw; p.sy = q.sy; p.sx = q.sx; p.ntaps = q.ntaps;
    fsv_pack_taps(q.ty, q.tx, q.ntaps, p.taps_lo, p.taps_hi, 8);
    p.w_bstride = q.w_bstride; p.per_sample = q.per_sample ? 1 : 0;
    p.Mz = q.per_sample ? q.OH * q.OW : q.N * q.OH * q.OW;
    p.pchunks = fsv_cdiv(p.Mz, FSV_BK); p.up = 0;
    nsamp[i] = q.per_sample ? q.N : 1;
    cout4 = cout4 && (q.Cout & 3) == 0;
    blocks += (long long)fsv_cdiv(p.K, 64) * fsv_cdiv(q.Cout, 64) * nsamp[i];
  }
  // the same ~2048-workgroup target as the single launches, for the group as a whole; at least 8 pixel chunks per split
  int want = (int)((2048 + blocks - 1) / (blocks > 0 ? blocks : 1));
  if (want < 1 || fsv_deterministic()) want = 1;
  for (int i = 0; i < n; ++i) {
    int sp = want;
    const int maxs = ps[i].pchunks / 8;
    if (sp > maxs) sp = maxs;
    if (sp < 1) sp = 1;
    ps[i].nsplit = sp;
    weight[i] = (long long)fsv_cdiv(ps[i].pchunks, sp);
  }
  int order[FSV_GROUP_LIMIT];
  fsv_group_order(weight, n, order);
  for (int b0 = 0; b0 < n; b0 += FSV_GROUP_MAX) {
    WgradGroup g;
    g.nprob = (n - b0 < FSV_GROUP_MAX) ? (n - b0) : FSV_GROUP_MAX;
    int tiles = 0;
    for (int j = 0; j < FSV_GROUP_MAX; ++j) {
      if (j < g.nprob) {
        const int i = order[b0 + j];
        g.p[j] = ps[i];
        tiles += fsv_cdiv(ps[i].K, 64) * fsv_cdiv(ps[i].Cout, 64) * nsamp[i] * ps[i].nsplit;
      } else {
        g.p[j] = ps[order[b0]];
      }
      g.tile_end[j] = tiles;
    }
    if (cout4) FSV_LAUNCH((fsv_conv_wgrad_group_kernel<64, 64, 2, 2, true>), dim3(tiles), dim3(256), stream, g);
    else FSV_LAUNCH((fsv_conv_wgrad_group_kernel<64, 64, 2, 2, false>), dim3(tiles), dim3(256), stream, g);
  }
  return fsv_check_launch();
}

}  // extern "C"
