// Deferred weight-gradient finalisation: every convolution / linear weight gradient of one backward pass in two launches.
//
// The weight-gradient GEMM (conv_igemm.hip: fsv_conv_wgrad) leaves dW in the K-major operand layout
// dwt[(tap, ci)][co].  Turning that into the parameter's OIHW gradient - plus, for spectral-normalised layers, the
// correction of torch.nn.utils.spectral_norm's backward,
//     dW = (dW_sn - <dW_sn, W> / sigma * u v^T) / sigma                       (models/networks/base_network.py:73-76
//                                                                              wraps every conv in spectral_norm)
// - used to cost three or four small launches per layer (~440 per training step).  The autograd functions now only
// queue a job (pointers + geometry); before the gradient exchange / Adam step the optimiser hands the whole queue to
// fsv_wgrad_finalize, which runs
//   1. fsv_wfin_dot_kernel    <dW_sn, W> of every spectral-normalised job (fp64 block sums, one fp64 atomic per block;
//                             W is read in the same K-major layout from the optimiser's layout cache: coalesced),
//   2. fsv_wfin_apply_kernel  an LDS-tiled transpose [(tap, ci)][co] -> [co][ci][kh][kw] that adds the finished gradient
//                             into the flat gradient buffer (both sides coalesced).
// HBM-bound: reads 4 B (+4 B of W for the dot) and read-modify-writes 8 B per weight element.
#include "fsv_common.h"

// job descriptors (device arrays, built by the host once per backward pass):
//   ptrs[job][6] = { dwt, wlay (K-major W, spectral jobs only), sink (OIHW gradient), u, v, sig } as 64-bit integers
//   dims[job][8] = { Cout, CinP (channels in dwt), CinR (channels of the parameter), KH, KW, ntaps, ldw, flags }
//                  flags bit 0: another job of this launch adds into the same sink -> atomic adds
//   taps[job][2] = packed (kh | kw << 4) codes of dwt's tap order
struct FinP {
  const long long* ptrs;
  const int* dims;
  const unsigned long long* taps;
  double* dots;
};

#define FSV_FIN_DOT_CHUNK 4096

__global__ __launch_bounds__(256) void fsv_wfin_dot_kernel(FinP f, const int* tmap) {
  __shared__ double red[256];
  const int job = tmap[blockIdx.x * 2], chunk = tmap[blockIdx.x * 2 + 1];
  const int* d = f.dims + job * 8;
  const int Cout = d[0], CinP = d[1], ntaps = d[5], ldw = d[6];
  const float* dwt = reinterpret_cast<const float*>(f.ptrs[job * 6]);
  const float* wl = reinterpret_cast<const float*>(f.ptrs[job * 6 + 1]);
  const long long total = (long long)ntaps * CinP * ldw;
  float acc = 0.f;
#pragma unroll 4
  for (int s = 0; s < FSV_FIN_DOT_CHUNK / 256; ++s) {
    const long long i = (long long)chunk * FSV_FIN_DOT_CHUNK + s * 256 + threadIdx.x;
    if (i < total) {
      const int c = (int)(i % ldw);
      if (c < Cout) acc += dwt[i] * wl[i];         // padding columns of dwt are never written: skip, do not multiply
    }
  }
  red[threadIdx.x] = (double)acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&f.dots[job], red[0]);
}

// tile: 32 output channels x CI_T input channels x all taps; CI_T = 32 for <= 8 taps, 16 otherwise (LDS <= 33.8 KB).
// A pure HBM stream (4 B read, 8 B read-modify-written per element): rows of dwt are read as float4 with every load of a
// work-item issued before the first LDS write, and the gradient runs are walked as a flat element sequence, four sink elements
// in flight per work-item; KKT = source tap count as a compile-time constant (0: any, run-time divisions).
template <int KKT>
__device__ __forceinline__ void fsv_wfin_tile(const FinP& f, const int job, const int cot, const int cit, float* t,
                                              const int* inv_tap) {
  const int* d = f.dims + job * 8;
  const int Cout = d[0], CinP = d[1], CinR = d[2], KW = d[4], ntaps = d[5], ldw = d[6], flags = d[7];
  const int KK = KKT ? KKT : d[3] * KW;
  const int CI_T = ntaps <= 8 ? 32 : 16;
  const float* dwt = reinterpret_cast<const float*>(f.ptrs[job * 6]);
  float* sink = reinterpret_cast<float*>(f.ptrs[job * 6 + 2]);
  const float* u = reinterpret_cast<const float*>(f.ptrs[job * 6 + 3]);
  const float* v = reinterpret_cast<const float*>(f.ptrs[job * 6 + 4]);
  const float* sig = reinterpret_cast<const float*>(f.ptrs[job * 6 + 5]);
  const int co0 = cot * 32, ci0 = cit * CI_T;
  const int tid = threadIdx.x;
  // ---- read: rows (tap, ci) of dwt, 32 consecutive output channels each: 8 work-items x float4 per row, 32 rows per pass
  {
    const int q = tid & 7, r0 = tid >> 3;
    const int nrows = ntaps * CI_T;
    for (int rb = r0; rb < nrows; rb += 128) {
      float4 val[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rb + 32 * k;
        const int j = r / CI_T, cil = r - j * CI_T;
        const int ci = ci0 + cil, co = co0 + 4 * q;
        // columns in [Cout, ldw) were never written by the weight-gradient kernel: they are masked per element below
        val[k] = (r < nrows && ci < CinP && co < ldw) ? *reinterpret_cast<const float4*>(&dwt[((long long)j * CinP + ci) * ldw + co])
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = rb + 32 * k;
        if (r < nrows) {
          float* dst = t + r * 33 + 4 * q;
          const int co = co0 + 4 * q;
          dst[0] = co < Cout ? val[k].x : 0.f; dst[1] = co + 1 < Cout ? val[k].y : 0.f;
          dst[2] = co + 2 < Cout ? val[k].z : 0.f; dst[3] = co + 3 < Cout ? val[k].w : 0.f;
        }
      }
    }
  }
  __syncthreads();
  // ---- write: per output channel a contiguous run of CI_T * KK gradient elements
  float inv = 1.f, coef = 0.f;
  if (sig) { inv = sig[1]; coef = inv * (float)f.dots[job]; }
  const int run = CI_T * KK;
  const int total = 32 * run;
  for (int e0 = tid; e0 < total; e0 += 1024) {
    float g[4], old[4];
    float* dst[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + 256 * k;
      const int col = e / run, idx = e - col * run;
      const int cil = idx / KK, tk = idx - cil * KK;
      const int co = co0 + col, ci = ci0 + cil;
      const int j = (e < total) ? inv_tap[tk] : -1;
      const bool ok = (e < total) & (co < Cout) & (ci < CinR) & (j >= 0);
      dst[k] = nullptr; g[k] = 0.f; old[k] = 0.f;
      if (ok) {
        float gv = t[(j * CI_T + cil) * 33 + col];
        if (sig) gv = inv * (gv - coef * u[co] * v[ci * KK + tk]);
        g[k] = gv;
        dst[k] = sink + ((long long)co * CinR + ci) * KK + tk;
        if (!(flags & 1)) old[k] = *dst[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (dst[k]) {
        if (flags & 1) atomicAdd(dst[k], g[k]); else *dst[k] = old[k] + g[k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void fsv_wfin_apply_kernel(FinP f, const int* tmap) {
  __shared__ float t[8 * 32 * 33];
  __shared__ int inv_tap[16];
  const int job = tmap[blockIdx.x * 3], cot = tmap[blockIdx.x * 3 + 1], cit = tmap[blockIdx.x * 3 + 2];
  const int* d = f.dims + job * 8;
  const int KW = d[4], ntaps = d[5];
  const int KK = d[3] * KW;
  const unsigned long long lo = f.taps[job * 2], hi = f.taps[job * 2 + 1];
  if (threadIdx.x < 16) inv_tap[threadIdx.x] = -1;
  __syncthreads();
  if ((int)threadIdx.x < ntaps) {
    const int j = threadIdx.x;
    const unsigned long long code = (j < 8) ? lo : hi;
    const int sh = (j & 7) * 8;
    const int kh = (int)((code >> sh) & 15ull), kw = (int)((code >> (sh + 4)) & 15ull);
    inv_tap[kh * KW + kw] = j;
  }
  __syncthreads();
  if (KK == 9) fsv_wfin_tile<9>(f, job, cot, cit, t, inv_tap);
  else if (KK == 1) fsv_wfin_tile<1>(f, job, cot, cit, t, inv_tap);
  else if (KK == 16) fsv_wfin_tile<16>(f, job, cot, cit, t, inv_tap);
  else fsv_wfin_tile<0>(f, job, cot, cit, t, inv_tap);
}

// ---- pointer-table upload through kernel arguments ----------------------------------------------------------------------
// The job table holds addresses of this pass's temporaries, so it changes every eager step.  A host-to-device copy would
// need pinned staging memory (whose allocation is illegal inside a hipGraph capture) and would make a captured graph depend
// on a host buffer; kernel arguments are copied at launch / baked into the graph node, so the words travel in them.
#define FSV_UPLOAD_WORDS 448
struct UploadWords { long long w[FSV_UPLOAD_WORDS]; };
__global__ __launch_bounds__(256) void fsv_upload_kernel(long long* dst, int n, UploadWords s) {
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = s.w[i];
}

extern "C" int fsv_upload_i64(long long* dst, const long long* host_src, int n, hipStream_t stream) {
  if (!dst || !host_src || n < 1) return FSV_ERR_BAD_ARG;
  for (int off = 0; off < n; off += FSV_UPLOAD_WORDS) {
    UploadWords s;
    const int m = (n - off < FSV_UPLOAD_WORDS) ? n - off : FSV_UPLOAD_WORDS;
    for (int i = 0; i < m; ++i) s.w[i] = host_src[off + i];
    for (int i = m; i < FSV_UPLOAD_WORDS; ++i) s.w[i] = 0;
    FSV_LAUNCH(fsv_upload_kernel, dim3(1), dim3(256), stream, dst + off, m, s);
  }
  return fsv_check_launch();
}

// ---- grouped dst += src over many small tensors (norm weights / biases, fixed SPADE weights) ------------------------------------
// table[job][3] = { src, dst, n }; tmap (job, 4096-element chunk).  Replaces one autograd AccumulateGrad add per parameter.
__global__ __launch_bounds__(256) void fsv_gather_add_kernel(const long long* table, const int* tmap) {
  const int job = tmap[blockIdx.x * 2], chunk = tmap[blockIdx.x * 2 + 1];
  const float* src = reinterpret_cast<const float*>(table[job * 3]);
  float* dst = reinterpret_cast<float*>(table[job * 3 + 1]);
  const long long n = table[job * 3 + 2];
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const long long i = (long long)chunk * 4096 + s * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
  }
}

extern "C" int fsv_gather_add(const long long* table, int njobs, const int* tmap, int nblk, hipStream_t stream) {
  if (!table || !tmap || njobs < 1 || nblk < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_gather_add_kernel, dim3(nblk), dim3(256), stream, table, tmap);
  return fsv_check_launch();
}

// ---- grouped fixed-order sums: dst_j = src_j[0] + src_j[1] + ... (left to right) for up to 8 jobs of up to 4 terms ------------
// (the data gradients of the weight-generator chains that share their input rows, ops._MlpBankFn: one grouped GEMM launch
// writes every chain's term, this launch adds them in a fixed order - no atomics, no chain of dependent GEMM launches)
struct SumJobs {
  int njobs;
  int blk_end[8];            // cumulative 1024-element blocks
  int nsrc[8];
  long long count[8];
  float* dst[8];
  const float* src[8][4];
};

__global__ __launch_bounds__(256) void fsv_sum_terms_kernel(SumJobs g) {
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < g.njobs && b >= g.blk_end[j]) ++j;
  const long long i0 = (long long)(b - (j ? g.blk_end[j - 1] : 0)) * 1024 + threadIdx.x * 4;
  const long long n = g.count[j];
  if (i0 >= n) return;
  const int ns = g.nsrc[j];
  if (i0 + 4 <= n) {
    float4 a = *reinterpret_cast<const float4*>(g.src[j][0] + i0);
    for (int t = 1; t < ns; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(g.src[j][t] + i0);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(g.dst[j] + i0) = a;
  } else {
    for (long long i = i0; i < n; ++i) {
      float a = g.src[j][0][i];
      for (int t = 1; t < ns; ++t) a += g.src[j][t][i];
      g.dst[j][i] = a;
    }
  }
}

// dst[j] / src[j * 4 + t]: device pointers (16-byte aligned), nsrc[j] in 1..4 terms, count[j] elements; njobs <= 8
extern "C" int fsv_sum_terms(float* const* dst, const float* const* src, const int* nsrc, const long long* count, int njobs,
                             hipStream_t stream) {
  if (!dst || !src || !nsrc || !count || njobs < 1 || njobs > 8) return FSV_ERR_BAD_ARG;
  SumJobs g;
  g.njobs = njobs;
  int blocks = 0;
  for (int j = 0; j < 8; ++j) {
    const int k = j < njobs ? j : 0;
    if (nsrc[k] < 1 || nsrc[k] > 4 || count[k] < 1 || !dst[k]) return FSV_ERR_BAD_ARG;
    g.nsrc[j] = nsrc[k]; g.count[j] = count[k]; g.dst[j] = dst[k];
    for (int t = 0; t < 4; ++t) {
      g.src[j][t] = src[k * 4 + (t < nsrc[k] ? t : 0)];
      if (!g.src[j][t] || ((uintptr_t)g.src[j][t] & 15)) return FSV_ERR_BAD_ARG;
    }
    if ((uintptr_t)g.dst[j] & 15) return FSV_ERR_BAD_ARG;
    if (j < njobs) blocks += (int)((count[k] + 1023) / 1024);
    g.blk_end[j] = blocks;
  }
  FSV_LAUNCH(fsv_sum_terms_kernel, dim3(blocks), dim3(256), stream, g);
  return fsv_check_launch();
}

extern "C" int fsv_wgrad_finalize(const long long* ptrs, const int* dims, const unsigned long long* taps, double* dots,
                                  int njobs, const int* tmap_dot, int nblk_dot, const int* tmap_apply, int nblk_apply,
                                  hipStream_t stream) {
  if (!ptrs || !dims || !taps || !dots || njobs < 1 || !tmap_apply || nblk_apply < 1 || (nblk_dot > 0 && !tmap_dot))
    return FSV_ERR_BAD_ARG;
  FinP f; f.ptrs = ptrs; f.dims = dims; f.taps = taps; f.dots = dots;
  (void)hipMemsetAsync(dots, 0, sizeof(double) * (size_t)njobs, stream);
  if (nblk_dot > 0) FSV_LAUNCH(fsv_wfin_dot_kernel, dim3(nblk_dot), dim3(256), stream, f, tmap_dot);
  FSV_LAUNCH(fsv_wfin_apply_kernel, dim3(nblk_apply), dim3(256), stream, f, tmap_apply);
  return fsv_check_launch();
}
