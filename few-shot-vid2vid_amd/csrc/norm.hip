// Channel statistics, normalisation and their gradients on NHWC activations (HBM-bound kernels).
//
// Covers the normalisation layers on the hot path:
//   * train-mode BatchNorm (apex SyncBatchNorm in one process == nn.BatchNorm2d): models/networks/normalization.py:33,80
//     - param-free inside SPADE, affine inside SPADEConv2d / FlowGenerator
//   * InstanceNorm2d(affine=True, eps=0.1) of the discriminator: normalization.py:82
// A tensor is viewed as [G groups][P pixels][C channels]: BatchNorm has G = 1, P = N*H*W; InstanceNorm has
// G = N, P = H*W.  Reductions are two-stage and deterministic: per-block partial sums (fp32 within a thread's
// short run, fp64 across threads and blocks) followed by a one-block-per-channel-slab finalize.
#include "fsv_common.h"

// ---- two-value column reductions over [G][P][C] -------------------------------------------------------------
// One template serves the three reductions of this file:
//   STATS  : (sum x, sum x^2)                           forward statistics
//   BWD    : (sum d, sum d * xhat), d = dy * act'(y)     normalisation backward
//   COLSUM : (sum x, -)                                  bias gradients
// Grid = (pixel chunks, channel slabs, groups).  A block covers TX column units (float4 = 4 channels when
// C % 4 == 0) x TY = 256/TX rows in flight; every thread keeps 4 independent row streams so that >= 4 vector loads
// per input are outstanding (these kernels are pure HBM streams).  Per-thread fp32 partials over a short run are
// combined in fp64 across the block and written as part[g][chunk][c][2]; a finalize kernel sums the chunks.
struct RedPlan { int V, CU, TX, TY, nslabs, rows_per_blk, nchunks; };

static inline RedPlan fsv_red_plan(int G, int P, int C, int max_chunks = 0) {
  RedPlan r;
  r.V = (C % 4 == 0) ? 4 : 1;
  r.CU = C / r.V;
  int cap = (r.V == 4) ? 32 : 64;
  r.TX = r.CU < cap ? r.CU : cap;
  r.TY = 256 / r.TX;
  r.nslabs = (r.CU + r.TX - 1) / r.TX;
  long long want = (2048 + (long long)r.nslabs * G - 1) / ((long long)r.nslabs * G);   // chunks for ~2048 blocks
  long long maxc = ((long long)P + r.TY * 4 - 1) / (r.TY * 4);                          // >= 4 rows per thread
  long long chunks = want < maxc ? want : maxc;
  if (max_chunks > 0 && chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  r.rows_per_blk = (int)(((long long)P + chunks - 1) / chunks);
  r.nchunks = (P + r.rows_per_blk - 1) / r.rows_per_blk;
  return r;
}

#define FSV_RED_STATS 0
#define FSV_RED_BWD 1
#define FSV_RED_COLSUM 2

__device__ __forceinline__ float fsv_act_grad(float dy, float y, int act) {
  if (act == FSV_ACT_LRELU) return y > 0.f ? dy : 0.2f * dy;
  if (act == FSV_ACT_TANH) return dy * (1.f - y * y);
  if (act == FSV_ACT_SIGMOID) return dy * y * (1.f - y);
  if (act == FSV_ACT_RELU) return y > 0.f ? dy : 0.f;
  if (act == FSV_ACT_LRELU01) return y > 0.f ? dy : 0.1f * dy;
  return dy;
}

struct RedP {
  const float* a;       // STATS/COLSUM: x;  BWD: dy
  const float* y;       // BWD: activated output (may be null when act == none)
  const float* x;       // BWD: normalisation input
  const float* mean;
  const float* rstd;
  double* part;
  int P, C, CU, TX, TY, rows_per_blk, nchunks, act;
  // fused second stage (counter != null): the workgroup that finishes last on a channel slab sums the partials of that
  // slab and writes the final values, so the reduction is ONE launch.  o0..o3 by mode:
  //   STATS : mean, rstd, run_mean, run_var      BWD : s1, s2, dw, db      COLSUM : out
  int* counter;         // one zeroed int per channel slab; the last workgroup resets it
  float *o0, *o1, *o2, *o3;
  float eps, momentum;
  int rep, accumulate;
};

__device__ __forceinline__ void fsv_sum_chunks(const double* part, int g, int c, int C, int nchunks, double& a, double& b);

template <int MODE, int V>
__global__ __launch_bounds__(256) void fsv_red2_kernel(RedP p) {
  __shared__ float red[256 * 2 * V];
  const int chunk = blockIdx.x, slab = blockIdx.y, g = blockIdx.z;
  const int tx = threadIdx.x % p.TX, ty = threadIdx.x / p.TX;
  const int cu = slab * p.TX + tx;
  const bool active = ty < p.TY && cu < p.CU;
  const int r0 = chunk * p.rows_per_blk;
  const int r1 = (r0 + p.rows_per_blk < p.P) ? r0 + p.rows_per_blk : p.P;
  const long long goff = (long long)g * p.P * p.C;
  float s1[V], s2[V], mu[V], rs[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; rs[j] = 1.f; }
  if (active) {
    const int c0 = cu * V;
    if (MODE == FSV_RED_BWD) {
#pragma unroll
      for (int j = 0; j < V; ++j) { mu[j] = p.mean[g * p.C + c0 + j]; rs[j] = p.rstd[g * p.C + c0 + j]; }
    }
    for (int r = r0 + ty; r < r1; r += p.TY * 4) {
      float va[4][V], vy[4][V], vx[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * p.TY;
        const bool ok = rr < r1;
        const long long off = goff + (long long)rr * p.C + c0;
        if constexpr (V == 4) {
          float4 t = ok ? *reinterpret_cast<const float4*>(p.a + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          va[u][0] = t.x; va[u][1] = t.y; va[u][2] = t.z; va[u][3] = t.w;
          if (MODE == FSV_RED_BWD) {
            float4 tx4 = ok ? *reinterpret_cast<const float4*>(p.x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            vx[u][0] = tx4.x; vx[u][1] = tx4.y; vx[u][2] = tx4.z; vx[u][3] = tx4.w;
            float4 ty4 = (ok && p.y) ? *reinterpret_cast<const float4*>(p.y + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            vy[u][0] = ty4.x; vy[u][1] = ty4.y; vy[u][2] = ty4.z; vy[u][3] = ty4.w;
          }
        } else {
          va[u][0] = ok ? p.a[off] : 0.f;
          if (MODE == FSV_RED_BWD) { vx[u][0] = ok ? p.x[off] : 0.f; vy[u][0] = (ok && p.y) ? p.y[off] : 0.f; }
        }
        if (MODE == FSV_RED_BWD && !ok) {
#pragma unroll
          for (int j = 0; j < V; ++j) vx[u][j] = mu[j];      // xhat = 0 for padding rows
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (MODE == FSV_RED_STATS) { s1[j] += va[u][j]; s2[j] += va[u][j] * va[u][j]; }
          else if (MODE == FSV_RED_COLSUM) { s1[j] += va[u][j]; }
          else {
            float d = fsv_act_grad(va[u][j], vy[u][j], p.act);
            s1[j] += d; s2[j] += d * ((vx[u][j] - mu[j]) * rs[j]);
          }
        }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { red[(threadIdx.x * V + j) * 2] = s1[j]; red[(threadIdx.x * V + j) * 2 + 1] = s2[j]; }
  __syncthreads();
  // thread t < TX*V reduces column unit t / V, component t % V over the TY rows (fp64)
  const int t = threadIdx.x;
  if (t < p.TX * V) {
    const int ux = t / V, j = t % V;
    const int c = (slab * p.TX + ux) * V + j;
    if (slab * p.TX + ux < p.CU) {
      double a = 0.0, b = 0.0;
      for (int yy = 0; yy < p.TY; ++yy) {
        const int src = yy * p.TX + ux;
        a += (double)red[(src * V + j) * 2];
        b += (double)red[(src * V + j) * 2 + 1];
      }
      double* dst = p.part + (((long long)g * p.nchunks + chunk) * p.C + c) * 2;
      dst[0] = a; dst[1] = b;
    }
  }
  if (p.counter == nullptr) return;
  // ---- fused second stage: last workgroup of this slab (over all chunks and groups) ------------------------------------
  __shared__ int is_last;
  __threadfence();                       // this thread's partials are visible device-wide before the ticket is taken
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = (int)(gridDim.x * gridDim.z);
    const int prev = atomicAdd(p.counter + slab, 1);
    is_last = (prev == total - 1) ? 1 : 0;
    if (is_last) p.counter[slab] = 0;    // nobody else touches the ticket any more: leave it ready for the next launch
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();                       // acquire: the other workgroups' partials
  // Column j of the slab's partial rows = (channel cc = j / 2, component j % 2); ncol <= 256 contiguous doubles per (g, chunk).
  // Every thread owns one column and a share of the chunks: independent, coalesced loads (the per-channel wave reduction of
  // the two-launch finalize would serialise up to 32 channels per wave here), then one LDS step pairs the components up.
  __shared__ double colsum[256];
  const int G = (int)gridDim.z;
  const int nch = (p.C - slab * p.TX * V) < p.TX * V ? (p.C - slab * p.TX * V) : p.TX * V;     // channels of this slab
  const int ncol = nch * 2;
  const int groups = 256 / ncol > 0 ? 256 / ncol : 1;
  const int col = (int)threadIdx.x % ncol, grp = (int)threadIdx.x / ncol;
  const bool loader = grp < groups;
  const int cc = (int)threadIdx.x;                      // finalising thread cc < nch owns channel c
  const int c = slab * p.TX * V + cc;
  double ta = 0.0, tb = 0.0;
  for (int gg = 0; gg < G; ++gg) {
    double acc = 0.0;
    if (loader) {
      const double* base = p.part + ((long long)gg * p.nchunks * p.C + (long long)slab * p.TX * V) * 2 + col;
      const long long rs = (long long)p.C * 2;
      int k = grp;
      for (; k + 3 * groups < p.nchunks; k += 4 * groups) {
        const double v0 = base[(long long)k * rs], v1 = base[(long long)(k + groups) * rs];
        const double v2 = base[(long long)(k + 2 * groups) * rs], v3 = base[(long long)(k + 3 * groups) * rs];
        acc += (v0 + v1) + (v2 + v3);
      }
      for (; k < p.nchunks; k += groups) acc += base[(long long)k * rs];
    }
    __syncthreads();                                     // previous group's colsum has been consumed
    if (loader && grp == 0) colsum[col] = acc;
    for (int r = 1; r < groups; ++r) {
      __syncthreads();
      if (loader && grp == r) colsum[col] += acc;
    }
    __syncthreads();
    if (cc < nch) {
      const double a = colsum[2 * cc], b = colsum[2 * cc + 1];
      const int idx = gg * p.C + c;
      if (MODE == FSV_RED_BWD) {
        p.o0[idx] = (float)a; p.o1[idx] = (float)b;
        ta += a; tb += b;
      } else if (MODE == FSV_RED_COLSUM) {
        p.o0[idx] = p.accumulate ? p.o0[idx] + (float)a : (float)a;
      } else {
        double mu = a / p.P;
        double var = b / p.P - mu * mu;
        if (var < 0.0) var = 0.0;
        p.o0[idx] = (float)mu;
        p.o1[idx] = (float)(1.0 / sqrt(var + (double)p.eps));
        if (p.o2 && G == 1) {
          const double cnt = (double)p.P * (double)p.rep;
          double unb = cnt > 1.0 ? var * (cnt / (cnt - 1.0)) : var;
          p.o2[c] = (1.f - p.momentum) * p.o2[c] + p.momentum * (float)mu;
          p.o3[c] = (1.f - p.momentum) * p.o3[c] + p.momentum * (float)unb;
        }
      }
    }
  }
  if (MODE == FSV_RED_BWD && cc < nch) {
    if (p.o2) p.o2[c] = (float)tb;
    if (p.o3) p.o3[c] = (float)ta;
  }
}

static inline void fsv_red_no_tail(RedP& p) {
  p.counter = nullptr; p.o0 = p.o1 = p.o2 = p.o3 = nullptr; p.eps = 0.f; p.momentum = 0.f; p.rep = 1; p.accumulate = 0;
}

// The fused second stage pays while the whole reduction is launch-bound: the last workgroup of a slab reads nchunks x slab
// partials alone, so the chunk count is capped and tensors above FSV_NORM_FUSE_MAX_MB keep two launches.  Measured on the C3
// bench step (profiles/r02_notes.md): threshold 0 (off) 58.9 ms, 1 MB 58.1, 4 MB 60.2, 16 MB 63.9, 64 MB 65.8 -> default 1 MB.
#define FSV_RED_FUSED_CHUNKS 32
#define FSV_RED_COUNTERS 64
static inline long long fsv_red_fuse_max_elems() {
  static long long v = -1;
  if (v < 0) {
    const char* e = getenv("FSV_NORM_FUSE_MAX_MB");
    double mb = e ? atof(e) : 1.0;
    v = (long long)(mb * 1024.0 * 1024.0 / 4.0);
  }
  return v;
}
static inline bool fsv_red_fused(const int* counters, int G, int P, int C) {
  return counters != nullptr && (long long)G * P * C <= fsv_red_fuse_max_elems();
}

template <int MODE>
static inline void fsv_launch_red(const RedPlan& pl, RedP p, int G, hipStream_t stream) {
  p.CU = pl.CU; p.TX = pl.TX; p.TY = pl.TY; p.rows_per_blk = pl.rows_per_blk; p.nchunks = pl.nchunks;
  dim3 grid(pl.nchunks, pl.nslabs, G);
  if (pl.V == 4) FSV_LAUNCH((fsv_red2_kernel<MODE, 4>), grid, dim3(256), stream, p);
  else FSV_LAUNCH((fsv_red2_kernel<MODE, 1>), grid, dim3(256), stream, p);
}

// sum of the per-chunk partials of one (g, c): executed by a whole wave, result valid in every lane
__device__ __forceinline__ void fsv_sum_chunks(const double* part, int g, int c, int C, int nchunks, double& a, double& b) {
  const int lane = threadIdx.x & 63;
  a = 0.0; b = 0.0;
  for (int k = lane; k < nchunks; k += 64) {
    const double* src = part + (((long long)g * nchunks + k) * C + c) * 2;
    a += src[0]; b += src[1];
  }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
}

// mean/rstd/var per (g, c); optional running-stat update (BatchNorm momentum semantics, unbiased running var)
// rep: the tensor the statistics describe holds every value `rep` times (statistics of a nearest x2 up-sampled tensor taken
// from its source: rep = 4) - mean and biased variance are those of the source, the unbiased correction counts P * rep values
__global__ __launch_bounds__(256) void fsv_stats_final_kernel(const double* part, float* mean, float* rstd, int G, int C,
                                                              int P, int nchunks, float eps, float* run_mean,
                                                              float* run_var, float momentum, int rep) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per (g, c)
  const bool ok = idx < G * C;
  const int g = ok ? idx / C : 0, c = ok ? idx - g * C : 0;
  double a, b;
  fsv_sum_chunks(part, g, c, C, nchunks, a, b);
  if (!ok || (threadIdx.x & 63) != 0) return;
  double mu = a / P;
  double var = b / P - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[idx] = (float)mu;
  rstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean && G == 1) {
    const double cnt = (double)P * (double)rep;
    double unb = cnt > 1.0 ? var * (cnt / (cnt - 1.0)) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

// ---- y = act(((x - mean) * rstd) * w + b) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsv_norm_apply_kernel(const float* x, const float* mean, const float* rstd,
                                                             const float* w, const float* b, float* y, long long total,
                                                             long long PC, int C, int act) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    int c = (int)(i % C);
    int g = (int)(i / PC);
    float xh = (x[i] - mean[g * C + c]) * rstd[g * C + c];
    float v = w ? xh * w[c] + b[c] : xh;
    y[i] = fsv_act(v, act);
  }
}

// the same for C % 4 == 0: four consecutive channels per work-item (16-byte loads / stores, one index division per four
// elements; 32-bit index arithmetic - the host takes this form only below 2^31 elements)
typedef _Float16 fsv_nh16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fsv_norm_apply4_kernel(const float* x, const float* mean, const float* rstd,
                                                              const float* w, const float* b, float* y, unsigned total4,
                                                              unsigned PC4, unsigned C4, int act, _Float16* yh) {
  const unsigned stride = gridDim.x * 256u;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += stride) {
    const unsigned c = (i % C4) * 4u, g = i / PC4;
    const unsigned gc = g * C4 * 4u + c;
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + gc), rs = *reinterpret_cast<const float4*>(rstd + gc);
    float4 v = make_float4((xv.x - mu.x) * rs.x, (xv.y - mu.y) * rs.y, (xv.z - mu.z) * rs.z, (xv.w - mu.w) * rs.w);
    if (w) {
      const float4 wv = *reinterpret_cast<const float4*>(w + c), bv = *reinterpret_cast<const float4*>(b + c);
      v = make_float4(v.x * wv.x + bv.x, v.y * wv.y + bv.y, v.z * wv.z + bv.z, v.w * wv.w + bv.w);
    }
    const float4 o = make_float4(fsv_act(v.x, act), fsv_act(v.y, act), fsv_act(v.z, act), fsv_act(v.w, act));
    *reinterpret_cast<float4*>(y + (size_t)i * 4) = o;
    if (yh) {                   // uniform: the half side output (fsv_common.h)
      fsv_nh16x4 h;
      h.x = (_Float16)o.x; h.y = (_Float16)o.y; h.z = (_Float16)o.z; h.w = (_Float16)o.w;
      *reinterpret_cast<fsv_nh16x4*>(yh + (size_t)i * 4) = h;
    }
  }
}

// (Round 3 tried a form with four 16-byte loads in flight per work-item - a block owning 1024 consecutive float4 - for this
// kernel and its backward twin: 213 vs 183 us for a forward + backward pair on a 67 MB tensor, +0.4 ms on the step, in-box;
// profiles/r03_notes.md.  The grid-stride form stays.)
// s1[g][c], s2[g][c]; optional affine grads dw[c] = sum_g s2, db[c] = sum_g s1 (one thread per channel)
__global__ __launch_bounds__(256) void fsv_norm_bwd_final_kernel(const double* part, float* s1, float* s2, float* dw,
                                                                 float* db, int G, int C, int nchunks) {
  const int cc = blockIdx.x * 4 + (threadIdx.x >> 6);        // one wave per channel
  const bool ok = cc < C;
  const int c = ok ? cc : 0;
  const bool lead = ok && (threadIdx.x & 63) == 0;
  double ta = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    double a, b;
    fsv_sum_chunks(part, g, c, C, nchunks, a, b);
    if (lead) { s1[g * C + c] = (float)a; s2[g * C + c] = (float)b; }
    ta += a; tb += b;
  }
  if (lead && dw) dw[c] = (float)tb;
  if (lead && db) db[c] = (float)ta;
}

// dx = w * rstd * (dyp - s1/P - xhat * s2/P)
__global__ __launch_bounds__(256) void fsv_norm_bwd_apply_kernel(const float* dy, const float* y, const float* x,
                                                                 const float* mean, const float* rstd, const float* w,
                                                                 const float* s1, const float* s2, float* dx,
                                                                 long long total, long long PC, int C, int P, int act,
                                                                 int fixed_stats) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  const float invP = 1.0f / (float)P;
  for (; i < total; i += stride) {
    int c = (int)(i % C);
    int g = (int)(i / PC);
    int gc = g * C + c;
    float rs = rstd[gc];
    float xh = (x[i] - mean[gc]) * rs;
    float d = fsv_act_grad(dy[i], y ? y[i] : 0.f, act);
    float wv = w ? w[c] : 1.f;
    // fixed_stats: eval-mode normalisation (running statistics are constants): only the affine scale remains
    dx[i] = fixed_stats ? wv * rs * d : wv * rs * (d - s1[gc] * invP - xh * (s2[gc] * invP));
  }
}

// four consecutive channels per work-item (C % 4 == 0, fewer than 2^31 elements)
__global__ __launch_bounds__(256) void fsv_norm_bwd_apply4_kernel(const float* dy, const float* y, const float* x,
                                                                  const float* mean, const float* rstd, const float* w,
                                                                  const float* s1, const float* s2, float* dx,
                                                                  unsigned total4, unsigned PC4, unsigned C4, int P, int act,
                                                                  int fixed_stats, _Float16* dxh) {
  const unsigned stride = gridDim.x * 256u;
  const float invP = 1.0f / (float)P;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += stride) {
    const unsigned c = (i % C4) * 4u, g = i / PC4;
    const unsigned gc = g * C4 * 4u + c;
    const size_t o = (size_t)i * 4;
    const float4 xv = *reinterpret_cast<const float4*>(x + o), dv = *reinterpret_cast<const float4*>(dy + o);
    const float4 yv = y ? *reinterpret_cast<const float4*>(y + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, da[4] = {dv.x, dv.y, dv.z, dv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float rs = rstd[gc + j];
      const float xh = (xa[j] - mean[gc + j]) * rs;
      const float d = fsv_act_grad(da[j], ya[j], act);
      const float wv = w ? w[c + j] : 1.f;
      r[j] = fixed_stats ? wv * rs * d : wv * rs * (d - s1[gc + j] * invP - xh * (s2[gc + j] * invP));
    }
    *reinterpret_cast<float4*>(dx + o) = make_float4(r[0], r[1], r[2], r[3]);
    if (dxh) {
      fsv_nh16x4 h;
      h.x = (_Float16)r[0]; h.y = (_Float16)r[1]; h.z = (_Float16)r[2]; h.w = (_Float16)r[3];
      *reinterpret_cast<fsv_nh16x4*>(dxh + o) = h;
    }
  }
}

// ---- column sums of a [G][P][C] tensor (bias gradients): out[g][c] ----------------------------------------------
__global__ __launch_bounds__(256) void fsv_colsum_final_kernel(const double* part, float* out, int G, int C, int nchunks, int accumulate) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool ok = idx < G * C;
  const int g = ok ? idx / C : 0, c = ok ? idx - g * C : 0;
  double a, b;
  fsv_sum_chunks(part, g, c, C, nchunks, a, b);
  if (ok && (threadIdx.x & 63) == 0) out[idx] = accumulate ? out[idx] + (float)a : (float)a;
}

// ---- grouped column sums: the bias gradients of every convolution of a backward pass in two launches -----------------------
// table[job][8] = { src ([P][C] NHWC rows), dst (float[C], accumulated into), part offset (doubles), P, C, rows_per_blk,
// nchunks, V | shared << 8 }.  tmap1 = (job, chunk, slab) triples, tmap2 = (job, block of 4 channels) pairs.  Same two-stage fp64 scheme
// and the same launch plan (fsv_red_plan) as fsv_colsum, so the sums are bit-identical to the per-layer path.
__global__ __launch_bounds__(256) void fsv_colsum_group_kernel(const long long* table, const int* tmap, double* part) {
  __shared__ float red[256 * 4];
  const int job = tmap[blockIdx.x * 3], chunk = tmap[blockIdx.x * 3 + 1], slab = tmap[blockIdx.x * 3 + 2];
  const long long* t = table + (long long)job * 8;
  const float* a = reinterpret_cast<const float*>(t[0]);
  double* pj = part + t[2];
  const int P = (int)t[3], C = (int)t[4], rpb = (int)t[5], V = (int)(t[7] & 0xff);
  const int CU = C / V;
  const int cap = (V == 4) ? 32 : 64;
  const int TX = CU < cap ? CU : cap, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int cu = slab * TX + tx;
  const bool active = ty < TY && cu < CU;
  const int r0 = chunk * rpb;
  const int r1 = (r0 + rpb < P) ? r0 + rpb : P;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int c0 = cu * V;
    for (int r = r0 + ty; r < r1; r += TY * 4) {
      float va[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + u * TY;
        const bool ok = rr < r1;
        const long long off = (long long)rr * C + c0;
        if (V == 4) {
          float4 q = ok ? *reinterpret_cast<const float4*>(a + off) : make_float4(0.f, 0.f, 0.f, 0.f);
          va[u][0] = q.x; va[u][1] = q.y; va[u][2] = q.z; va[u][3] = q.w;
        } else {
          va[u][0] = ok ? a[off] : 0.f; va[u][1] = va[u][2] = va[u][3] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += va[u][j];
    }
  }
  for (int j = 0; j < V; ++j) red[threadIdx.x * V + j] = s[j];
  __syncthreads();
  const int tt = threadIdx.x;
  if (tt < TX * V) {
    const int ux = tt / V, j = tt % V;
    if (slab * TX + ux < CU) {
      const int c = (slab * TX + ux) * V + j;
      double acc = 0.0;
      for (int yy = 0; yy < TY; ++yy) acc += (double)red[(yy * TX + ux) * V + j];
      pj[(long long)chunk * C + c] = acc;
    }
  }
}

__global__ __launch_bounds__(256) void fsv_colsum_group_final_kernel(const long long* table, const int* tmap, const double* part) {
  const int job = tmap[blockIdx.x * 2], cb = tmap[blockIdx.x * 2 + 1];
  const long long* t = table + (long long)job * 8;
  float* dst = reinterpret_cast<float*>(t[1]);
  const double* pj = part + t[2];
  const int C = (int)t[4], nchunks = (int)t[6];
  const bool shared = ((t[7] >> 8) & 1) != 0;          // another job of this launch adds into the same dst (a module used twice)
  const int c = cb * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool ok = c < C;
  double a = 0.0;
  if (ok)
    for (int k = lane; k < nchunks; k += 64) a += pj[(long long)k * C + c];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (ok && lane == 0) {
    if (shared) atomicAdd(&dst[c], (float)a); else dst[c] += (float)a;
  }
}

extern "C" {

// launch plan of the column reductions for a [P][C] tensor: out = {V, TX, nslabs, rows_per_blk, nchunks}
int fsv_colsum_plan(int P, int C, int* out) {
  if (P < 1 || C < 1 || !out) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(1, P, C);
  out[0] = pl.V; out[1] = pl.TX; out[2] = pl.nslabs; out[3] = pl.rows_per_blk; out[4] = pl.nchunks;
  return FSV_OK;
}

int fsv_colsum_grouped(const long long* table, int njobs, const int* tmap1, int nblk1, const int* tmap2, int nblk2,
                       double* part, hipStream_t stream) {
  if (!table || !tmap1 || !tmap2 || !part || njobs < 1 || nblk1 < 1 || nblk2 < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_colsum_group_kernel, dim3(nblk1), dim3(256), stream, table, tmap1, part);
  FSV_LAUNCH(fsv_colsum_group_final_kernel, dim3(nblk2), dim3(256), stream, table, tmap2, (const double*)part);
  return fsv_check_launch();
}

int fsv_norm_workspace_doubles(int G, int P, int C) {
  if (G < 1 || P < 1 || C < 1) return 2;
  RedPlan pl = fsv_red_plan(G, P, C);
  return G * pl.nchunks * C * 2;
}

int fsv_norm_stats_rep(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                       float* run_mean, float* run_var, float momentum, int rep, hipStream_t stream);
int fsv_norm_stats(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                   float* run_mean, float* run_var, float momentum, hipStream_t stream) {
  return fsv_norm_stats_rep(x, workspace, mean, rstd, G, P, C, eps, run_mean, run_var, momentum, 1, stream);
}

int fsv_norm_stats_rep(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                       float* run_mean, float* run_var, float momentum, int rep, hipStream_t stream) {
  if (!x || !workspace || !mean || !rstd || G < 1 || P < 1 || C < 1 || rep < 1) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(G, P, C);
  const int nchunks = pl.nchunks;
  RedP rp; rp.a = x; rp.y = nullptr; rp.x = nullptr; rp.mean = nullptr; rp.rstd = nullptr; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = 0; fsv_red_no_tail(rp);
  fsv_launch_red<FSV_RED_STATS>(pl, rp, G, stream);
  FSV_LAUNCH(fsv_stats_final_kernel, dim3(fsv_cdiv(G * C, 4)), dim3(256), stream, (const double*)workspace, mean, rstd,
             G, C, P, nchunks, eps, run_mean, run_var, momentum, rep);
  return fsv_check_launch();
}

// mean / rstd / running statistics from per-slot partial sums part[g][slot][c] = (sum x, sum x^2) that a producing kernel left
// behind (the gather-GEMM epilogue, fsv_conv_gather_fwd_stats): the second stage of fsv_norm_stats alone
int fsv_norm_stats_finish(const double* part, float* mean, float* rstd, int G, int P, int C, int nslots, float eps,
                          float* run_mean, float* run_var, float momentum, int rep, hipStream_t stream) {
  if (!part || !mean || !rstd || G < 1 || P < 1 || C < 1 || nslots < 1 || rep < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_stats_final_kernel, dim3(fsv_cdiv(G * C, 4)), dim3(256), stream, part, mean, rstd, G, C, P, nslots, eps,
             run_mean, run_var, momentum, rep);
  return fsv_check_launch();
}

static inline int fsv_ew_grid(long long total) {
  long long g = (total + 256 * 4 - 1) / (256 * 4);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

static inline bool fsv_ew_vec4(long long total, int C) {
  static int on = -1;            // FSV_NORM_VEC4=0: in-box A/B switch (profiles/r02_notes.md section 13)
  if (on < 0) { const char* e = getenv("FSV_NORM_VEC4"); on = (e && e[0] == '0') ? 0 : 1; }
  return on && C % 4 == 0 && total < (1LL << 31);
}

static inline void fsv_launch_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                                        const float* w, const float* s1, const float* s2, float* dx, long long total,
                                        long long PC, int C, int P, int act, int fixed_stats, void* dx_half,
                                        hipStream_t stream) {
  // (a half side output only exists in the four-channels-per-work-item form: fsv_half_side_ok was checked by the entry point)
  const bool v4 = dx_half ? true : fsv_ew_vec4(total, C);
  _Float16* dxh = reinterpret_cast<_Float16*>(dx_half);
  if (v4) {
    FSV_LAUNCH(fsv_norm_bwd_apply4_kernel, dim3(fsv_ew_grid(total / 4)), dim3(256), stream, dy, y, x, mean, rstd, w, s1, s2, dx,
               (unsigned)(total / 4), (unsigned)(PC / 4), (unsigned)(C / 4), P, act, fixed_stats, dxh);
  } else {
    FSV_LAUNCH(fsv_norm_bwd_apply_kernel, dim3(fsv_ew_grid(total)), dim3(256), stream, dy, y, x, mean, rstd, w, s1, s2, dx,
               total, PC, C, P, act, fixed_stats);
  }
}

// Half side output (`--amp`): y_half / dx_half != null -> the result is ALSO stored as IEEE half there, same element order (the
// consumer convolution reads that copy instead of converting the fp32 tensor).  An explicit, nullable argument of the entry point
// that writes the tensor - the library keeps no "armed" pointer between calls.  Only the four-channels-per-work-item kernels
// carry it: C % 4 == 0 and fewer than 2^31 elements, FSV_ERR_UNSUPPORTED (nothing launched) otherwise.
static inline bool fsv_half_side_ok(const void* h, long long total, int C) {
  return !h || (C % 4 == 0 && total < (1LL << 31));
}

int fsv_norm_apply(const float* x, const float* mean, const float* rstd, const float* w, const float* b, float* y,
                   int G, int P, int C, int act, void* y_half, hipStream_t stream) {
  if (!x || !mean || !rstd || !y || (w && !b)) return FSV_ERR_BAD_ARG;
  long long total = (long long)G * P * C;
  if (!fsv_half_side_ok(y_half, total, C)) return FSV_ERR_UNSUPPORTED;
  const bool v4 = y_half ? true : fsv_ew_vec4(total, C);
  _Float16* yh = reinterpret_cast<_Float16*>(y_half);
  if (v4) {
    FSV_LAUNCH(fsv_norm_apply4_kernel, dim3(fsv_ew_grid(total / 4)), dim3(256), stream, x, mean, rstd, w, b, y,
               (unsigned)(total / 4), (unsigned)((long long)P * C / 4), (unsigned)(C / 4), act, yh);
  } else {
    FSV_LAUNCH(fsv_norm_apply_kernel, dim3(fsv_ew_grid(total)), dim3(256), stream, x, mean, rstd, w, b, y, total,
               (long long)P * C, C, act);
  }
  return fsv_check_launch();
}

// dy: upstream gradient w.r.t. the activated output y (y may be null when act == none).  Produces dx and, when
// dw/db are non-null, the affine parameter gradients.  s1/s2: [G*C] scratch.
int fsv_norm_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                 double* workspace, float* s1, float* s2, float* dx, float* dw, float* db, int G, int P, int C, int act,
                 int fixed_stats, void* dx_half, hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !workspace || !s1 || !s2 || !dx) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_NONE && !y) return FSV_ERR_BAD_ARG;
  if (!fsv_half_side_ok(dx_half, (long long)G * P * C, C)) return FSV_ERR_UNSUPPORTED;
  RedPlan pl = fsv_red_plan(G, P, C);
  const int nchunks = pl.nchunks;
  RedP rp; rp.a = dy; rp.y = y; rp.x = x; rp.mean = mean; rp.rstd = rstd; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = act; fsv_red_no_tail(rp);
  fsv_launch_red<FSV_RED_BWD>(pl, rp, G, stream);
  FSV_LAUNCH(fsv_norm_bwd_final_kernel, dim3(fsv_cdiv(C, 4)), dim3(256), stream, (const double*)workspace, s1, s2, dw,
             db, G, C, nchunks);
  long long total = (long long)G * P * C;
  fsv_launch_bwd_apply(dy, y, x, mean, rstd, w, (const float*)s1, (const float*)s2, dx, total, (long long)P * C, C, P, act,
                       fixed_stats, dx_half, stream);
  return fsv_check_launch();
}

int fsv_colsum(const float* x, double* workspace, float* out, int G, int P, int C, int accumulate, hipStream_t stream) {
  if (!x || !workspace || !out) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(G, P, C);
  const int nchunks = pl.nchunks;
  RedP rp; rp.a = x; rp.y = nullptr; rp.x = nullptr; rp.mean = nullptr; rp.rstd = nullptr; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = 0; fsv_red_no_tail(rp);
  fsv_launch_red<FSV_RED_COLSUM>(pl, rp, G, stream);
  FSV_LAUNCH(fsv_colsum_final_kernel, dim3(fsv_cdiv(G * C, 4)), dim3(256), stream, (const double*)workspace, out, G,
             C, nchunks, accumulate);
  return fsv_check_launch();
}

// ---- one-launch forms: `counters` = zeroed ints (>= FSV_RED_COUNTERS, owned by the caller, left zeroed) -----------------------
int fsv_norm_stats_fused(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                         float* run_mean, float* run_var, float momentum, int rep, int* counters, hipStream_t stream) {
  if (!fsv_red_fused(counters, G, P, C))
    return fsv_norm_stats_rep(x, workspace, mean, rstd, G, P, C, eps, run_mean, run_var, momentum, rep, stream);
  if (!x || !workspace || !mean || !rstd || G < 1 || P < 1 || C < 1 || rep < 1) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(G, P, C, FSV_RED_FUSED_CHUNKS);
  // more channel slabs than ticket counters (C > 8192 on a small tensor): the two-launch form has no such limit
  if (pl.nslabs > FSV_RED_COUNTERS)
    return fsv_norm_stats_rep(x, workspace, mean, rstd, G, P, C, eps, run_mean, run_var, momentum, rep, stream);
  RedP rp; rp.a = x; rp.y = nullptr; rp.x = nullptr; rp.mean = nullptr; rp.rstd = nullptr; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = 0; fsv_red_no_tail(rp);
  rp.counter = counters; rp.o0 = mean; rp.o1 = rstd; rp.o2 = run_mean; rp.o3 = run_var; rp.eps = eps; rp.momentum = momentum;
  rp.rep = rep;
  fsv_launch_red<FSV_RED_STATS>(pl, rp, G, stream);
  return fsv_check_launch();
}

int fsv_norm_bwd_fused(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                       double* workspace, float* s1, float* s2, float* dx, float* dw, float* db, int G, int P, int C, int act,
                       int fixed_stats, int* counters, void* dx_half, hipStream_t stream) {
  if (!fsv_red_fused(counters, G, P, C))
    return fsv_norm_bwd(dy, y, x, mean, rstd, w, workspace, s1, s2, dx, dw, db, G, P, C, act, fixed_stats, dx_half, stream);
  if (!dy || !x || !mean || !rstd || !workspace || !s1 || !s2 || !dx) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_NONE && !y) return FSV_ERR_BAD_ARG;
  if (!fsv_half_side_ok(dx_half, (long long)G * P * C, C)) return FSV_ERR_UNSUPPORTED;
  RedPlan pl = fsv_red_plan(G, P, C, FSV_RED_FUSED_CHUNKS);
  if (pl.nslabs > FSV_RED_COUNTERS)
    return fsv_norm_bwd(dy, y, x, mean, rstd, w, workspace, s1, s2, dx, dw, db, G, P, C, act, fixed_stats, dx_half, stream);
  RedP rp; rp.a = dy; rp.y = y; rp.x = x; rp.mean = mean; rp.rstd = rstd; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = act; fsv_red_no_tail(rp);
  rp.counter = counters; rp.o0 = s1; rp.o1 = s2; rp.o2 = dw; rp.o3 = db;
  fsv_launch_red<FSV_RED_BWD>(pl, rp, G, stream);
  long long total = (long long)G * P * C;
  fsv_launch_bwd_apply(dy, y, x, mean, rstd, w, (const float*)s1, (const float*)s2, dx, total, (long long)P * C, C, P, act,
                       fixed_stats, dx_half, stream);
  return fsv_check_launch();
}

int fsv_colsum_fused(const float* x, double* workspace, float* out, int G, int P, int C, int accumulate, int* counters,
                     hipStream_t stream) {
  if (!fsv_red_fused(counters, G, P, C)) return fsv_colsum(x, workspace, out, G, P, C, accumulate, stream);
  if (!x || !workspace || !out) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(G, P, C, FSV_RED_FUSED_CHUNKS);
  if (pl.nslabs > FSV_RED_COUNTERS) return fsv_colsum(x, workspace, out, G, P, C, accumulate, stream);
  RedP rp; rp.a = x; rp.y = nullptr; rp.x = nullptr; rp.mean = nullptr; rp.rstd = nullptr; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = 0; fsv_red_no_tail(rp);
  rp.counter = counters; rp.o0 = out; rp.accumulate = accumulate;
  fsv_launch_red<FSV_RED_COLSUM>(pl, rp, G, stream);
  return fsv_check_launch();
}

}  // extern "C"

// ---- cross-replica BatchNorm ("SyncBN", opt-in): the statistics of one BatchNorm site split into reduce | exchange | finish ----
// The reference's multi-process path uses apex.parallel.SyncBatchNorm for every BatchNorm of the generator
// (models/networks/normalization.py:15,33,80): per-channel sums are exchanged between the ranks in the
// forward pass and the two gradient sums in the backward pass.  The exchange itself is the host's business (one small
// all-reduce of 2C doubles, torch.distributed over RCCL); these entry points are the device halves on either side of it.
__global__ __launch_bounds__(256) void fsv_sums_final_kernel(const double* part, double* sums, int C, int nchunks) {
  const int cc = blockIdx.x * 4 + (threadIdx.x >> 6);        // one wave per channel
  const bool ok = cc < C;
  double a, b;
  fsv_sum_chunks(part, 0, ok ? cc : 0, C, nchunks, a, b);
  if (ok && (threadIdx.x & 63) == 0) { sums[cc] = a; sums[C + cc] = b; }
}

// sums = {sum x [C], sum x^2 [C]} over `count` values per channel (all ranks together)
__global__ __launch_bounds__(256) void fsv_stats_from_sums_kernel(const double* sums, double count, float* mean, float* rstd,
                                                                  int C, float eps, float* run_mean, float* run_var,
                                                                  float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double mu = sums[c] / count;
  double var = sums[C + c] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean) {
    double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

extern "C" {

// local {sum x, sum x^2} of a [P][C] tensor as doubles [2C]
int fsv_norm_sums(const float* x, double* workspace, double* sums, int P, int C, hipStream_t stream) {
  if (!x || !workspace || !sums || P < 1 || C < 1) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(1, P, C);
  RedP rp; rp.a = x; rp.y = nullptr; rp.x = nullptr; rp.mean = nullptr; rp.rstd = nullptr; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = 0; fsv_red_no_tail(rp);
  fsv_launch_red<FSV_RED_STATS>(pl, rp, 1, stream);
  FSV_LAUNCH(fsv_sums_final_kernel, dim3(fsv_cdiv(C, 4)), dim3(256), stream, (const double*)workspace, sums, C, pl.nchunks);
  return fsv_check_launch();
}

// mean / rstd (and the running statistics, momentum semantics of nn.BatchNorm2d) from exchanged sums over `count` values
int fsv_norm_stats_from_sums(const double* sums, double count, float* mean, float* rstd, int C, float eps, float* run_mean,
                             float* run_var, float momentum, hipStream_t stream) {
  if (!sums || !mean || !rstd || C < 1 || !(count >= 1.0)) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_stats_from_sums_kernel, dim3(fsv_cdiv(C, 256)), dim3(256), stream, sums, count, mean, rstd, C, eps, run_mean,
             run_var, momentum);
  return fsv_check_launch();
}

// local {sum d, sum d * xhat} (d = dy * act'(y)) as doubles [2C]: the two sums of the BatchNorm backward before the exchange
int fsv_norm_bwd_sums(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                      double* workspace, double* sums, int P, int C, int act, hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !workspace || !sums || P < 1 || C < 1) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_NONE && !y) return FSV_ERR_BAD_ARG;
  RedPlan pl = fsv_red_plan(1, P, C);
  RedP rp; rp.a = dy; rp.y = y; rp.x = x; rp.mean = mean; rp.rstd = rstd; rp.part = workspace;
  rp.P = P; rp.C = C; rp.act = act; fsv_red_no_tail(rp);
  fsv_launch_red<FSV_RED_BWD>(pl, rp, 1, stream);
  FSV_LAUNCH(fsv_sums_final_kernel, dim3(fsv_cdiv(C, 4)), dim3(256), stream, (const double*)workspace, sums, C, pl.nchunks);
  return fsv_check_launch();
}

// dx = w * rstd * (d - s1 / count - xhat * s2 / count) over the local [P][C] tensor with the exchanged sums s1, s2 [C]
int fsv_norm_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* s1, const float* s2, float* dx, int P, int C, int count, int act, void* dx_half,
                       hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !s1 || !s2 || !dx || P < 1 || C < 1 || count < 1) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_NONE && !y) return FSV_ERR_BAD_ARG;
  long long total = (long long)P * C;
  if (!fsv_half_side_ok(dx_half, total, C)) return FSV_ERR_UNSUPPORTED;
  fsv_launch_bwd_apply(dy, y, x, mean, rstd, w, s1, s2, dx, total, total, C, count, act, 0, dx_half, stream);
  return fsv_check_launch();
}

}  // extern "C"
