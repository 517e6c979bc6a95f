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

#define FSV_RED_ROWS 2048   // pixels per partial-reduction block

// thread mapping for [P][C] tiles: q = channel quad (or scalar channel), rows strided by 256/Q
struct RowMap {
  int Q, rows_per_pass, q, r;
  bool active;
};
__device__ __forceinline__ RowMap fsv_rowmap(int ncolunits, int tid) {
  RowMap m;
  m.Q = ncolunits < 256 ? ncolunits : 256;
  m.rows_per_pass = 256 / m.Q;
  m.q = tid % m.Q;
  m.r = tid / m.Q;
  m.active = m.r < m.rows_per_pass;
  return m;
}

// ---- forward statistics -------------------------------------------------------------------------------------
// part[g][chunk][c][2] (double): sum, sum of squares over the chunk's pixel range
__global__ __launch_bounds__(256) void fsv_stats_partial_kernel(const float* x, double* part, int P, int C, int nchunks) {
  __shared__ double red[256 * 2];
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * FSV_RED_ROWS;
  const int p1 = (p0 + FSV_RED_ROWS < P) ? p0 + FSV_RED_ROWS : P;
  const float* xg = x + (long long)g * P * C;
  const RowMap m = fsv_rowmap(C, threadIdx.x);
  for (int c0 = 0; c0 < C; c0 += m.Q) {
    const int c = c0 + m.q;
    float s = 0.f, s2 = 0.f;
    if (m.active && c < C)
      for (int p = p0 + m.r; p < p1; p += m.rows_per_pass) {
        float v = xg[(long long)p * C + c];
        s += v; s2 += v * v;
      }
    red[threadIdx.x * 2] = (double)s; red[threadIdx.x * 2 + 1] = (double)s2;
    __syncthreads();
    if (threadIdx.x < m.Q && c0 + (int)threadIdx.x < C) {
      double a = 0.0, b = 0.0;
      for (int r = 0; r < m.rows_per_pass; ++r) { a += red[(r * m.Q + threadIdx.x) * 2]; b += red[(r * m.Q + threadIdx.x) * 2 + 1]; }
      double* dst = part + (((long long)g * nchunks + chunk) * C + c0 + threadIdx.x) * 2;
      dst[0] = a; dst[1] = b;
    }
    __syncthreads();
  }
}

// mean/rstd/var per (g, c); optional running-stat update (BatchNorm momentum semantics, unbiased running var)
__global__ __launch_bounds__(256) void fsv_stats_final_kernel(const double* part, float* mean, float* rstd, int G, int C,
                                                              int P, int nchunks, float eps, float* run_mean,
                                                              float* run_var, float momentum) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= G * C) return;
  int g = idx / C, c = idx - g * C;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nchunks; ++k) {
    const double* src = part + (((long long)g * nchunks + k) * C + c) * 2;
    a += src[0]; b += src[1];
  }
  double mu = a / P;
  double var = b / P - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[idx] = (float)mu;
  rstd[idx] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean && G == 1) {
    double unb = P > 1 ? var * ((double)P / (double)(P - 1)) : var;
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

// ---- backward: reductions sum(dyp), sum(dyp * xhat) with dyp = dy * act'(y) -----------------------------------
__device__ __forceinline__ float fsv_act_grad(float dy, float y, int act) {
  if (act == FSV_ACT_LRELU) return y > 0.f ? dy : 0.2f * dy;
  if (act == FSV_ACT_TANH) return dy * (1.f - y * y);
  if (act == FSV_ACT_SIGMOID) return dy * y * (1.f - y);
  return dy;
}

__global__ __launch_bounds__(256) void fsv_norm_bwd_partial_kernel(const float* dy, const float* y, const float* x,
                                                                   const float* mean, const float* rstd, double* part,
                                                                   int P, int C, int nchunks, int act) {
  __shared__ double red[256 * 2];
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * FSV_RED_ROWS;
  const int p1 = (p0 + FSV_RED_ROWS < P) ? p0 + FSV_RED_ROWS : P;
  const long long goff = (long long)g * P * C;
  const RowMap m = fsv_rowmap(C, threadIdx.x);
  for (int c0 = 0; c0 < C; c0 += m.Q) {
    const int c = c0 + m.q;
    float s = 0.f, s2 = 0.f;
    if (m.active && c < C) {
      const float mu = mean[g * C + c], rs = rstd[g * C + c];
      for (int p = p0 + m.r; p < p1; p += m.rows_per_pass) {
        long long i = goff + (long long)p * C + c;
        float d = fsv_act_grad(dy[i], y ? y[i] : 0.f, act);
        s += d; s2 += d * ((x[i] - mu) * rs);
      }
    }
    red[threadIdx.x * 2] = (double)s; red[threadIdx.x * 2 + 1] = (double)s2;
    __syncthreads();
    if (threadIdx.x < m.Q && c0 + (int)threadIdx.x < C) {
      double a = 0.0, b = 0.0;
      for (int r = 0; r < m.rows_per_pass; ++r) { a += red[(r * m.Q + threadIdx.x) * 2]; b += red[(r * m.Q + threadIdx.x) * 2 + 1]; }
      double* dst = part + (((long long)g * nchunks + chunk) * C + c0 + threadIdx.x) * 2;
      dst[0] = a; dst[1] = b;
    }
    __syncthreads();
  }
}

// s1[g][c], s2[g][c]; optional affine grads dw[c] = sum_g s2, db[c] = sum_g s1 (one thread per channel)
__global__ __launch_bounds__(256) void fsv_norm_bwd_final_kernel(const double* part, float* s1, float* s2, float* dw,
                                                                 float* db, int G, int C, int nchunks) {
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double ta = 0.0, tb = 0.0;
  for (int g = 0; g < G; ++g) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      const double* src = part + (((long long)g * nchunks + k) * C + c) * 2;
      a += src[0]; b += src[1];
    }
    s1[g * C + c] = (float)a; s2[g * C + c] = (float)b;
    ta += a; tb += b;
  }
  if (dw) dw[c] = (float)tb;
  if (db) db[c] = (float)ta;
}

// dx = w * rstd * (dyp - s1/P - xhat * s2/P)
__global__ __launch_bounds__(256) void fsv_norm_bwd_apply_kernel(const float* dy, const float* y, const float* x,
                                                                 const float* mean, const float* rstd, const float* w,
                                                                 const float* s1, const float* s2, float* dx,
                                                                 long long total, long long PC, int C, int P, int act) {
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
    dx[i] = wv * rs * (d - s1[gc] * invP - xh * (s2[gc] * invP));
  }
}

// ---- column sums of a [G][P][C] tensor (bias gradients): out[g][c] ----------------------------------------------
__global__ __launch_bounds__(256) void fsv_colsum_partial_kernel(const float* x, double* part, int P, int C, int nchunks) {
  __shared__ double red[256];
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * FSV_RED_ROWS;
  const int p1 = (p0 + FSV_RED_ROWS < P) ? p0 + FSV_RED_ROWS : P;
  const float* xg = x + (long long)g * P * C;
  const RowMap m = fsv_rowmap(C, threadIdx.x);
  for (int c0 = 0; c0 < C; c0 += m.Q) {
    const int c = c0 + m.q;
    float s = 0.f;
    if (m.active && c < C)
      for (int p = p0 + m.r; p < p1; p += m.rows_per_pass) s += xg[(long long)p * C + c];
    red[threadIdx.x] = (double)s;
    __syncthreads();
    if (threadIdx.x < m.Q && c0 + (int)threadIdx.x < C) {
      double a = 0.0;
      for (int r = 0; r < m.rows_per_pass; ++r) a += red[r * m.Q + threadIdx.x];
      part[((long long)g * nchunks + chunk) * C + c0 + threadIdx.x] = a;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void fsv_colsum_final_kernel(const double* part, float* out, int G, int C, int nchunks) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= G * C) return;
  int g = idx / C, c = idx - g * C;
  double a = 0.0;
  for (int k = 0; k < nchunks; ++k) a += part[((long long)g * nchunks + k) * C + c];
  out[idx] = (float)a;
}

extern "C" {

int fsv_norm_workspace_doubles(int G, int P, int C) {
  return G * fsv_cdiv(P, FSV_RED_ROWS) * C * 2;
}

int fsv_norm_stats(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                   float* run_mean, float* run_var, float momentum, hipStream_t stream) {
  if (!x || !workspace || !mean || !rstd || G < 1 || P < 1 || C < 1) return FSV_ERR_BAD_ARG;
  int nchunks = fsv_cdiv(P, FSV_RED_ROWS);
  FSV_LAUNCH(fsv_stats_partial_kernel, dim3(nchunks, G), dim3(256), stream, x, workspace, P, C, nchunks);
  FSV_LAUNCH(fsv_stats_final_kernel, dim3(fsv_cdiv(G * C, 256)), dim3(256), stream, (const double*)workspace, mean, rstd,
             G, C, P, nchunks, eps, run_mean, run_var, momentum);
  return fsv_check_launch();
}

static inline int fsv_ew_grid(long long total) {
  long long g = (total + 256 * 4 - 1) / (256 * 4);
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

int fsv_norm_apply(const float* x, const float* mean, const float* rstd, const float* w, const float* b, float* y,
                   int G, int P, int C, int act, hipStream_t stream) {
  if (!x || !mean || !rstd || !y || (w && !b)) return FSV_ERR_BAD_ARG;
  long long total = (long long)G * P * C;
  FSV_LAUNCH(fsv_norm_apply_kernel, dim3(fsv_ew_grid(total)), dim3(256), stream, x, mean, rstd, w, b, y, total,
             (long long)P * C, C, act);
  return fsv_check_launch();
}

// dy: upstream gradient w.r.t. the activated output y (y may be null when act == none).  Produces dx and, when
// dw/db are non-null, the affine parameter gradients.  s1/s2: [G*C] scratch.
int fsv_norm_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                 double* workspace, float* s1, float* s2, float* dx, float* dw, float* db, int G, int P, int C, int act,
                 hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !workspace || !s1 || !s2 || !dx) return FSV_ERR_BAD_ARG;
  if (act != FSV_ACT_NONE && !y) return FSV_ERR_BAD_ARG;
  int nchunks = fsv_cdiv(P, FSV_RED_ROWS);
  FSV_LAUNCH(fsv_norm_bwd_partial_kernel, dim3(nchunks, G), dim3(256), stream, dy, y, x, mean, rstd, workspace, P, C,
             nchunks, act);
  FSV_LAUNCH(fsv_norm_bwd_final_kernel, dim3(fsv_cdiv(C, 256)), dim3(256), stream, (const double*)workspace, s1, s2, dw,
             db, G, C, nchunks);
  long long total = (long long)G * P * C;
  FSV_LAUNCH(fsv_norm_bwd_apply_kernel, dim3(fsv_ew_grid(total)), dim3(256), stream, dy, y, x, mean, rstd, w,
             (const float*)s1, (const float*)s2, dx, total, (long long)P * C, C, P, act);
  return fsv_check_launch();
}

int fsv_colsum(const float* x, double* workspace, float* out, int G, int P, int C, hipStream_t stream) {
  if (!x || !workspace || !out) return FSV_ERR_BAD_ARG;
  int nchunks = fsv_cdiv(P, FSV_RED_ROWS);
  FSV_LAUNCH(fsv_colsum_partial_kernel, dim3(nchunks, G), dim3(256), stream, x, workspace, P, C, nchunks);
  FSV_LAUNCH(fsv_colsum_final_kernel, dim3(fsv_cdiv(G * C, 256)), dim3(256), stream, (const double*)workspace, out, G,
             C, nchunks);
  return fsv_check_launch();
}

}  // extern "C"
