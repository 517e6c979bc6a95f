// Spectral normalisation (torch.nn.utils.spectral_norm, one power iteration per training forward) for every
// conv / linear weight of G and D (reference: models/networks/architecture.py:60,81-84, generator.py:106-109,
// normalization.py:65).  W is viewed as a [R][Cc] matrix (R = out channels):
//     v = normalize(W^T u);  u = normalize(W v);  sigma = u . (W v);  W_sn = W / sigma
// u and v are persistent buffers updated in place (also under no_grad: the reference runs G forward twice per
// iteration in train() mode).  1/sigma is left on the device and folded into the weight re-arrangement
// (fsv_prep_weight scale pointer), so W_sn is never materialised for the forward pass.
// Backward (u, v constant):  dW = (dW_sn - <dW_sn, W_sn> u v^T) / sigma.
#include "fsv_common.h"

// W^T u in two deterministic steps (round 3; the row slabs used to be added into t with fp32 atomics, whose arrival order
// changed v, u and sigma in the last bits from run to run - enough to flip exact cancellations that sit on a LeakyReLU kink
// downstream: the two-outcome gradients of the C1 full-size step, tests/test_fullsize_gpu.py):
//   part[slab][j] = sum_{i in row slab} W[i][j] u[i]          fsv_sn_gemv_t_kernel
//   t[j]          = sum_{slab} part[slab][j], ascending slabs  fsv_sn_sum_t_kernel
#define FSV_SN_SLAB 64
__global__ __launch_bounds__(256) void fsv_sn_gemv_t_kernel(const float* W, const float* u, float* part, int R, int Cc) {
  int j = blockIdx.x * 256 + threadIdx.x;
  int r0 = blockIdx.y * FSV_SN_SLAB;
  int r1 = (r0 + FSV_SN_SLAB < R) ? r0 + FSV_SN_SLAB : R;
  if (j >= Cc) return;
  float acc = 0.f;
  for (int i = r0; i < r1; ++i) acc += W[(long long)i * Cc + j] * u[i];
  part[(long long)blockIdx.y * Cc + j] = acc;
}

__global__ __launch_bounds__(256) void fsv_sn_sum_t_kernel(const float* part, float* t, int Cc, int nslab) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Cc) return;
  float acc = part[j];
  for (int k = 1; k < nslab; ++k) acc += part[(long long)k * Cc + j];
  t[j] = acc;
}

// s[i] = sum_j W[i][j] t[j]      one wave per row
__global__ __launch_bounds__(256) void fsv_sn_gemv_kernel(const float* W, const float* t, float* s, int R, int Cc) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool ok = row < R;
  const float* w = W + (long long)(ok ? row : 0) * Cc;
  float acc = 0.f;
  if (ok) for (int j = lane; j < Cc; j += 64) acc += w[j] * t[j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (ok && lane == 0) s[row] = acc;
}

// single block: norms, u, v, sigma.  out[0] = sigma, out[1] = 1/sigma
__global__ __launch_bounds__(256) void fsv_sn_finalize_kernel(const float* t, const float* s, float* u, float* v, float* out,
                                                              int R, int Cc, float eps) {
  __shared__ float red[256];
  float a = 0.f;
  for (int j = threadIdx.x; j < Cc; j += 256) a += t[j] * t[j];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float nt = fmaxf(sqrtf(red[0]), eps);
  __syncthreads();
  for (int j = threadIdx.x; j < Cc; j += 256) v[j] = t[j] / nt;
  // W v = s / nt
  float b = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) { float wv = s[i] / nt; b += wv * wv; }
  red[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float ns = fmaxf(sqrtf(red[0]), eps);
  __syncthreads();
  float c = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) { float wv = s[i] / nt; float ui = wv / ns; u[i] = ui; c += ui * wv; }
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { out[0] = red[0]; out[1] = 1.f / red[0]; }
}

// eval-mode sigma (no power iteration): sigma = u . (W v)
__global__ __launch_bounds__(256) void fsv_sn_sigma_kernel(const float* s, const float* u, float* out, int R) {
  __shared__ float red[256];
  float c = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) c += u[i] * s[i];
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { out[0] = red[0]; out[1] = 1.f / red[0]; }
}

// partial dot products <A, B> -> part[block] (double)
__global__ __launch_bounds__(256) void fsv_dot_partial_kernel(const float* a, const float* b, double* part, long long n) {
  __shared__ double red[256];
  const long long stride = (long long)gridDim.x * 256;
  float acc = 0.f; double dacc = 0.0; int cnt = 0;
  const long long n4 = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0 ? n / 4 : 0;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 x = a4[i], y = b4[i];
    acc += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    if (++cnt == 16) { dacc += (double)acc; acc = 0.f; cnt = 0; }
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += a[i] * b[i];
  red[threadIdx.x] = dacc + (double)acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// dW[i][j] = inv_sigma * (dWsn[i][j] - inv_sigma * dot * u[i] * v[j]),  dot = <dWsn, W>
__global__ __launch_bounds__(256) void fsv_sn_bwd_kernel(const float* dWsn, const double* part, int nparts, const float* u,
                                                         const float* v, const float* sig, float* dW, int R, int Cc, int accumulate) {
  __shared__ float sdot;
  if (threadIdx.x == 0) { double d = 0.0; for (int k = 0; k < nparts; ++k) d += part[k]; sdot = (float)d; }
  __syncthreads();
  const float inv = sig[1];
  const float coef = inv * sdot;
  const long long total = (long long)R * Cc;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    int r = (int)(i / Cc), c = (int)(i - (long long)r * Cc);
    const float g = inv * (dWsn[i] - coef * u[r] * v[c]);
    dW[i] = accumulate ? dW[i] + g : g;
  }
}

extern "C" {

// scratch: float[fsv_sn_scratch_floats(R, Cc)] device scratch (R + Cc * (1 + ceil(R / 64))); sig: float[2] -> (sigma, 1/sigma)
int fsv_sn_scratch_floats(int R, int Cc) { return R + Cc * (1 + fsv_cdiv(R, FSV_SN_SLAB)); }

int fsv_sn_power_iter(const float* W, float* u, float* v, float* scratch, float* sig, int R, int Cc, float eps,
                      int training, hipStream_t stream) {
  if (!W || !u || !v || !scratch || !sig || R < 1 || Cc < 1) return FSV_ERR_BAD_ARG;
  float* t = scratch;          // [Cc]
  float* s = scratch + Cc;     // [R]
  float* part = s + R;         // [ceil(R / 64)][Cc]
  if (training) {
    const int nslab = fsv_cdiv(R, FSV_SN_SLAB);
    dim3 g(fsv_cdiv(Cc, 256), nslab);
    FSV_LAUNCH(fsv_sn_gemv_t_kernel, g, dim3(256), stream, W, (const float*)u, part, R, Cc);
    FSV_LAUNCH(fsv_sn_sum_t_kernel, dim3(fsv_cdiv(Cc, 256)), dim3(256), stream, (const float*)part, t, Cc, nslab);
    FSV_LAUNCH(fsv_sn_gemv_kernel, dim3(fsv_cdiv(R, 4)), dim3(256), stream, W, (const float*)t, s, R, Cc);
    FSV_LAUNCH(fsv_sn_finalize_kernel, dim3(1), dim3(256), stream, (const float*)t, (const float*)s, u, v, sig, R, Cc, eps);
  } else {
    FSV_LAUNCH(fsv_sn_gemv_kernel, dim3(fsv_cdiv(R, 4)), dim3(256), stream, W, (const float*)v, s, R, Cc);
    FSV_LAUNCH(fsv_sn_sigma_kernel, dim3(1), dim3(256), stream, (const float*)s, (const float*)u, sig, R);
  }
  return fsv_check_launch();
}

// part: double[256] scratch
int fsv_sn_backward(const float* dWsn, const float* W, const float* u, const float* v, const float* sig, double* part,
                    float* dW, int R, int Cc, int accumulate, hipStream_t stream) {
  if (!dWsn || !W || !u || !v || !sig || !part || !dW) return FSV_ERR_BAD_ARG;
  long long n = (long long)R * Cc;
  int nparts = (int)((n + 256 * 16 - 1) / (256 * 16));
  if (nparts > 256) nparts = 256;
  if (nparts < 1) nparts = 1;
  FSV_LAUNCH(fsv_dot_partial_kernel, dim3(nparts), dim3(256), stream, dWsn, W, part, n);
  long long g = (n + 1023) / 1024;
  if (g > 4096) g = 4096;
  FSV_LAUNCH(fsv_sn_bwd_kernel, dim3((unsigned)g), dim3(256), stream, dWsn, (const double*)part, nparts, u, v, sig, dW, R, Cc, accumulate);
  return fsv_check_launch();
}

}  // extern "C"

// ---- batched power iteration: every spectral-normalised layer of a network in three launches -------------------
// A network's weights do not change during its forward pass, so the per-layer power iterations of the reference
// (one forward pre-hook per module) can run up front as one grouped pass.  Layer geometry comes from device-side
// descriptor arrays built once by the host; `tmap` maps a flat block index to (layer, tile).
struct SnBatch {
  const long long* W;      // [L] device pointers (as integers)
  const long long* u;
  const long long* v;
  const int* rows;
  const int* cols;
  const int* t_off;        // offsets into scratch: t[cols]
  const int* s_off;        //                       s[rows]
  float* scratch;
  float* sig;              // [L][2]
  float* snap;             // per-pass copies of the new u / v (kept by autograd; the persistent buffers move on)
  const int* u_off;
  const int* v_off;
};

__global__ __launch_bounds__(256) void fsv_snb_gemv_t_kernel(SnBatch b, const int* tmap) {
  const int layer = tmap[blockIdx.x * 2], tile = tmap[blockIdx.x * 2 + 1];
  const int R = b.rows[layer], Cc = b.cols[layer];
  const int ncb = (Cc + 255) / 256;
  const int cb = tile % ncb, rs = tile / ncb;
  const float* W = reinterpret_cast<const float*>(b.W[layer]);
  const float* u = reinterpret_cast<const float*>(b.u[layer]);
  float* part = b.scratch + b.t_off[layer] + Cc;        // the layer's t region: t[Cc], then part[ceil(R / 64)][Cc]
  const int j = cb * 256 + threadIdx.x;
  const int r0 = rs * FSV_SN_SLAB, r1 = (r0 + FSV_SN_SLAB < R) ? r0 + FSV_SN_SLAB : R;
  if (j >= Cc) return;
  float acc = 0.f;
  for (int i = r0; i < r1; ++i) acc += W[(long long)i * Cc + j] * u[i];
  part[(long long)rs * Cc + j] = acc;
}

// t[j] = sum over the row slabs in ascending order (same block map as fsv_snb_gemv_t_kernel: the blocks of slab 0 do the work)
__global__ __launch_bounds__(256) void fsv_snb_sum_t_kernel(SnBatch b, const int* tmap) {
  const int layer = tmap[blockIdx.x * 2], tile = tmap[blockIdx.x * 2 + 1];
  const int R = b.rows[layer], Cc = b.cols[layer];
  const int ncb = (Cc + 255) / 256;
  if (tile >= ncb) return;                                // row slab != 0
  const int j = tile * 256 + threadIdx.x;
  if (j >= Cc) return;
  float* t = b.scratch + b.t_off[layer];
  const float* part = t + Cc;
  const int nslab = (R + FSV_SN_SLAB - 1) / FSV_SN_SLAB;
  float acc = part[j];
  for (int k = 1; k < nslab; ++k) acc += part[(long long)k * Cc + j];
  t[j] = acc;
}

__global__ __launch_bounds__(256) void fsv_snb_gemv_kernel(SnBatch b, const int* tmap) {
  const int layer = tmap[blockIdx.x * 2], grp = tmap[blockIdx.x * 2 + 1];
  const int R = b.rows[layer], Cc = b.cols[layer];
  const float* W = reinterpret_cast<const float*>(b.W[layer]);
  const float* t = b.scratch + b.t_off[layer];
  float* s = b.scratch + b.s_off[layer];
  const int lane = threadIdx.x & 63;
  const int row = grp * 4 + (threadIdx.x >> 6);
  const bool ok = row < R;
  const float* w = W + (long long)(ok ? row : 0) * Cc;
  float acc = 0.f;
  if (ok) for (int j = lane; j < Cc; j += 64) acc += w[j] * t[j];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (ok && lane == 0) s[row] = acc;
}

__global__ __launch_bounds__(256) void fsv_snb_finalize_kernel(SnBatch b, float eps) {
  __shared__ float red[256];
  const int layer = blockIdx.x;
  const int R = b.rows[layer], Cc = b.cols[layer];
  const float* t = b.scratch + b.t_off[layer];
  const float* s = b.scratch + b.s_off[layer];
  float* u = reinterpret_cast<float*>(b.u[layer]);
  float* v = reinterpret_cast<float*>(b.v[layer]);
  float a = 0.f;
  for (int j = threadIdx.x; j < Cc; j += 256) a += t[j] * t[j];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float nt = fmaxf(sqrtf(red[0]), eps);
  __syncthreads();
  float* vs = b.snap + b.v_off[layer];
  float* us = b.snap + b.u_off[layer];
  for (int j = threadIdx.x; j < Cc; j += 256) { float vj = t[j] / nt; v[j] = vj; vs[j] = vj; }
  float bb = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) { float wv = s[i] / nt; bb += wv * wv; }
  red[threadIdx.x] = bb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float ns = fmaxf(sqrtf(red[0]), eps);
  __syncthreads();
  float c = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) { float wv = s[i] / nt; float ui = wv / ns; u[i] = ui; us[i] = ui; c += ui * wv; }
  red[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) { b.sig[layer * 2] = red[0]; b.sig[layer * 2 + 1] = 1.f / red[0]; }
}

extern "C" int fsv_sn_power_iter_batched(const long long* W, const long long* u, const long long* v, const int* rows,
                                         const int* cols, const int* t_off, const int* s_off, float* scratch,
                                         long long scratch_floats, float* sig, float* snap, const int* u_off,
                                         const int* v_off, int nlayers, const int* tmap_t,
                                         int nblk_t, const int* tmap_s, int nblk_s, float eps, hipStream_t stream) {
  if (!W || !u || !v || !rows || !cols || !t_off || !s_off || !scratch || !sig || !snap || !u_off || !v_off || nlayers < 1)
    return FSV_ERR_BAD_ARG;
  SnBatch b;
  b.W = W; b.u = u; b.v = v; b.rows = rows; b.cols = cols; b.t_off = t_off; b.s_off = s_off; b.scratch = scratch; b.sig = sig;
  b.snap = snap; b.u_off = u_off; b.v_off = v_off;
  (void)scratch_floats;       // every value that is read is written first: nothing to zero
  FSV_LAUNCH(fsv_snb_gemv_t_kernel, dim3(nblk_t), dim3(256), stream, b, tmap_t);
  FSV_LAUNCH(fsv_snb_sum_t_kernel, dim3(nblk_t), dim3(256), stream, b, tmap_t);
  FSV_LAUNCH(fsv_snb_gemv_kernel, dim3(nblk_s), dim3(256), stream, b, tmap_s);
  FSV_LAUNCH(fsv_snb_finalize_kernel, dim3(nlayers), dim3(256), stream, b, eps);
  return fsv_check_launch();
}
