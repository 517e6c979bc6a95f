// Small HBM-bound helpers of the G/D step: nearest x2 up-sampling (generator.py:124 `self.up`, nn.Upsample in
// FlowGenerator / LabelEmbedder), activation forward/backward, and the fused Adam update on flat parameter
// buffers (reference optimiser: models/base_model.py:39-48, torch.optim.Adam with TTUR betas (0, 0.999)).
#include "fsv_common.h"

static inline int fsv_grid_for(long long units) {
  long long g = (units + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

// y[n, 2h+a, 2w+b, :] = x[n, h, w, :]  -- one work-item per output float4 (or scalar when C % 4 != 0)
template <int V>
__global__ __launch_bounds__(256) void fsv_up2x_fwd_kernel(const float* x, float* y, int N, int H, int W, int C) {
  const int CQ = C / V;
  const long long total = (long long)N * (2 * H) * (2 * W) * CQ;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    int cq = (int)(i % CQ);
    long long pix = i / CQ;
    int ox = (int)(pix % (2 * W));
    long long t = pix / (2 * W);
    int oy = (int)(t % (2 * H));
    int n = (int)(t / (2 * H));
    long long src = ((((long long)n * H + (oy >> 1)) * W + (ox >> 1)) * CQ + cq) * V;
    if (V == 4) *reinterpret_cast<float4*>(y + i * 4) = *reinterpret_cast<const float4*>(x + src);
    else y[i] = x[src];
  }
}

// dx[n, h, w, :] = sum_{a,b} dy[n, 2h+a, 2w+b, :]
template <int V>
__global__ __launch_bounds__(256) void fsv_up2x_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C) {
  const int CQ = C / V;
  const long long total = (long long)N * H * W * CQ;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  const long long rowq = (long long)(2 * W) * CQ;
  for (; i < total; i += stride) {
    int cq = (int)(i % CQ);
    long long pix = i / CQ;
    int w = (int)(pix % W);
    long long t = pix / W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    long long s00 = ((((long long)n * 2 * H + 2 * h) * (2 * W) + 2 * w) * CQ + cq);
    if (V == 4) {
      const float4* d = reinterpret_cast<const float4*>(dy);
      float4 a = d[s00], b = d[s00 + CQ], c = d[s00 + rowq], e = d[s00 + rowq + CQ];
      *reinterpret_cast<float4*>(dx + i * 4) = make_float4((a.x + b.x) + (c.x + e.x), (a.y + b.y) + (c.y + e.y),
                                                           (a.z + b.z) + (c.z + e.z), (a.w + b.w) + (c.w + e.w));
    } else {
      dx[i] = (dy[s00] + dy[s00 + CQ]) + (dy[s00 + rowq] + dy[s00 + rowq + CQ]);
    }
  }
}

__global__ __launch_bounds__(256) void fsv_act_fwd_kernel(const float* x, float* y, long long total, int act) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) y[i] = fsv_act(x[i], act);
}

__global__ __launch_bounds__(256) void fsv_act_bwd_kernel(const float* dy, const float* y, float* dx, long long total,
                                                          int act, float scale) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < total; i += stride) {
    float d = dy[i] * scale, v = y[i];
    if (act == FSV_ACT_LRELU) d = v > 0.f ? d : 0.2f * d;
    else if (act == FSV_ACT_TANH) d = d * (1.f - v * v);
    else if (act == FSV_ACT_SIGMOID) d = d * v * (1.f - v);
    else if (act == FSV_ACT_RELU) d = v > 0.f ? d : 0.f;
    else if (act == FSV_ACT_LRELU01) d = v > 0.f ? d : 0.1f * d;
    dx[i] = d;
  }
}

// ---- softmax over the channel (last, contiguous) dimension of an NHWC tensor: one wave per pixel row ---------------
// (reference: nn.Softmax(dim=1) on the encoded reference label features, generator.py:384)
__global__ __launch_bounds__(256) void fsv_softmax_rows_fwd_kernel(const float* x, float* y, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool ok = row < rows;
  const float* xr = x + (ok ? row : 0) * C;
  float m = -3.0e38f;
  for (int j = lane; j < C; j += 64) m = fmaxf(m, xr[j]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += expf(xr[j] - m);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.f / s;
  if (ok) {
    float* yr = y + row * C;
    for (int j = lane; j < C; j += 64) yr[j] = expf(xr[j] - m) * inv;
  }
}

// dx = y * (dy - sum_j dy_j y_j)
__global__ __launch_bounds__(256) void fsv_softmax_rows_bwd_kernel(const float* dy, const float* y, float* dx, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool ok = row < rows;
  const float* yr = y + (ok ? row : 0) * C;
  const float* gr = dy + (ok ? row : 0) * C;
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += gr[j] * yr[j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (ok) {
    float* dr = dx + row * C;
    for (int j = lane; j < C; j += 64) dr[j] = yr[j] * (gr[j] - s);
  }
}

// ---- Adam ------------------------------------------------------------------------------------------------------
// state[0] = step count (as float), state[1] = 1 - beta1^t, state[2] = 1 - beta2^t, state[3] = lr (set by the host)
__global__ void fsv_adam_tick_kernel(float* state, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = state[0] + 1.f;
    state[0] = t;
    state[1] = 1.f - powf(beta1, t);
    state[2] = 1.f - powf(beta2, t);
  }
}

__global__ __launch_bounds__(256) void fsv_adam_kernel(float* param, const float* grad, float* m, float* v,
                                                       const float* state, long long n, float beta1, float beta2,
                                                       float eps, float gscale) {
  const float bc1 = state[1], bc2 = state[2], lr = state[3];
  const float step_size = lr / bc1;
  const float rbc2 = 1.f / sqrtf(bc2);
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) {
    float g = grad[i] * gscale;
    float mi = beta1 * m[i] + (1.f - beta1) * g;
    float vi = beta2 * v[i] + (1.f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) * rbc2 + eps;
    param[i] = param[i] - step_size * (mi / denom);
  }
}

extern "C" {

int fsv_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 1) return FSV_ERR_BAD_ARG;
  if (C % 4 == 0) FSV_LAUNCH((fsv_up2x_fwd_kernel<4>), dim3(fsv_grid_for((long long)N * H * W * C)), dim3(256), stream, x, y, N, H, W, C);
  else FSV_LAUNCH((fsv_up2x_fwd_kernel<1>), dim3(fsv_grid_for((long long)N * H * W * C * 4)), dim3(256), stream, x, y, N, H, W, C);
  return fsv_check_launch();
}

int fsv_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, hipStream_t stream) {
  if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 1) return FSV_ERR_BAD_ARG;
  if (C % 4 == 0) FSV_LAUNCH((fsv_up2x_bwd_kernel<4>), dim3(fsv_grid_for((long long)N * H * W * C / 4)), dim3(256), stream, dy, dx, N, H, W, C);
  else FSV_LAUNCH((fsv_up2x_bwd_kernel<1>), dim3(fsv_grid_for((long long)N * H * W * C)), dim3(256), stream, dy, dx, N, H, W, C);
  return fsv_check_launch();
}

typedef _Float16 fsv_eh16x4 __attribute__((ext_vector_type(4)));

// four elements per work-item, optional half side output (fsv_common.h)
__global__ __launch_bounds__(256) void fsv_act_bwd4_kernel(const float* dy, const float* y, float* dx, long long total4, int act,
                                                           float scale, _Float16* dxh) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const float4 dv = *reinterpret_cast<const float4*>(dy + i * 4), yv = *reinterpret_cast<const float4*>(y + i * 4);
    const float da[4] = {dv.x, dv.y, dv.z, dv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float d = da[j] * scale;
      const float v = ya[j];
      if (act == FSV_ACT_LRELU) d = v > 0.f ? d : 0.2f * d;
      else if (act == FSV_ACT_TANH) d = d * (1.f - v * v);
      else if (act == FSV_ACT_SIGMOID) d = d * v * (1.f - v);
      else if (act == FSV_ACT_RELU) d = v > 0.f ? d : 0.f;
      else if (act == FSV_ACT_LRELU01) d = v > 0.f ? d : 0.1f * d;
      r[j] = d;
    }
    *reinterpret_cast<float4*>(dx + i * 4) = make_float4(r[0], r[1], r[2], r[3]);
    if (dxh) {
      fsv_eh16x4 h;
      h.x = (_Float16)r[0]; h.y = (_Float16)r[1]; h.z = (_Float16)r[2]; h.w = (_Float16)r[3];
      *reinterpret_cast<fsv_eh16x4*>(dxh + i * 4) = h;
    }
  }
}

int fsv_act_fwd(const float* x, float* y, long long total, int act, hipStream_t stream) {
  if (!x || !y || total < 0) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_act_fwd_kernel, dim3(fsv_grid_for(total / 4 + 1)), dim3(256), stream, x, y, total, act);
  return fsv_check_launch();
}

// dx = dy * scale * act'(y)   (y is the activation OUTPUT)
// dx_half != null: dx is also stored as IEEE half there (total % 4 == 0 only: FSV_ERR_UNSUPPORTED otherwise, nothing launched)
int fsv_act_bwd(const float* dy, const float* y, float* dx, long long total, int act, float scale, void* dx_half,
                hipStream_t stream) {
  if (!dy || !y || !dx || total < 0) return FSV_ERR_BAD_ARG;
  const bool v4 = (total & 3) == 0 && total > 0;
  if (dx_half && !v4) return FSV_ERR_UNSUPPORTED;
  _Float16* dxh = reinterpret_cast<_Float16*>(dx_half);
  if (v4) FSV_LAUNCH(fsv_act_bwd4_kernel, dim3(fsv_grid_for(total / 4)), dim3(256), stream, dy, y, dx, total / 4, act, scale, dxh);
  else FSV_LAUNCH(fsv_act_bwd_kernel, dim3(fsv_grid_for(total / 4 + 1)), dim3(256), stream, dy, y, dx, total, act, scale);
  return fsv_check_launch();
}

int fsv_softmax_rows_fwd(const float* x, float* y, long long rows, int C, hipStream_t stream) {
  if (!x || !y || rows < 1 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_softmax_rows_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, x, y, rows, C);
  return fsv_check_launch();
}

int fsv_softmax_rows_bwd(const float* dy, const float* y, float* dx, long long rows, int C, hipStream_t stream) {
  if (!dy || !y || !dx || rows < 1 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), stream, dy, y, dx, rows, C);
  return fsv_check_launch();
}

int fsv_adam_step(float* param, const float* grad, float* m, float* v, float* state, long long n, float beta1,
                  float beta2, float eps, float gscale, hipStream_t stream) {
  if (!param || !grad || !m || !v || !state || n < 0) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_adam_tick_kernel, dim3(1), dim3(64), stream, state, beta1, beta2);
  FSV_LAUNCH(fsv_adam_kernel, dim3(fsv_grid_for(n / 4 + 1)), dim3(256), stream, param, grad, m, v, (const float*)state, n,
             beta1, beta2, eps, gscale);
  return fsv_check_launch();
}

// One optimiser step issued in several pieces (ranges of the flat buffers whose gradients become final at different times): the
// piece with tick != 0 advances the step count / bias corrections in `state`, the others - ordered behind it by the caller - only
// read them.  Same arithmetic per element as fsv_adam_step.
int fsv_adam_step_range(float* param, const float* grad, float* m, float* v, float* state, long long n, float beta1,
                        float beta2, float eps, float gscale, int tick, hipStream_t stream) {
  if (!param || !grad || !m || !v || !state || n < 0) return FSV_ERR_BAD_ARG;
  if (tick) FSV_LAUNCH(fsv_adam_tick_kernel, dim3(1), dim3(64), stream, state, beta1, beta2);
  if (n > 0)
    FSV_LAUNCH(fsv_adam_kernel, dim3(fsv_grid_for(n / 4 + 1)), dim3(256), stream, param, grad, m, v, (const float*)state, n,
               beta1, beta2, eps, gscale);
  return fsv_check_launch();
}

}  // extern "C"

// ---- channel concatenation into NHWC (U-Net skips generator.py:563, flow-net input :498, ds_ref :441-443) -----------------
// out[n][px][coff + c] = src[n, c, px]  for one source (strided: batch, channel, pixel); launched once per source.
__global__ __launch_bounds__(256) void fsv_cat_put_kernel(const float* src, float* out, long long N, int C, long long P,
                                                          long long sn, long long sc, long long sp, int Ct, int coff) {
  const long long total = N * P * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const long long px = t % P, n = t / P;
    out[(n * P + px) * Ct + coff + c] = src[n * sn + c * sc + px * sp];
  }
}

// float4 form for channel-contiguous sources (an NHWC tensor or a channel slice of one; every stride, offset and count a multiple of
// four): one work-item per quad, 32-bit index arithmetic.  (The element form above spends its time in 64-bit integer divisions: 1.6 TB/s
// on the 33 MB concatenations of the step.)
__global__ __launch_bounds__(256) void fsv_cat_put4_kernel(const float* src, float* out, unsigned total4, unsigned C4, unsigned P,
                                                           long long sn, long long sp, unsigned Ct, unsigned coff) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned c4 = i % C4, t = i / C4;
    const unsigned px = t % P, n = t / P;
    const float4 v = *reinterpret_cast<const float4*>(src + n * sn + px * sp + c4 * 4);
    *reinterpret_cast<float4*>(out + ((long long)n * P + px) * Ct + coff + c4 * 4) = v;
  }
}

// pixel-contiguous source planes (an NCHW tensor: sp == 1) -> dense [px][Ct]: one work-item per pixel reads its C values from the C
// planes (each plane read is coalesced across the wave) and writes its Ct values as float4 (consecutive work-items, consecutive
// addresses).  The element form reads C different planes within one wave instruction.
typedef _Float16 fsv_ew_h16x4 __attribute__((ext_vector_type(4)));
template <int CT, bool HALF = false>
__global__ __launch_bounds__(256) void fsv_pad_channels_px_kernel(const float* src, float* out, unsigned N, int C, unsigned P, long long sn,
                                                                  long long sc) {
  const unsigned total = N * P;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned n = i / P, px = i - n * P;
    const float* s0 = src + n * sn + px;
    float v[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) v[c] = c < C ? s0[c * sc] : 0.f;
    if constexpr (HALF) {            // the `--amp` path: the padded tensor is a convolution input - rounded to half right here
      fsv_ew_h16x4* o = reinterpret_cast<fsv_ew_h16x4*>(reinterpret_cast<_Float16*>(out) + (long long)i * CT);
#pragma unroll
      for (int q = 0; q < CT / 4; ++q) {
        fsv_ew_h16x4 t;
        t[0] = (_Float16)v[4 * q]; t[1] = (_Float16)v[4 * q + 1]; t[2] = (_Float16)v[4 * q + 2]; t[3] = (_Float16)v[4 * q + 3];
        o[q] = t;
      }
    } else {
      float4* o = reinterpret_cast<float4*>(out + (long long)i * CT);
#pragma unroll
      for (int q = 0; q < CT / 4; ++q) o[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  }
}

// any channel count, plane-strided source (NCHW labels: 35 -> 40, a packed 76 -> 80): one work-item per (pixel, group of 8 output
// channels) - consecutive work-items take consecutive pixels of the same group, so every plane read is coalesced across the wave -
// and one 16-byte (half) resp. two 16-byte (fp32) stores per work-item
template <bool HALF>
__global__ __launch_bounds__(256) void fsv_pad_channels_g8_kernel(const float* src, float* out, unsigned N, int C, unsigned P, long long sn,
                                                                  long long sc, int Ct) {
  const unsigned G = (unsigned)Ct >> 3;
  const unsigned long long total = (unsigned long long)N * P * G;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * 256u) {
    const unsigned px = (unsigned)(i % P);
    const unsigned long long t = i / P;
    const unsigned g = (unsigned)(t % G), n = (unsigned)(t / G);
    const float* s0 = src + n * sn + px;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int c = (int)g * 8 + e; v[e] = c < C ? s0[c * sc] : 0.f; }
    const long long o = ((long long)n * P + px) * Ct + g * 8;
    if constexpr (HALF) {
      fsv_ew_h16x4 a, b;
      a[0] = (_Float16)v[0]; a[1] = (_Float16)v[1]; a[2] = (_Float16)v[2]; a[3] = (_Float16)v[3];
      b[0] = (_Float16)v[4]; b[1] = (_Float16)v[5]; b[2] = (_Float16)v[6]; b[3] = (_Float16)v[7];
      fsv_ew_h16x4* d = reinterpret_cast<fsv_ew_h16x4*>(reinterpret_cast<_Float16*>(out) + o);
      d[0] = a; d[1] = b;
    } else {
      float4* d = reinterpret_cast<float4*>(out + o);
      d[0] = make_float4(v[0], v[1], v[2], v[3]); d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// channel padding for the float4 gather path of the convolutions (labels 6 -> 8, RGB 3 -> 4, flow-net input 15 -> 16):
// out[n][px][c] = c < C ? src[n, c, px] : 0 for any (batch, channel, pixel)-strided source, one pass instead of
// layout conversion + zero fill + copy
__global__ __launch_bounds__(256) void fsv_pad_channels_kernel(const float* src, float* out, long long N, int C, long long P,
                                                               long long sn, long long sc, long long sp, int Ct) {
  const long long total = N * P * Ct;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Ct);
    const long long t = i / Ct;
    const long long px = t % P, n = t / P;
    out[i] = c < C ? src[n * sn + c * sc + px * sp] : 0.f;
  }
}

// gradient of one source: dst[n][px][c] (dense NHWC with C channels) = dout[n][px][coff + c]
__global__ __launch_bounds__(256) void fsv_cat_get_kernel(const float* dout, float* dst, long long N, int C, long long P, int Ct, int coff) {
  const long long total = N * P * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long long pix = i / C;
    dst[i] = dout[pix * Ct + coff + c];
  }
}

__global__ __launch_bounds__(256) void fsv_cat_get4_kernel(const float* dout, float* dst, unsigned total4, unsigned C4, unsigned Ct, unsigned coff) {
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total4; i += gridDim.x * 256u) {
    const unsigned c4 = i % C4, pix = i / C4;
    reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(dout + (long long)pix * Ct + coff + c4 * 4);
  }
}

// ---- occlusion-mask compositing (generator.py:217,224): out = a * m + b * (1 - m), m broadcast over channels --------------
// a, b, out: [N][C][P] with (batch, channel, pixel) strides; m: [N][P] contiguous.
struct BlendP {
  const float* a; const float* b; const float* m; float* out;
  long long asn, asc, asp, bsn, bsc, bsp, osn, osc, osp;
  int N, C; long long P;
};
__global__ __launch_bounds__(256) void fsv_blend_fwd_kernel(BlendP p) {
  const long long total = (long long)p.N * p.P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.P, px = i - n * p.P;
    const float mv = p.m[i];
    for (int c = 0; c < p.C; ++c) {
      float av = p.a[n * p.asn + c * p.asc + px * p.asp], bv = p.b[n * p.bsn + c * p.bsc + px * p.bsp];
      p.out[n * p.osn + c * p.osc + px * p.osp] = av * mv + bv * (1.f - mv);
    }
  }
}
// da = g * m, db = g * (1 - m), dm = sum_c g * (a - b); g has the strides of `out`, da/db are dense NCHW
__global__ __launch_bounds__(256) void fsv_blend_bwd_kernel(BlendP p, const float* g, float* da, float* db, float* dm) {
  const long long total = (long long)p.N * p.P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.P, px = i - n * p.P;
    const float mv = p.m[i];
    float acc = 0.f;
    for (int c = 0; c < p.C; ++c) {
      float gv = g[n * p.osn + c * p.osc + px * p.osp];
      float av = p.a[n * p.asn + c * p.asc + px * p.asp], bv = p.b[n * p.bsn + c * p.bsc + px * p.bsp];
      const long long o = (n * p.C + c) * p.P + px;
      if (da) da[o] = gv * mv;
      if (db) db[o] = gv * (1.f - mv);
      acc += gv * (av - bv);
    }
    if (dm) dm[i] = acc;
  }
}

extern "C" {

int fsv_cat_put(const float* src, float* out, long long N, int C, long long P, const long long* strides, int Ct, int coff,
                hipStream_t stream) {
  if (!src || !out || N < 1 || C < 1 || P < 1 || coff < 0 || coff + C > Ct) return FSV_ERR_BAD_ARG;
  const long long total = N * P * C;
  if (strides[1] == 1 && ((C | Ct | coff) & 3) == 0 && ((strides[0] | strides[2]) & 3) == 0 && total < (1ll << 31) &&
      (((uintptr_t)src | (uintptr_t)out) & 15) == 0) {
    FSV_LAUNCH(fsv_cat_put4_kernel, dim3(fsv_grid_for(total / 4)), dim3(256), stream, src, out, (unsigned)(total / 4), (unsigned)(C / 4),
               (unsigned)P, strides[0], strides[2], (unsigned)Ct, (unsigned)coff);
    return fsv_check_launch();
  }
  FSV_LAUNCH(fsv_cat_put_kernel, dim3(fsv_grid_for(N * P * C / 4 + 1)), dim3(256), stream, src, out, N, C, P, strides[0], strides[1],
             strides[2], Ct, coff);
  return fsv_check_launch();
}

// fsv_pad_channels with the padded tensor written as IEEE half (the `--amp` path: it is a convolution input); pixel-contiguous
// planes (strides[2] == 1) and Ct % 8 == 0 only
int fsv_pad_channels_h(const float* src, void* out, long long N, int C, long long P, const long long* strides, int Ct,
                       hipStream_t stream) {
  if (!src || !out || !strides || N < 1 || C < 1 || P < 1 || Ct < C) return FSV_ERR_BAD_ARG;
  if (strides[2] != 1 || (Ct & 7) != 0 || N * P >= (1ll << 31) || ((uintptr_t)out & 15) != 0) return FSV_ERR_UNSUPPORTED;
  float* o = reinterpret_cast<float*>(out);
  if (Ct == 8) FSV_LAUNCH((fsv_pad_channels_px_kernel<8, true>), dim3(fsv_grid_for(N * P)), dim3(256), stream, src, o, (unsigned)N, C, (unsigned)P, strides[0], strides[1]);
  else if (Ct == 16) FSV_LAUNCH((fsv_pad_channels_px_kernel<16, true>), dim3(fsv_grid_for(N * P)), dim3(256), stream, src, o, (unsigned)N, C, (unsigned)P, strides[0], strides[1]);
  else FSV_LAUNCH((fsv_pad_channels_g8_kernel<true>), dim3(fsv_grid_for(N * P * (Ct >> 3))), dim3(256), stream, src, o, (unsigned)N, C, (unsigned)P, strides[0], strides[1], Ct);
  return fsv_check_launch();
}

int fsv_pad_channels(const float* src, float* out, long long N, int C, long long P, const long long* strides, int Ct,
                     hipStream_t stream) {
  if (!src || !out || !strides || N < 1 || C < 1 || P < 1 || Ct < C) return FSV_ERR_BAD_ARG;
  if (strides[2] == 1 && (Ct & 7) == 0 && Ct > 16 && N * P < (1ll << 31) && ((uintptr_t)out & 15) == 0) {
    FSV_LAUNCH((fsv_pad_channels_g8_kernel<false>), dim3(fsv_grid_for(N * P * (Ct >> 3))), dim3(256), stream, src, out, (unsigned)N, C, (unsigned)P, strides[0], strides[1], Ct);
    return fsv_check_launch();
  }
  if (strides[2] == 1 && (Ct == 4 || Ct == 8 || Ct == 12 || Ct == 16) && N * P < (1ll << 31) && ((uintptr_t)out & 15) == 0) {
    const dim3 g(fsv_grid_for(N * P));
    switch (Ct) {
      case 4: FSV_LAUNCH((fsv_pad_channels_px_kernel<4>), g, dim3(256), stream, src, out, (unsigned)N, C, (unsigned)P, strides[0], strides[1]); break;
      case 8: FSV_LAUNCH((fsv_pad_channels_px_kernel<8>), g, dim3(256), stream, src, out, (unsigned)N, C, (unsigned)P, strides[0], strides[1]); break;
      case 12: FSV_LAUNCH((fsv_pad_channels_px_kernel<12>), g, dim3(256), stream, src, out, (unsigned)N, C, (unsigned)P, strides[0], strides[1]); break;
      default: FSV_LAUNCH((fsv_pad_channels_px_kernel<16>), g, dim3(256), stream, src, out, (unsigned)N, C, (unsigned)P, strides[0], strides[1]); break;
    }
    return fsv_check_launch();
  }
  FSV_LAUNCH(fsv_pad_channels_kernel, dim3(fsv_grid_for(N * P * Ct / 4 + 1)), dim3(256), stream, src, out, N, C, P, strides[0],
             strides[1], strides[2], Ct);
  return fsv_check_launch();
}

int fsv_cat_get(const float* dout, float* dst, long long N, int C, long long P, int Ct, int coff, hipStream_t stream) {
  if (!dout || !dst || N < 1 || C < 1 || P < 1 || coff < 0 || coff + C > Ct) return FSV_ERR_BAD_ARG;
  if (((C | Ct | coff) & 3) == 0 && N * P * C < (1ll << 31) && (((uintptr_t)dout | (uintptr_t)dst) & 15) == 0) {
    const long long t4 = N * P * C / 4;
    FSV_LAUNCH(fsv_cat_get4_kernel, dim3(fsv_grid_for(t4)), dim3(256), stream, dout, dst, (unsigned)t4, (unsigned)(C / 4), (unsigned)Ct, (unsigned)coff);
    return fsv_check_launch();
  }
  FSV_LAUNCH(fsv_cat_get_kernel, dim3(fsv_grid_for(N * P * C / 4 + 1)), dim3(256), stream, dout, dst, N, C, P, Ct, coff);
  return fsv_check_launch();
}

int fsv_blend_fwd(const float* a, const float* b, const float* m, float* out, int N, int C, long long P,
                  const long long* a_strides, const long long* b_strides, const long long* out_strides, hipStream_t stream) {
  if (!a || !b || !m || !out || N < 1 || C < 1 || P < 1) return FSV_ERR_BAD_ARG;
  BlendP p;
  p.a = a; p.b = b; p.m = m; p.out = out; p.N = N; p.C = C; p.P = P;
  p.asn = a_strides[0]; p.asc = a_strides[1]; p.asp = a_strides[2];
  p.bsn = b_strides[0]; p.bsc = b_strides[1]; p.bsp = b_strides[2];
  p.osn = out_strides[0]; p.osc = out_strides[1]; p.osp = out_strides[2];
  FSV_LAUNCH(fsv_blend_fwd_kernel, dim3(fsv_grid_for((long long)N * P)), dim3(256), stream, p);
  return fsv_check_launch();
}

int fsv_blend_bwd(const float* a, const float* b, const float* m, const float* g, float* da, float* db, float* dm, int N, int C,
                  long long P, const long long* a_strides, const long long* b_strides, const long long* g_strides,
                  hipStream_t stream) {
  if (!a || !b || !m || !g || N < 1 || C < 1 || P < 1) return FSV_ERR_BAD_ARG;
  BlendP p;
  p.a = a; p.b = b; p.m = m; p.out = nullptr; p.N = N; p.C = C; p.P = P;
  p.asn = a_strides[0]; p.asc = a_strides[1]; p.asp = a_strides[2];
  p.bsn = b_strides[0]; p.bsc = b_strides[1]; p.bsp = b_strides[2];
  p.osn = g_strides[0]; p.osc = g_strides[1]; p.osp = g_strides[2];
  FSV_LAUNCH(fsv_blend_bwd_kernel, dim3(fsv_grid_for((long long)N * P)), dim3(256), stream, p, g, da, db, dm);
  return fsv_check_launch();
}

}  // extern "C"


// ---- 2x2 stride-2 max pooling (VGG19, models/networks/vgg.py:45-59 through torchvision's feature stack), NHWC ----------
__global__ __launch_bounds__(256) void fsv_maxpool2_fwd_kernel(const float* x, float* y, int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    const float* b = x + (((n * H + 2 * oy) * W + 2 * ox) * C + c);
    y[i] = fmaxf(fmaxf(b[0], b[C]), fmaxf(b[(long long)W * C], b[(long long)W * C + C]));
  }
}
// the gradient goes to the first maximal element in (0,0),(0,1),(1,0),(1,1) order, as ATen's max_pool2d does
__global__ __launch_bounds__(256) void fsv_maxpool2_bwd_kernel(const float* x, const float* dy, float* dx, int N, int H, int W, int C) {
  const int OH = H / 2, OW = W / 2;
  const long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const long long n = t / H;
    const int oy = yy >> 1, ox = xx >> 1;
    float g = 0.f;
    if (oy < OH && ox < OW) {
      const float* b = x + (((n * H + 2 * oy) * W + 2 * ox) * C + c);
      float v[4] = {b[0], b[C], b[(long long)W * C], b[(long long)W * C + C]};
      int arg = 0;
      float m = v[0];
      for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; arg = k; }
      if (arg == ((yy & 1) * 2 + (xx & 1))) g = dy[((n * OH + oy) * OW + ox) * C + c];
    }
    dx[i] = g;
  }
}

extern "C" {
int fsv_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream) {
  if (!x || !y || N < 1 || H < 2 || W < 2 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_maxpool2_fwd_kernel, dim3(fsv_grid_for((long long)N * (H / 2) * (W / 2) * C / 2 + 1)), dim3(256), stream, x, y, N, H, W, C);
  return fsv_check_launch();
}
int fsv_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, hipStream_t stream) {
  if (!x || !dy || !dx || N < 1 || H < 2 || W < 2 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_maxpool2_bwd_kernel, dim3(fsv_grid_for((long long)N * H * W * C / 2 + 1)), dim3(256), stream, x, dy, dx, N, H, W, C);
  return fsv_check_launch();
}
}  // extern "C"

// ---- 3x3 stride-2 average pooling, padding 1, count_include_pad = False (MultiscaleDiscriminator's pyramid between its
// discriminators, models/networks/discriminator.py:28,56), NHWC; OH = (H - 1) / 2 + 1 -------------------------------------
__device__ __forceinline__ int fsv_ap_count(int o, int extent) {      // valid taps of output index o along one axis
  const int lo = 2 * o - 1, hi = 2 * o + 1;
  return (hi < extent ? hi : extent - 1) - (lo > 0 ? lo : 0) + 1;
}
__global__ __launch_bounds__(256) void fsv_avgpool3s2_fwd_kernel(const float* x, float* y, int N, int H, int W, int C) {
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const long long n = t / OH;
    float s = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = 2 * oy + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = 2 * ox + dx;
        if ((unsigned)xx < (unsigned)W) s += x[((n * H + yy) * W + xx) * C + c];
      }
    }
    y[i] = s / (float)(fsv_ap_count(oy, H) * fsv_ap_count(ox, W));
  }
}
__global__ __launch_bounds__(256) void fsv_avgpool3s2_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C) {
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * H * W * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long t = i / C;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const long long n = t / H;
    // an even row belongs to the window of output yy / 2 only, an odd row to those of (yy - 1) / 2 and (yy + 1) / 2
    const int oy0 = yy >> 1, oy1 = (yy & 1) ? oy0 + 1 : oy0;
    const int ox0 = xx >> 1, ox1 = (xx & 1) ? ox0 + 1 : ox0;
    float g = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ox0; ox <= ox1; ++ox)
        if (ox < OW) g += dy[((n * OH + oy) * OW + ox) * C + c] / (float)(fsv_ap_count(oy, H) * fsv_ap_count(ox, W));
    }
    dx[i] = g;
  }
}

extern "C" {
// nn.AdaptiveAvgPool2d((OH, OW)) on NHWC (discriminator.py:146,153: the pooled reference encoding the AdaptiveDiscriminator's weight
// generator reads): output (oy, ox) averages rows [floor(oy H / OH), ceil((oy + 1) H / OH)) x the same in x - ATen's window rule
__device__ __forceinline__ void fsv_adapt_win(int o, int in, int out, int& s, int& e) {
  s = (int)(((long long)o * in) / out);
  e = (int)((((long long)(o + 1)) * in + out - 1) / out);
}

__global__ __launch_bounds__(256) void fsv_adaptive_avgpool_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int OH,
                                                                       int OW) {
  const long long total = (long long)N * OH * OW * C;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    int ys, ye, xs, xe;
    fsv_adapt_win(oy, H, OH, ys, ye);
    fsv_adapt_win(ox, W, OW, xs, xe);
    float acc = 0.f;
    for (int yy = ys; yy < ye; ++yy)
      for (int xx = xs; xx < xe; ++xx) acc += x[(((long long)n * H + yy) * W + xx) * C + c];
    y[i] = acc / (float)((ye - ys) * (xe - xs));
  }
}

// every input element collects dy / window size of the windows that contain it (no atomics).  ATen's rule - window oy covers
// [floor(oy H / OH), ceil((oy + 1) H / OH)) - puts input row yy into the outputs floor(yy OH / H) ... ceil((yy + 1) OH / H) - 1: at
// most two per axis when the map shrinks, ~OH / H + 1 when it GROWS (OH > H: discriminator.py:146,153 pool to fineSize / 8 whatever
// the encoded map's size - with adaptive_D_layers = 3 and num_D = 2 the second scale's 33-pixel map is pooled UP to 64)
__global__ __launch_bounds__(256) void fsv_adaptive_avgpool_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int OH,
                                                                       int OW) {
  const long long total = (long long)N * H * W * C;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    const int oy_lo = (int)(((long long)yy * OH) / H), oy_hi = (int)((((long long)yy + 1) * OH + H - 1) / H) - 1;
    const int ox_lo = (int)(((long long)xx * OW) / W), ox_hi = (int)((((long long)xx + 1) * OW + W - 1) / W) - 1;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi && oy < OH; ++oy) {
      int ys, ye;
      fsv_adapt_win(oy, H, OH, ys, ye);
      if (yy < ys || yy >= ye) continue;
      for (int ox = ox_lo; ox <= ox_hi && ox < OW; ++ox) {
        int xs, xe;
        fsv_adapt_win(ox, W, OW, xs, xe);
        if (xx < xs || xx >= xe) continue;
        acc += dy[(((long long)n * OH + oy) * OW + ox) * C + c] / (float)((ye - ys) * (xe - xs));
      }
    }
    dx[i] = acc;
  }
}

int fsv_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, hipStream_t stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 1 || OH < 1 || OW < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_adaptive_avgpool_fwd_kernel, dim3(fsv_grid_for((long long)N * OH * OW * C)), dim3(256), stream, x, y, N, H, W, C, OH, OW);
  return fsv_check_launch();
}

int fsv_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, hipStream_t stream) {
  if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 1 || OH < 1 || OW < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_adaptive_avgpool_bwd_kernel, dim3(fsv_grid_for((long long)N * H * W * C)), dim3(256), stream, dy, dx, N, H, W, C, OH, OW);
  return fsv_check_launch();
}

int fsv_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_avgpool3s2_fwd_kernel, dim3(fsv_grid_for((long long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C)), dim3(256), stream,
             x, y, N, H, W, C);
  return fsv_check_launch();
}
int fsv_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, hipStream_t stream) {
  if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_avgpool3s2_bwd_kernel, dim3(fsv_grid_for((long long)N * H * W * C)), dim3(256), stream, dy, dx, N, H, W, C);
  return fsv_check_launch();
}
}  // extern "C"
