// The three native operators of the FlowNet2 teacher (the reference's only C++/CUDA code), re-designed for CDNA4.
//
// Reference: models/networks/flownet2_pytorch/networks/{correlation,resample2d,channelnorm}_package.
// They are reached only through models/flownet.py under torch.no_grad(), so only the forward operators exist here.
//
//  * fsv_correlation_fwd  FlowNetC cost volume (correlation_cuda_kernel.cu:74-147): for every output pixel and each of the
//    (2*(max_disp/stride2)+1)^2 displacements the mean over channels of f1 . f2 (kernel_size 1, zero padding).
//    The reference uses one 32-lane warp per output pixel with a shuffle reduction per displacement and re-reads the
//    f2 vectors 441 times through the cache hierarchy.  Here a workgroup owns 32 consecutive output pixels of one row and
//    ONE displacement row: the f1 segment and the (32 + halo)-pixel f2 segment are staged through LDS in 64-channel
//    chunks and every work-item accumulates its own (pixel, displacement) outputs in registers - no cross-lane reduction,
//    f2 is read from HBM/L2 ~2.25 x D times instead of D^2 times.  Inputs NHWC (the layout of the producing convolution),
//    output NHWC.  Floating-point summation order differs from the reference (tolerance 1e-5, tests/op_checks.py).
//  * fsv_resample2d_fwd   backward warp with pixel-unit flow and clamped tap INDICES (resample2d_kernel.cu:16-64); the
//    four tap weights are evaluated in double and each product rounded to float before the float accumulation, exactly
//    as the reference's mixed `1. - alpha` arithmetic does: results are bit-identical.
//  * fsv_channelnorm_fwd  per-pixel L2 norm over channels (channelnorm_kernel.cu:18-60), fp32 accumulation in channel
//    order: bit-identical.
#include "fsv_common.h"

#define FSV_CORR_PX 32
#define FSV_CORR_CH 64
#define FSV_CORR_P2MAX 96      // staged f2 pixels: 31 * stride1 + 2 * (max_disp / stride2) * stride2 + 1 (72 in FlowNet2)

struct CorrP {
  const float* f1;      // [N][H][W][C]
  const float* f2;
  float* out;           // [N][OH][OW][D*D]
  int N, H, W, C, OH, OW;
  int pad, max_disp, stride1, stride2, drad, D;
  int P2;               // f2 pixels staged per tile
};

__global__ __launch_bounds__(256) void fsv_correlation_kernel(CorrP p) {
  __shared__ float A[FSV_CORR_PX * (FSV_CORR_CH + 1)];      // [px][ch]
  __shared__ float Bt[FSV_CORR_P2MAX * (FSV_CORR_CH + 1)];  // [staged f2 px][ch]
  const int ox0 = blockIdx.x * FSV_CORR_PX, oy = blockIdx.y;
  const int n = blockIdx.z / p.D, tj = (int)(blockIdx.z % p.D) - p.drad;
  // padded coordinates of the reference -> un-padded: y = oy * stride1 + max_disp - pad
  const int y1 = oy * p.stride1 + p.max_disp - p.pad;
  const int y2 = y1 + tj * p.stride2;
  const int x1_0 = ox0 * p.stride1 + p.max_disp - p.pad;   // x of the tile's first f1 pixel
  const int x2_0 = x1_0 - p.drad * p.stride2;              // x of the first staged f2 pixel
  const int px = threadIdx.x & (FSV_CORR_PX - 1), g = threadIdx.x / FSV_CORR_PX;      // 8 displacement groups
  const int nti = (p.D + 7) / 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const bool row1 = (unsigned)y1 < (unsigned)p.H, row2 = (unsigned)y2 < (unsigned)p.H;
  for (int c0 = 0; c0 < p.C; c0 += FSV_CORR_CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < FSV_CORR_PX * FSV_CORR_CH; i += 256) {
      const int q = i / FSV_CORR_CH, c = i - q * FSV_CORR_CH;
      const int x = x1_0 + q * p.stride1;
      float v = 0.f;
      if (row1 && (unsigned)x < (unsigned)p.W && c0 + c < p.C && ox0 + q < p.OW)
        v = p.f1[(((long long)n * p.H + y1) * p.W + x) * p.C + c0 + c];
      A[q * (FSV_CORR_CH + 1) + c] = v;
    }
    for (int i = threadIdx.x; i < p.P2 * FSV_CORR_CH; i += 256) {
      const int q = i / FSV_CORR_CH, c = i - q * FSV_CORR_CH;
      const int x = x2_0 + q;
      float v = 0.f;
      if (row2 && (unsigned)x < (unsigned)p.W && c0 + c < p.C)
        v = p.f2[(((long long)n * p.H + y2) * p.W + x) * p.C + c0 + c];
      Bt[q * (FSV_CORR_CH + 1) + c] = v;
    }
    __syncthreads();
    for (int c = 0; c < FSV_CORR_CH; ++c) {
      const float a = A[px * (FSV_CORR_CH + 1) + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ti = g + 8 * k;
        if (k < nti && ti < p.D) acc[k] += a * Bt[(px * p.stride1 + ti * p.stride2) * (FSV_CORR_CH + 1) + c];
      }
    }
  }
  const int ox = ox0 + px;
  if (ox < p.OW) {
    const float nelems = (float)p.C;                       // kernel_size^2 * C with kernel_size 1
    float* o = p.out + (((long long)n * p.OH + oy) * p.OW + ox) * (p.D * p.D) + (tj + p.drad) * p.D;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ti = g + 8 * k;
      if (k < nti && ti < p.D) o[ti] = acc[k] / nelems;
    }
  }
}

// out[n][c][y][x]: all tensors with explicit (n, c, y, x) strides
struct Str4 { long long v[4]; };
__global__ __launch_bounds__(256) void fsv_resample2d_kernel(const float* img, const float* flow, float* out, int N, int C,
                                                             int H, int W, Str4 is_, Str4 fs_, Str4 os_) {
  const long long* is = is_.v;
  const long long* fs = fs_.v;
  const long long* os = os_.v;
  const long long total = (long long)N * C * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long long)W * H)) % C);
    const int n = (int)(i / ((long long)W * H * C));
    const float dx = flow[n * fs[0] + y * fs[2] + x * fs[3]];
    const float dy = flow[n * fs[0] + fs[1] + y * fs[2] + x * fs[3]];
    const float xf = (float)x + dx, yf = (float)y + dy;
    const float alpha = xf - floorf(xf), beta = yf - floorf(yf);
    int xL = (int)floorf(xf), xR = (int)(floorf(xf) + 1.f), yT = (int)floorf(yf), yB = (int)(floorf(yf) + 1.f);
    xL = xL < W - 1 ? xL : W - 1; xL = xL > 0 ? xL : 0;
    xR = xR < W - 1 ? xR : W - 1; xR = xR > 0 ? xR : 0;
    yT = yT < H - 1 ? yT : H - 1; yT = yT > 0 ? yT : 0;
    yB = yB < H - 1 ? yB : H - 1; yB = yB > 0 ? yB : 0;
    const float* b = img + n * is[0] + c * is[1];
    const double a1 = 1.0 - (double)alpha, b1 = 1.0 - (double)beta;
    float val = 0.f;
    val += (float)(a1 * b1 * (double)b[yT * is[2] + xL * is[3]]);
    val += (float)((double)alpha * b1 * (double)b[yT * is[2] + xR * is[3]]);
    val += (float)(a1 * (double)beta * (double)b[yB * is[2] + xL * is[3]]);
    // the fourth tap of the reference is `(alpha)*(beta) * x` - no double literal in it: float products (resample2d_kernel.cu:60;
    // found by holding this kernel to the reference's own kernel compiled for the host, oracle/build_ref.py)
    val += (alpha * beta) * b[yB * is[2] + xR * is[3]];
    out[n * os[0] + c * os[1] + y * os[2] + x * os[3]] = val;
  }
}

// Bilinear resize, align_corners = False - F.interpolate(mode='bilinear') as FlowNet2 uses it (models.py:119,137-142 x4 flow
// up-sampling; models/flownet.py:66-77 resize to a multiple of 64 and back).  ATen's op sequence (UpSample.h
// area_pixel_compute_source_index: src = scale * (dst + 0.5) - 0.5 clamped at 0, scale = in / out; UpSampleBilinear2d.cu: the
// four taps weighted (1 - l) / l in float, h-major), one work-item per output element, explicit (n, c, y, x) strides.
__global__ __launch_bounds__(256) void fsv_bilinear_resize_kernel(const float* in, float* out, int N, int C, int IH, int IW,
                                                                  int OH, int OW, float rh, float rw, Str4 is_, Str4 os_) {
  const long long* is = is_.v;
  const long long* os = os_.v;
  const long long total = (long long)N * C * OH * OW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % OW), y = (int)((i / OW) % OH), c = (int)((i / ((long long)OW * OH)) % C);
    const int n = (int)(i / ((long long)OW * OH * C));
    float sy = rh * ((float)y + 0.5f) - 0.5f, sx = rw * ((float)x + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = in + n * is[0] + c * is[1];
    const float v = hy * (hx * b[y0 * is[2] + x0 * is[3]] + lx * b[y0 * is[2] + x1 * is[3]]) +
                    ly * (hx * b[y1 * is[2] + x0 * is[3]] + lx * b[y1 * is[2] + x1 * is[3]]);
    out[n * os[0] + c * os[1] + y * os[2] + x * os[3]] = v;
  }
}

// Correctly rounded float square root from basic IEEE operations: the device sqrt (float and, as measured on gfx950, the
// float rounding of the double one) can be 1 ulp off the CPU / CUDA result.  y is within 1 ulp; the midpoints to its
// neighbours have 25 significant bits, so their squares are exact in double and decide the rounding.
__device__ __forceinline__ float fsv_sqrt_rn(float r) {
  float y = sqrtf(r);
  if (!(r > 0.f) || !(y > 0.f)) return y;
  const float lo = nextafterf(y, 0.f), hi = nextafterf(y, 3.0e38f);
  const double rd = (double)r;
  const double m1 = 0.5 * ((double)y + (double)lo), m2 = 0.5 * ((double)y + (double)hi);
  if (rd < m1 * m1) y = lo;
  else if (rd > m2 * m2) y = hi;
  return y;
}

__global__ __launch_bounds__(256) void fsv_channelnorm_kernel(const float* x, float* out, int N, int C, long long HW,
                                                              long long sn, long long sc, long long sp) {
  const long long total = (long long)N * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / HW, p = i - n * HW;
    float r = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = x[n * sn + c * sc + p * sp];
      r += v * v;
    }
    out[i] = fsv_sqrt_rn(r);
  }
}

extern "C" {

// f1 / f2: NHWC [N][H][W][C]; out: NHWC [N][OH][OW][D*D] with OH = ceil((H + 2*pad - 2*max_disp) / stride1) (kernel_size 1)
int fsv_correlation_fwd(const float* f1, const float* f2, float* out, int N, int H, int W, int C, int pad, int kernel_size,
                        int max_disp, int stride1, int stride2, hipStream_t stream) {
  if (!f1 || !f2 || !out || N < 1 || H < 1 || W < 1 || C < 1 || stride1 < 1 || stride2 < 1 || max_disp < 0)
    return FSV_ERR_BAD_ARG;
  if (kernel_size != 1) return FSV_ERR_UNSUPPORTED;
  CorrP p;
  p.f1 = f1; p.f2 = f2; p.out = out; p.N = N; p.H = H; p.W = W; p.C = C;
  p.pad = pad; p.max_disp = max_disp; p.stride1 = stride1; p.stride2 = stride2;
  p.drad = max_disp / stride2; p.D = 2 * p.drad + 1;
  const int ph = H + 2 * pad - 2 * max_disp, pw = W + 2 * pad - 2 * max_disp;
  if (ph < 1 || pw < 1 || p.D > 64) return FSV_ERR_BAD_ARG;
  p.OH = (ph + stride1 - 1) / stride1; p.OW = (pw + stride1 - 1) / stride1;
  p.P2 = (FSV_CORR_PX - 1) * stride1 + 2 * p.drad * stride2 + 1;
  if (p.P2 > FSV_CORR_P2MAX) return FSV_ERR_UNSUPPORTED;
  dim3 grid(fsv_cdiv(p.OW, FSV_CORR_PX), p.OH, N * p.D);
  FSV_LAUNCH(fsv_correlation_kernel, grid, dim3(256), stream, p);
  return fsv_check_launch();
}

int fsv_resample2d_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W, const long long* img_strides,
                       const long long* flow_strides, const long long* out_strides, hipStream_t stream) {
  if (!img || !flow || !out || !img_strides || !flow_strides || !out_strides || N < 1 || C < 1) return FSV_ERR_BAD_ARG;
  Str4 is, fs, os;
  for (int k = 0; k < 4; ++k) { is.v[k] = img_strides[k]; fs.v[k] = flow_strides[k]; os.v[k] = out_strides[k]; }
  long long g = ((long long)N * C * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  FSV_LAUNCH(fsv_resample2d_kernel, dim3((unsigned)g), dim3(256), stream, img, flow, out, N, C, H, W, is, fs, os);
  return fsv_check_launch();
}

int fsv_bilinear_resize_fwd(const float* in, float* out, int N, int C, int IH, int IW, int OH, int OW,
                            const long long* in_strides, const long long* out_strides, hipStream_t stream) {
  if (!in || !out || N < 1 || C < 1 || IH < 1 || IW < 1 || OH < 1 || OW < 1 || !in_strides || !out_strides) return FSV_ERR_BAD_ARG;
  Str4 is, os;
  for (int k = 0; k < 4; ++k) { is.v[k] = in_strides[k]; os.v[k] = out_strides[k]; }
  long long g = ((long long)N * C * OH * OW + 255) / 256;
  if (g > 8192) g = 8192;
  // ATen computes the scale in float as (float)in / out
  FSV_LAUNCH(fsv_bilinear_resize_kernel, dim3((unsigned)g), dim3(256), stream, in, out, N, C, IH, IW, OH, OW,
             (float)IH / (float)OH, (float)IW / (float)OW, is, os);
  return fsv_check_launch();
}

int fsv_channelnorm_fwd(const float* x, float* out, int N, int C, long long HW, long long sn, long long sc, long long sp,
                        hipStream_t stream) {
  if (!x || !out || N < 1 || C < 1 || HW < 1) return FSV_ERR_BAD_ARG;
  long long g = ((long long)N * HW + 255) / 256;
  if (g > 8192) g = 8192;
  FSV_LAUNCH(fsv_channelnorm_kernel, dim3((unsigned)g), dim3(256), stream, x, out, N, C, HW, sn, sc, sp);
  return fsv_check_launch();
}

}  // extern "C"
