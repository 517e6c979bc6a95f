// Face-region discriminator inputs (--add_face_D): bounding boxes and crop + nearest resize, without host round trips.
//
// Reference: models/face_refiner.py
//   get_face_region  (:56-87)  box = f(min / max row and column of the face pixels of the label map)   [.nonzero() + .item()]
//   crop_face_region (:32-39)  per sample  F.interpolate(image[i, -3:, ys:ye, xs:xe], size=(S, S))      [Python loop + cat]
// Here the boxes are computed on the device (one workgroup per sample) and stay there; the crop kernel reads them from
// device memory, so the whole face branch is a fixed launch sequence (hipGraph-capturable) and one launch per batch.
// Integer / index work: bit-exact with the reference (tests/op_checks.py check_face_ops).
#include "fsv_common.h"

// boxes[n] = {ys, ye, xs, xe}.  pose: [N][C][H][W] with strides (sn, sc, W, 1).
__global__ __launch_bounds__(256) void fsv_face_box_kernel(const float* pose, long long sn, long long sc, int C, int H, int W,
                                                           int use_openpose, int crop_smaller, int* boxes) {
  __shared__ int red[4][256];
  const int n = blockIdx.x;
  const float* p = pose + n * sn;
  int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    bool hit;
    if (use_openpose) hit = p[(C - 3) * sc + i] > 0.f && p[(C - 2) * sc + i] > 0.f && p[(C - 1) * sc + i] > 0.f;
    else hit = p[2 * sc + i] > 0.9f;
    if (hit) {
      const int y = i / W, x = i - y * W;
      ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
      xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
    }
  }
  red[0][threadIdx.x] = ymin; red[1][threadIdx.x] = ymax; red[2][threadIdx.x] = xmin; red[3][threadIdx.x] = xmax;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const int t = threadIdx.x;
      red[0][t] = red[0][t + o] < red[0][t] ? red[0][t + o] : red[0][t];
      red[1][t] = red[1][t + o] > red[1][t] ? red[1][t + o] : red[1][t];
      red[2][t] = red[2][t + o] < red[2][t] ? red[2][t + o] : red[2][t];
      red[3][t] = red[3][t + o] > red[3][t] ? red[3][t + o] : red[3][t];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int ys0 = red[0][0], ye0 = red[1][0], xs0 = red[2][0], xe0 = red[3][0];
    int yc, xc, len;
    if (ye0 >= 0) {
      if (use_openpose) {
        xc = (xs0 + xe0) / 2; yc = (ys0 * 3 + ye0 * 2) / 5;
        len = (int)((double)(xe0 - xs0) * 2.5);
      } else {
        xc = (xs0 + xe0) / 2; yc = (ys0 + ye0) / 2;
        len = (int)((double)(ye0 - ys0) * 1.25);
      }
      len = len > 32 ? len : 32;
      len = len < W ? len : W;
      int lo = len / 2, hi = H - 1 - len / 2;
      yc = yc < hi ? yc : hi; yc = yc > lo ? yc : lo;            // max(len//2, min(h-1-len//2, yc))
      hi = W - 1 - len / 2;
      xc = xc < hi ? xc : hi; xc = xc > lo ? xc : lo;
    } else {
      yc = H / 4; xc = W / 2; len = H / 32 * 8;
    }
    int ys = yc - len / 2, ye = yc + len / 2, xs = xc - len / 2, xe = xc + len / 2;
    ys += crop_smaller; xs += crop_smaller; ye -= crop_smaller; xe -= crop_smaller;
    boxes[n * 4 + 0] = ys; boxes[n * 4 + 1] = ye; boxes[n * 4 + 2] = xs; boxes[n * 4 + 3] = xe;
  }
}

// F.interpolate(mode='nearest', size=S): source index = min(floor(dst * (float)in / S), in - 1)
__device__ __forceinline__ int fsv_nearest_src(int dst, int in_size, int out_size) {
  const float scale = (float)in_size / (float)out_size;
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

// out[n][c][oy][ox] (contiguous NCHW, 3 channels) from the LAST three channels of img (strides sn, sc, sy, sx)
__global__ __launch_bounds__(256) void fsv_crop_resize_fwd_kernel(const float* img, long long sn, long long sc, long long sy,
                                                                  long long sx, int C, const int* boxes, float* out, int N, int S) {
  const long long total = (long long)N * 3 * S * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S), c = (int)((i / ((long long)S * S)) % 3);
    const int n = (int)(i / ((long long)3 * S * S));
    const int ys = boxes[n * 4], ye = boxes[n * 4 + 1], xs = boxes[n * 4 + 2], xe = boxes[n * 4 + 3];
    const int iy = ys + fsv_nearest_src(oy, ye - ys, S), ix = xs + fsv_nearest_src(ox, xe - xs, S);
    out[i] = img[n * sn + (C - 3 + c) * sc + iy * sy + ix * sx];
  }
}

// dimg (zero-initialised, same strides as img) receives the scattered dout
__global__ __launch_bounds__(256) void fsv_crop_resize_bwd_kernel(const float* dout, const int* boxes, float* dimg, long long sn,
                                                                  long long sc, long long sy, long long sx, int C, int N, int S) {
  const long long total = (long long)N * 3 * S * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S), c = (int)((i / ((long long)S * S)) % 3);
    const int n = (int)(i / ((long long)3 * S * S));
    const int ys = boxes[n * 4], ye = boxes[n * 4 + 1], xs = boxes[n * 4 + 2], xe = boxes[n * 4 + 3];
    const int iy = ys + fsv_nearest_src(oy, ye - ys, S), ix = xs + fsv_nearest_src(ox, xe - xs, S);
    atomicAdd(&dimg[n * sn + (C - 3 + c) * sc + iy * sy + ix * sx], dout[i]);
  }
}

// ---- face refinement paste (face_refiner.py:42-54 replace_face_region) -------------------------------------------------------
// out = img everywhere except inside box n, where out = clamp(bilinear_resize(face[n] -> box size), -1, 1)
// (F.interpolate(mode='bilinear', align_corners=False): src = (dst + 0.5) * in / out - 0.5, clamped at 0).
struct PasteTap { int y0, y1, x0, x1; float ly0, ly1, lx0, lx1; };
__device__ __forceinline__ void fsv_lin_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l0, float& l1) {
  const float scale = (float)in_size / (float)out_size;
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  l0 = 1.f - l1;
}

// img / out: [N][3][H][W] with strides (sn, sc, sy, sx) each; face: contiguous [N][3][S][S]
__global__ __launch_bounds__(256) void fsv_paste_face_fwd_kernel(const float* img, const float* face, const int* boxes, float* out,
                                                                 int N, int H, int W, int S, long long isn, long long isc,
                                                                 long long isy, long long isx) {
  const long long total = (long long)N * 3 * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long long)W * H)) % 3);
    const int n = (int)(i / ((long long)3 * H * W));
    const int ys = boxes[n * 4], ye = boxes[n * 4 + 1], xs = boxes[n * 4 + 2], xe = boxes[n * 4 + 3];
    float v = img[n * isn + c * isc + y * isy + x * isx];
    if (y >= ys && y < ye && x >= xs && x < xe) {
      int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
      fsv_lin_src(y - ys, S, ye - ys, y0, y1, ly0, ly1);
      fsv_lin_src(x - xs, S, xe - xs, x0, x1, lx0, lx1);
      const float* f = face + ((long long)n * 3 + c) * S * S;
      v = ly0 * (lx0 * f[y0 * S + x0] + lx1 * f[y0 * S + x1]) + ly1 * (lx0 * f[y1 * S + x0] + lx1 * f[y1 * S + x1]);
      v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
    }
    out[i] = v;
  }
}

// dimg (contiguous NCHW) = dout outside the boxes, 0 inside; dface (zero-initialised) += bilinear weights * dout where the
// clamp was inactive (out strictly inside (-1, 1))
__global__ __launch_bounds__(256) void fsv_paste_face_bwd_kernel(const float* dout, const float* out, const int* boxes, float* dimg,
                                                                 float* dface, int N, int H, int W, int S) {
  const long long total = (long long)N * 3 * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((long long)W * H)) % 3);
    const int n = (int)(i / ((long long)3 * H * W));
    const int ys = boxes[n * 4], ye = boxes[n * 4 + 1], xs = boxes[n * 4 + 2], xe = boxes[n * 4 + 3];
    const float d = dout[i];
    if (y >= ys && y < ye && x >= xs && x < xe) {
      dimg[i] = 0.f;
      const float o = out[i];
      if (o > -1.f && o < 1.f) {
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        fsv_lin_src(y - ys, S, ye - ys, y0, y1, ly0, ly1);
        fsv_lin_src(x - xs, S, xe - xs, x0, x1, lx0, lx1);
        float* f = dface + ((long long)n * 3 + c) * S * S;
        atomicAdd(&f[y0 * S + x0], d * ly0 * lx0); atomicAdd(&f[y0 * S + x1], d * ly0 * lx1);
        atomicAdd(&f[y1 * S + x0], d * ly1 * lx0); atomicAdd(&f[y1 * S + x1], d * ly1 * lx1);
      }
    } else {
      dimg[i] = d;
    }
  }
}

extern "C" {

int fsv_paste_face_fwd(const float* img, const float* face, const int* boxes, float* out, int N, int H, int W, int S,
                       long long isn, long long isc, long long isy, long long isx, hipStream_t stream) {
  if (!img || !face || !boxes || !out || N < 1 || S < 1) return FSV_ERR_BAD_ARG;
  long long g = ((long long)N * 3 * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  FSV_LAUNCH(fsv_paste_face_fwd_kernel, dim3((unsigned)g), dim3(256), stream, img, face, boxes, out, N, H, W, S, isn, isc, isy,
             isx);
  return fsv_check_launch();
}

int fsv_paste_face_bwd(const float* dout, const float* out, const int* boxes, float* dimg, float* dface, int N, int H, int W,
                       int S, hipStream_t stream) {
  if (!dout || !out || !boxes || !dimg || !dface || N < 1 || S < 1) return FSV_ERR_BAD_ARG;
  long long g = ((long long)N * 3 * H * W + 255) / 256;
  if (g > 8192) g = 8192;
  FSV_LAUNCH(fsv_paste_face_bwd_kernel, dim3((unsigned)g), dim3(256), stream, dout, out, boxes, dimg, dface, N, H, W, S);
  return fsv_check_launch();
}

int fsv_face_boxes(const float* pose, long long sn, long long sc, int N, int C, int H, int W, int use_openpose,
                   int crop_smaller, int* boxes, hipStream_t stream) {
  if (!pose || !boxes || N < 1 || C < 3 || H < 1 || W < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_face_box_kernel, dim3(N), dim3(256), stream, pose, sn, sc, C, H, W, use_openpose, crop_smaller, boxes);
  return fsv_check_launch();
}

int fsv_crop_resize_fwd(const float* img, long long sn, long long sc, long long sy, long long sx, int C, const int* boxes,
                        float* out, int N, int S, hipStream_t stream) {
  if (!img || !boxes || !out || N < 1 || C < 3 || S < 1) return FSV_ERR_BAD_ARG;
  long long g = ((long long)N * 3 * S * S + 255) / 256;
  if (g > 4096) g = 4096;
  FSV_LAUNCH(fsv_crop_resize_fwd_kernel, dim3((unsigned)g), dim3(256), stream, img, sn, sc, sy, sx, C, boxes, out, N, S);
  return fsv_check_launch();
}

int fsv_crop_resize_bwd(const float* dout, const int* boxes, float* dimg, long long sn, long long sc, long long sy,
                        long long sx, int C, int N, int S, hipStream_t stream) {
  if (!dout || !boxes || !dimg || N < 1 || C < 3 || S < 1) return FSV_ERR_BAD_ARG;
  long long g = ((long long)N * 3 * S * S + 255) / 256;
  if (g > 4096) g = 4096;
  FSV_LAUNCH(fsv_crop_resize_bwd_kernel, dim3((unsigned)g), dim3(256), stream, dout, boxes, dimg, sn, sc, sy, sx, C, N, S);
  return fsv_check_launch();
}

}  // extern "C"
