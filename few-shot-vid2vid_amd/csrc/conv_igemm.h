// Parameter block and tap decoding shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_db.hip).
#pragma once
#include "fsv_common.h"

#define FSV_BK 32

struct ConvP {
  const float* in;
  const float* wt;
  const float* bias;
  const float* res;
  const float* wscale;     // optional device scalar multiplying the accumulator (spectral-norm 1/sigma), or null
  float* out;
  int N, H, W, Cin;        // input tensor NHWC
  int OH, OW, Cout;        // iteration grid and number of output channels
  int K, nchunks, ldw;     // K = ntaps*Cin; nchunks = ceil(K/32); ldw = weight row stride
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;   // (ty+8) | (tx+8)<<4 per tap, 8 taps per word
  int outH, outW, osy, osx, ooy, oox, dense_out;
  long long w_bstride, b_bstride;        // per-sample weight / bias strides (0: shared)
  int per_sample, nsplit;                // blockIdx.z = sample*nsplit + ksplit
  int act; float scale;
  int Mz;                                // rows (pixels) per z group
};

__device__ __forceinline__ void fsv_tap(const ConvP& p, int t, int& ty, int& tx) {
  unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
  int sh = (t & 7) * 8;
  ty = (int)((code >> sh) & 15ull) - 8;
  tx = (int)((code >> (sh + 4)) & 15ull) - 8;
}

struct WgradP {
  const float* in;
  const float* dout;
  float* dwt;              // [K_pad][ldw] per z-sample
  int N, H, W, Cin;
  int OH, OW, Cout;
  int K, ldw;
  int sy, sx, ntaps;
  unsigned long long taps_lo, taps_hi;
  long long w_bstride;
  int per_sample, nsplit;
  int Mz;                  // pixels per z group
  int pchunks;             // ceil(Mz/32)
};

// double-buffered variants (conv_igemm_db.hip); tile ids 13 / 14 / 15 = 64x64 / 64x128 / 128x64
int fsv_launch_conv_db(const ConvP& p, int nz, hipStream_t stream, int tile);
int fsv_launch_wgrad_db(const WgradP& p, int bmk, int bn, dim3 grid, hipStream_t stream);      // 64x64 / 64x128, force_tile 7 / 8
