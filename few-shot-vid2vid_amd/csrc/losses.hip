// Loss reductions, discriminator-input packing and label-mask pooling of the G/D step (SURVEY.md section 8a rows
// a-10 and a-13) - all HBM-bound, all fused into single passes.
//
// Reference: models/networks/loss.py:69-83 (hinge), :130-138 (MaskedL1Loss), torch.nn.L1Loss at
// models/loss_collector.py:36,152,156,206-215; D input concatenation loss_collector.py:47-58,105-110;
// MaxPool2d(15)/AvgPool2d(15) masks input_process.py:59, loss_collector.py:180.
#include "fsv_common.h"

#define FSV_LOSS_BLOCKS 512

__device__ __forceinline__ double fsv_block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  double r = red[0];
  __syncthreads();
  return r;
}

// ---- masked L1: sum_p sum_c | m_p * (a_pc - b_pc) | -----------------------------------------------------------------
// a, b: [N][C][P] with explicit element strides (batch, channel, pixel); b may be null -> constant bconst.
// m: [N][P] contiguous or null (-> 1).  One work-item per (n, pixel); channels are looped.
struct L1P {
  const float* a; const float* b; const float* m;
  long long asn, asc, asp, bsn, bsc, bsp;
  float bconst;
  int N, C; long long P;
};

// Second stage of a two-stage loss reduction inside the first launch (round 6): the workgroup that takes the last ticket sums the
// gridDim.x partials - in index order, whoever it is: the same bits as fsv_loss_final_kernel - and leaves the ticket at zero for the
// next launch.  `ticket`: one zeroed int per launch in flight (conv.ticket_range); null: the caller finishes with fsv_loss_final_kernel.
__device__ __forceinline__ void fsv_loss_finish(double* part, int* ticket, float* out, double scale, double* red) {
  __shared__ int is_last;
  __threadfence();                       // this workgroup's partial is visible device-wide before the ticket is taken
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atomicAdd(ticket, 1);
    is_last = (prev == (int)gridDim.x - 1) ? 1 : 0;
    if (is_last) ticket[0] = 0;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();                       // acquire: the other workgroups' partials
  double a = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += part[i];
  const double t = fsv_block_sum(a, red);
  if (threadIdx.x == 0) out[0] = (float)(t * scale);
}

__global__ __launch_bounds__(256) void fsv_l1_fwd_kernel(L1P p, double* part, int* ticket, float* out, double scale) {
  __shared__ double red[256];
  const long long total = (long long)p.N * p.P;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.P, px = i - n * p.P;
    const float mv = p.m ? p.m[i] : 1.f;
    float s = 0.f;
    for (int c = 0; c < p.C; ++c) {
      float av = p.a[n * p.asn + c * p.asc + px * p.asp];
      float bv = p.b ? p.b[n * p.bsn + c * p.bsc + px * p.bsp] : p.bconst;
      s += fabsf(av * mv - bv * mv);
    }
    acc += (double)s;
  }
  double t = fsv_block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
  if (ticket) fsv_loss_finish(part, ticket, out, scale, red);
}

// out[0] = scale * sum(part)
__global__ void fsv_loss_final_kernel(const double* part, int nparts, float* out, double scale) {
  __shared__ double red[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  double t = fsv_block_sum(a, red);
  if (threadIdx.x == 0) out[0] = (float)(t * scale);
}

// gradients: da = g * sgn * m / cnt, db = -da, dm = g * sum_c sgn * (a - b) / cnt, sgn = sign(a*m - b*m)
__global__ __launch_bounds__(256) void fsv_l1_bwd_kernel(L1P p, const float* gptr, float inv_cnt, float* da, float* db, float* dm) {
  const long long total = (long long)p.N * p.P;
  const float g = gptr[0] * inv_cnt;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / p.P, px = i - n * p.P;
    const float mv = p.m ? p.m[i] : 1.f;
    float gm = 0.f;
    for (int c = 0; c < p.C; ++c) {
      const long long ia = n * p.asn + c * p.asc + px * p.asp;
      const long long ib = n * p.bsn + c * p.bsc + px * p.bsp;
      float av = p.a[ia];
      float bv = p.b ? p.b[ib] : p.bconst;
      float d = av * mv - bv * mv;
      float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      if (da) da[ia] = g * sg * mv;             // da / db share the layout (strides) of a / b
      if (db) db[ib] = -g * sg * mv;
      gm += sg * (av - bv);
    }
    if (dm) dm[i] = g * gm;
  }
}

// ---- hinge: sum min(sign * x - 1, 0) ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsv_hinge_fwd_kernel(const float* x, long long n, float sign, double* part, int* ticket,
                                                            float* out, double scale) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    acc += (double)fminf(sign * x[i] - 1.f, 0.f);
  double t = fsv_block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
  if (ticket) fsv_loss_finish(part, ticket, out, scale, red);
}

// loss = -(1/n) sum min(s x - 1, 0)  ->  dx = g * (-1/n) * s * [s x - 1 < 0]   (ties: 1/2, as torch.min does)
__global__ __launch_bounds__(256) void fsv_hinge_bwd_kernel(const float* x, long long n, float sign, const float* gptr, float* dx) {
  const float g = -gptr[0] / (float)n * sign;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = sign * x[i] - 1.f;
    dx[i] = v < 0.f ? g : (v == 0.f ? 0.5f * g : 0.f);
  }
}

// ---- discriminator input: out[2B][H][W][Cr + Cl + Ci] (NHWC) = [ref | label | (fake for n < B, real otherwise)] -------------
struct PackP {
  const float* ref; const float* lab; const float* fake; const float* real;
  long long rs[3], ls[3], fs[3], es[3];      // element strides (batch, channel, pixel) of each source
  int Cr, Cl, Ci, B; long long P;
  float* out;
  int halves;        // 2: [fake | real] on the batch axis; 1: `fake` only (out [B])
  int Cto;           // channels of `out` (>= Cr + Cl + Ci: zero channels appended - the padding of the first discriminator convolution)
  int out_half;      // `out` as IEEE half (the `--amp` path: the packed tensor is a convolution input)
};

// A workgroup owns 64 consecutive pixels of one sample: per source channel the 64 values are read as one run (coalesced for
// planar sources, the layout the data loader delivers), staged in LDS and written out as 64 * Ct consecutive floats (the
// one-thread-per-pixel form of round 1 wrote 4-byte words Ct floats apart: 225 us for the 512x512 B=2 input).
#define FSV_PACK_PX 64
#define FSV_PACK_MAXC 64
__global__ __launch_bounds__(256) void fsv_pack_d_kernel(PackP p) {
  __shared__ float t[FSV_PACK_PX * (FSV_PACK_MAXC + 1)];
  const int Ct = p.Cr + p.Cl + p.Ci;
  const long long tiles = (p.P + FSV_PACK_PX - 1) / FSV_PACK_PX;
  const long long ntile = (long long)p.halves * p.B * tiles;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long long n2 = tile / tiles, px0 = (tile - n2 * tiles) * FSV_PACK_PX;
    const long long n = n2 % p.B;
    const bool isfake = n2 < p.B;
    const long long px = px0 + lane;
    const bool ok = px < p.P;
    const long long npx = (p.P - px0) < FSV_PACK_PX ? (p.P - px0) : FSV_PACK_PX;
    const long long obase = (n2 * p.P + px0) * p.Cto;
    float* o = p.out + obase;
    _Float16* oh = reinterpret_cast<_Float16*>(p.out) + obase;
    for (int c0 = 0; c0 < p.Cto; c0 += FSV_PACK_MAXC) {        // 64 channels per round (one-hot street labels: Ct = 76)
      const int cw = (p.Cto - c0) < FSV_PACK_MAXC ? (p.Cto - c0) : FSV_PACK_MAXC;
      for (int cl = grp; cl < cw; cl += 4) {
        const int c = c0 + cl;
        float v = 0.f;
        if (ok && c < Ct) {
          if (c < p.Cr) v = p.ref[n * p.rs[0] + c * p.rs[1] + px * p.rs[2]];
          else if (c < p.Cr + p.Cl) v = p.lab[n * p.ls[0] + (c - p.Cr) * p.ls[1] + px * p.ls[2]];
          else {
            const float* src = isfake ? p.fake : p.real;
            const long long* st = isfake ? p.fs : p.es;
            v = src[n * st[0] + (c - p.Cr - p.Cl) * st[1] + px * st[2]];
          }
        }
        t[lane * (FSV_PACK_MAXC + 1) + cl] = v;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < (int)npx * cw; i += 256) {
        const int pl = i / cw, cl = i - pl * cw;
        const float v = t[pl * (FSV_PACK_MAXC + 1) + cl];
        if (p.out_half) oh[(long long)pl * p.Cto + c0 + cl] = (_Float16)v; else o[(long long)pl * p.Cto + c0 + cl] = v;
      }
      __syncthreads();
    }
  }
}

// dfake[n][c][px] (contiguous NCHW) = dout[n][px][Cr + Cl + c]
template <bool HALF>
__global__ __launch_bounds__(256) void fsv_unpack_d_kernel(const float* dout, float* dfake, int B, int Ci, int Coff, int Ct, long long P) {
  const long long total = (long long)B * Ci * P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long px = i % P;
    const long long t = i / P;
    const int c = (int)(t % Ci);
    const long long n = t / Ci;
    if constexpr (HALF) dfake[i] = (float)reinterpret_cast<const _Float16*>(dout)[(n * P + px) * Ct + Coff + c];
    else dfake[i] = dout[(n * P + px) * Ct + Coff + c];
  }
}

// ---- 15x15 stride-1 pooling of single-channel masks (zero / -inf padding 7) ----------------------------------------------
// mode 0: max pool followed by (v > thresh) ? 1 : 0 ;  mode 1: average pool (count_include_pad = True)
// Separable: a workgroup stages a (32+14)^2 input tile in LDS, reduces 15-wide row windows, then 15-tall column windows
// (30 LDS reads per output instead of 225 global ones).
#define FSV_P15_T 32
#define FSV_P15_IN (FSV_P15_T + 14)
__global__ __launch_bounds__(256) void fsv_pool15_kernel(const float* x, float* y, int N, int H, int W, long long sn, long long sy,
                                                         long long sx, int mode, float thresh) {
  __shared__ float tin[FSV_P15_IN][FSV_P15_IN + 1];
  __shared__ float mid[FSV_P15_IN][FSV_P15_T + 1];
  const int n = blockIdx.z, y0 = blockIdx.y * FSV_P15_T, x0 = blockIdx.x * FSV_P15_T;
  const float* base = x + n * sn;
  const float padv = mode == 0 ? -3.0e38f : 0.f;
  for (int i = threadIdx.x; i < FSV_P15_IN * FSV_P15_IN; i += 256) {
    const int r = i / FSV_P15_IN, c = i - r * FSV_P15_IN;
    const int yy = y0 + r - 7, xx = x0 + c - 7;
    tin[r][c] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? base[yy * sy + xx * sx] : padv;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FSV_P15_IN * FSV_P15_T; i += 256) {
    const int r = i / FSV_P15_T, c = i - r * FSV_P15_T;
    float acc = padv;
#pragma unroll
    for (int d = 0; d < 15; ++d) acc = mode == 0 ? fmaxf(acc, tin[r][c + d]) : acc + tin[r][c + d];
    mid[r][c] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < FSV_P15_T * FSV_P15_T; i += 256) {
    const int r = i / FSV_P15_T, c = i - r * FSV_P15_T;
    const int yy = y0 + r, xx = x0 + c;
    if (yy >= H || xx >= W) continue;
    float acc = padv;
#pragma unroll
    for (int d = 0; d < 15; ++d) acc = mode == 0 ? fmaxf(acc, mid[r + d][c]) : acc + mid[r + d][c];
    y[((long long)n * H + yy) * W + xx] = mode == 0 ? (acc > thresh ? 1.f : 0.f) : acc * (1.f / 225.f);
  }
}

// ---- DensePose body-part group masks (models/input_process.py:64-94) --------------------------------------------------------
// pose channel value -> part = (v / 2 + 0.5) * 24; group g is hit when part lies in (j - 0.1, j + 0.1) for a member j.
// y[(n * ngroups + g) * P + p]; groups g0 .. g0 + ngroups - 1 of the table below (the last one, {23, 24}, is the face).
__constant__ int fsv_part_first[10] = {0, 1, 3, 5, 7, 11, 15, 19, 23, 25};        // members of group g: order below
__constant__ int fsv_part_member[25] = {0, 1, 2, 3, 4, 5, 6, 7, 9, 8, 10, 11, 13, 12, 14, 15, 17, 16, 18, 19, 21, 20, 22, 23, 24};
__global__ __launch_bounds__(256) void fsv_part_masks_kernel(const float* x, float* y, long long N, long long P, int T, long long sb,
                                                             long long st, int g0, int ngroups) {
  const long long total = N * P;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long n = i / P, p = i - n * P;
    const float v = x[(n / T) * sb + (n % T) * st + p];
    const float part = (v / 2.f + 0.5f) * 24.f;
    for (int g = 0; g < ngroups; ++g) {
      bool hit = false;
      for (int m = fsv_part_first[g0 + g]; m < fsv_part_first[g0 + g + 1]; ++m) {
        const int j = fsv_part_member[m];
        const float lo = (float)((double)j - 0.1), hi = (float)((double)j + 0.1);   // the scalars torch compares against
        hit = hit || (part > lo && part < hi);
      }
      y[(n * ngroups + g) * P + p] = hit ? 1.f : 0.f;
    }
  }
}

static inline int fsv_loss_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g > FSV_LOSS_BLOCKS) g = FSV_LOSS_BLOCKS;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- weighted sum of one-element loss tensors (round 6) ---------------------------------------------------------------------------
// `loss = sum(lambda_i * term_i)` (loss_collector.py:60-67,85,161-162,204 and :218-219, the sum of the means): the collector combines two
// dozen scalars per iteration in five such sums; as torch ops each is cat + mul + sum (and a mul + slices on the way back) of
// 5-us launches at the one serial point of the step, between the forward pass and the backward pass.  One launch each way: the
// pointers and weights travel in the kernel argument.
#define FSV_WSUM_MAX 32
struct WsumP { const float* t[FSV_WSUM_MAX]; float w[FSV_WSUM_MAX]; int n; };

__global__ void fsv_wsum_fwd_kernel(WsumP p, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < p.n; ++i) s += p.w[i] * p.t[i][0];       // ascending order, fp32 (what cat -> mul -> sum computed)
    out[0] = s;
  }
}

__global__ void fsv_wsum_bwd_kernel(WsumP p, const float* g, float* dterms) {
  const int i = threadIdx.x;
  if (blockIdx.x == 0 && i < p.n) dterms[i] = p.w[i] * g[0];
}

extern "C" {

// loss[0] = (1 / (N*C*P)) * sum |a*m - b*m|.  strides: 3 x long long (batch, channel, pixel).  part: double[512].
int fsv_l1_fwd(const float* a, const float* b, float bconst, const float* m, int N, int C, long long P,
               const long long* a_strides, const long long* b_strides, double* part, float* loss, int* ticket,
               hipStream_t stream) {
  if (!a || !part || !loss || N < 1 || C < 1 || P < 1) return FSV_ERR_BAD_ARG;
  L1P p;
  p.a = a; p.b = b; p.m = m; p.bconst = bconst; p.N = N; p.C = C; p.P = P;
  p.asn = a_strides[0]; p.asc = a_strides[1]; p.asp = a_strides[2];
  if (b) { p.bsn = b_strides[0]; p.bsc = b_strides[1]; p.bsp = b_strides[2]; } else { p.bsn = p.bsc = p.bsp = 0; }
  const int grid = fsv_loss_grid((long long)N * P);
  const double scale = 1.0 / ((double)N * C * P);
  FSV_LAUNCH(fsv_l1_fwd_kernel, dim3(grid), dim3(256), stream, p, part, ticket, loss, scale);
  if (!ticket) FSV_LAUNCH(fsv_loss_final_kernel, dim3(1), dim3(256), stream, (const double*)part, grid, loss, scale);
  return fsv_check_launch();
}

int fsv_l1_bwd(const float* a, const float* b, float bconst, const float* m, int N, int C, long long P,
               const long long* a_strides, const long long* b_strides, const float* gloss, float* da, float* db, float* dm,
               hipStream_t stream) {
  if (!a || !gloss || N < 1 || C < 1 || P < 1) return FSV_ERR_BAD_ARG;
  L1P p;
  p.a = a; p.b = b; p.m = m; p.bconst = bconst; p.N = N; p.C = C; p.P = P;
  p.asn = a_strides[0]; p.asc = a_strides[1]; p.asp = a_strides[2];
  if (b) { p.bsn = b_strides[0]; p.bsc = b_strides[1]; p.bsp = b_strides[2]; } else { p.bsn = p.bsc = p.bsp = 0; }
  FSV_LAUNCH(fsv_l1_bwd_kernel, dim3(fsv_loss_grid((long long)N * P)), dim3(256), stream, p, gloss,
             (float)(1.0 / ((double)N * C * P)), da, db, dm);
  return fsv_check_launch();
}

// loss[0] = -(1/n) * sum min(sign * x - 1, 0)
int fsv_hinge_fwd(const float* x, long long n, float sign, double* part, float* loss, int* ticket, hipStream_t stream) {
  if (!x || !part || !loss || n < 1) return FSV_ERR_BAD_ARG;
  const int grid = fsv_loss_grid(n);
  FSV_LAUNCH(fsv_hinge_fwd_kernel, dim3(grid), dim3(256), stream, x, n, sign, part, ticket, loss, -1.0 / (double)n);
  if (!ticket) FSV_LAUNCH(fsv_loss_final_kernel, dim3(1), dim3(256), stream, (const double*)part, grid, loss, -1.0 / (double)n);
  return fsv_check_launch();
}

int fsv_hinge_bwd(const float* x, long long n, float sign, const float* gloss, float* dx, hipStream_t stream) {
  if (!x || !gloss || !dx || n < 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_hinge_bwd_kernel, dim3(fsv_loss_grid(n)), dim3(256), stream, x, n, sign, gloss, dx);
  return fsv_check_launch();
}

static inline int fsv_pack_grid(long long tiles) { return (int)(tiles < 16384 ? (tiles < 1 ? 1 : tiles) : 16384); }

static int fsv_pack_d_impl(const float* ref, const float* lab, const float* fake, const float* real, float* out,
                           int B, int Cr, int Cl, int Ci, long long P, const long long* ref_strides, const long long* lab_strides,
                           const long long* fake_strides, const long long* real_strides, int halves, int Cto, int out_half,
                           hipStream_t stream) {
  if (!fake || !real || !out || B < 1 || Ci < 1 || P < 1 || (Cr > 0 && !ref) || (Cl > 0 && !lab) || Cto < Cr + Cl + Ci) return FSV_ERR_BAD_ARG;
  PackP p;
  p.ref = ref; p.lab = lab; p.fake = fake; p.real = real; p.out = out;
  for (int i = 0; i < 3; ++i) {
    p.rs[i] = ref ? ref_strides[i] : 0; p.ls[i] = lab ? lab_strides[i] : 0; p.fs[i] = fake_strides[i]; p.es[i] = real_strides[i];
  }
  p.Cr = Cr; p.Cl = Cl; p.Ci = Ci; p.B = B; p.P = P; p.halves = halves; p.Cto = Cto; p.out_half = out_half ? 1 : 0;
  FSV_LAUNCH(fsv_pack_d_kernel, dim3(fsv_pack_grid((long long)halves * B * ((P + FSV_PACK_PX - 1) / FSV_PACK_PX))), dim3(256), stream, p);
  return fsv_check_launch();
}

// the two packing entry points below with `Cto` >= Cr + Cl + Ci output channels (zero channels appended: the channel padding of the
// first discriminator convolution done here) and, out_half != 0, `out` as IEEE half - the `--amp` path, where the packed tensor
// is read by a half-precision convolution.  halves: 2 = [fake | real] on the batch axis, 1 = `fake` only.
int fsv_pack_d_x(const float* ref, const float* lab, const float* fake, const float* real, void* out,
                 int B, int Cr, int Cl, int Ci, long long P, const long long* ref_strides, const long long* lab_strides,
                 const long long* fake_strides, const long long* real_strides, int halves, int Cto, int out_half, hipStream_t stream) {
  if (halves != 1 && halves != 2) return FSV_ERR_BAD_ARG;
  return fsv_pack_d_impl(ref, lab, fake, real ? real : fake, reinterpret_cast<float*>(out), B, Cr, Cl, Ci, P, ref_strides, lab_strides,
                         fake_strides, real ? real_strides : fake_strides, halves, Cto, out_half, stream);
}

int fsv_pack_d_input(const float* ref, const float* lab, const float* fake, const float* real, float* out,
                     int B, int Cr, int Cl, int Ci, long long P, const long long* ref_strides, const long long* lab_strides,
                     const long long* fake_strides, const long long* real_strides, hipStream_t stream) {
  if (!fake || !real || !out || B < 1 || Ci < 1 || P < 1 || (Cr > 0 && !ref) || (Cl > 0 && !lab)) return FSV_ERR_BAD_ARG;
  PackP p;
  p.Cto = Cr + Cl + Ci; p.out_half = 0;
  p.ref = ref; p.lab = lab; p.fake = fake; p.real = real; p.out = out;
  for (int i = 0; i < 3; ++i) {
    p.rs[i] = ref ? ref_strides[i] : 0; p.ls[i] = lab ? lab_strides[i] : 0; p.fs[i] = fake_strides[i]; p.es[i] = real_strides[i];
  }
  p.Cr = Cr; p.Cl = Cl; p.Ci = Ci; p.B = B; p.P = P; p.halves = 2;
  FSV_LAUNCH(fsv_pack_d_kernel, dim3(fsv_pack_grid(2LL * B * ((P + FSV_PACK_PX - 1) / FSV_PACK_PX))), dim3(256), stream, p);
  return fsv_check_launch();
}

// one half only: out [B][P][Cr + Cl + Ci] = [ref | label | img] (the G step runs the discriminator on the real and on the
// generated images in separate passes: only the latter needs a backward pass)
int fsv_pack_d_single(const float* ref, const float* lab, const float* img, float* out, int B, int Cr, int Cl, int Ci,
                      long long P, const long long* ref_strides, const long long* lab_strides, const long long* img_strides,
                      hipStream_t stream) {
  if (!img || !out || B < 1 || Ci < 1 || P < 1 || (Cr > 0 && !ref) || (Cl > 0 && !lab)) return FSV_ERR_BAD_ARG;
  PackP p;
  p.Cto = Cr + Cl + Ci; p.out_half = 0;
  p.ref = ref; p.lab = lab; p.fake = img; p.real = img; p.out = out;
  for (int i = 0; i < 3; ++i) {
    p.rs[i] = ref ? ref_strides[i] : 0; p.ls[i] = lab ? lab_strides[i] : 0; p.fs[i] = img_strides[i]; p.es[i] = img_strides[i];
  }
  p.Cr = Cr; p.Cl = Cl; p.Ci = Ci; p.B = B; p.P = P; p.halves = 1;
  FSV_LAUNCH(fsv_pack_d_kernel, dim3(fsv_pack_grid((long long)B * ((P + FSV_PACK_PX - 1) / FSV_PACK_PX))), dim3(256), stream, p);
  return fsv_check_launch();
}

int fsv_unpack_d_grad(const float* dout, float* dfake, int B, int Ci, int Coff, int Ct, long long P, hipStream_t stream) {
  if (!dout || !dfake) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_unpack_d_kernel<false>, dim3(fsv_loss_grid((long long)B * Ci * P) * 4), dim3(256), stream, dout, dfake, B, Ci, Coff, Ct, P);
  return fsv_check_launch();
}

// the same from a half gradient tensor dout [B][P][Ct] (the data gradient of a half-precision convolution)
int fsv_unpack_d_grad_h(const void* dout, float* dfake, int B, int Ci, int Coff, int Ct, long long P, hipStream_t stream) {
  if (!dout || !dfake) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_unpack_d_kernel<true>, dim3(fsv_loss_grid((long long)B * Ci * P) * 4), dim3(256), stream,
             reinterpret_cast<const float*>(dout), dfake, B, Ci, Coff, Ct, P);
  return fsv_check_launch();
}

int fsv_part_masks(const float* x, float* y, long long N, long long P, int T, long long sb, long long st, int g0, int ngroups,
                   hipStream_t stream) {
  if (!x || !y || N < 1 || P < 1 || T < 1 || g0 < 0 || ngroups < 1 || g0 + ngroups > 9) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_part_masks_kernel, dim3(fsv_loss_grid(N * P) * 8), dim3(256), stream, x, y, N, P, T, sb, st, g0, ngroups);
  return fsv_check_launch();
}

int fsv_pool15(const float* x, float* y, int N, int H, int W, long long sn, long long sy, long long sx, int mode, float thresh,
               hipStream_t stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || mode < 0 || mode > 1) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_pool15_kernel, dim3(fsv_cdiv(W, FSV_P15_T), fsv_cdiv(H, FSV_P15_T), N), dim3(256), stream, x, y, N, H, W, sn,
             sy, sx, mode, thresh);
  return fsv_check_launch();
}

// out[0] = sum_i weights[i] * terms[i][0], i ascending; n <= 32 one-element device tensors
int fsv_wsum_fwd(const float* const* terms, const float* weights, int n, float* out, hipStream_t stream) {
  if (!terms || !weights || !out || n < 1 || n > FSV_WSUM_MAX) return FSV_ERR_BAD_ARG;
  WsumP p;
  for (int i = 0; i < FSV_WSUM_MAX; ++i) { p.t[i] = i < n ? terms[i] : nullptr; p.w[i] = i < n ? weights[i] : 0.f; }
  for (int i = 0; i < n; ++i) if (!terms[i]) return FSV_ERR_BAD_ARG;
  p.n = n;
  FSV_LAUNCH(fsv_wsum_fwd_kernel, dim3(1), dim3(64), stream, p, out);
  return fsv_check_launch();
}

// dterms[i] = weights[i] * g[0]: the gradients of the n terms, one float each
int fsv_wsum_bwd(const float* weights, int n, const float* g, float* dterms, hipStream_t stream) {
  if (!weights || !g || !dterms || n < 1 || n > FSV_WSUM_MAX) return FSV_ERR_BAD_ARG;
  WsumP p;
  for (int i = 0; i < FSV_WSUM_MAX; ++i) { p.t[i] = nullptr; p.w[i] = i < n ? weights[i] : 0.f; }
  p.n = n;
  FSV_LAUNCH(fsv_wsum_bwd_kernel, dim3(1), dim3(64), stream, p, g, dterms);
  return fsv_check_launch();
}

}  // extern "C"
