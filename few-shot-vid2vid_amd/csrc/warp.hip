// Flow warp ("resample") of the few-shot-vid2vid generator and flow losses.
//
// Reference: models/networks/base_network.py:13-37 (get_grid + resample): the pixel-unit flow is turned into
// a normalised grid  g = linspace(-1, 1, n)[i] + flow / ((n - 1) / 2)  and handed to
// F.grid_sample(bilinear, padding_mode='border', align_corners=True).  ATen then un-normalises
// ix = (g + 1) * ((n - 1) / 2), clips to [0, n - 1] and floors.  That round trip is not the identity in fp32
// (at zero flow 76 of 512 columns land on x - 1), so this kernel replays the very same
// sequence of fp32 operations - it is compiled with -ffp-contract=off - and therefore selects bit-identical
// tap indices.  The base grid is taken from the host (torch.linspace, exactly what the reference uploads).
//
// HBM-bound: per output pixel 8 B of flow + (C taps, mostly L2 hits) + 4*C B written.  One work-item per
// output pixel, a wave covers 64 consecutive x so flow reads and output writes are coalesced for planar
// (NCHW) images; strides are explicit so channels-last images work too.
#include "fsv_common.h"

struct WarpP {
  const float* img;
  const float* flow;
  const float* lin_x;
  const float* lin_y;
  float* out;
  int* taps;       // optional [B,H,W,2] (x_w, y_n) for the index-parity tests
  int B, C, H, W;
  long long isb, isc, isy, isx;
  long long fsb, fsc, fsy, fsx;
  long long osb, osc, osy, osx;
};

struct WarpCoord {
  int x0, y0;
  float wx, wy;      // fractional parts (weight of the east / south tap)
  float mx, my;      // d(ix)/d(flow_x) in {0, 1}: 0 where the coordinate was clipped
};

__device__ __forceinline__ WarpCoord fsv_warp_coord(float fx, float fy, float linx, float liny, int W, int H) {
  WarpCoord c;
  const float hx = (float)(W - 1) / 2.0f, hy = (float)(H - 1) / 2.0f;
  float gx = linx + fx / hx;                 // IEEE division, as on the CPU path of the reference
  float gy = liny + fy / hy;
  float ix = (gx + 1.0f) * hx;               // grid_sampler_unnormalize, align_corners=True
  float iy = (gy + 1.0f) * hy;
  const float maxx = (float)(W - 1), maxy = (float)(H - 1);
  c.mx = (ix > 0.0f && ix < maxx) ? 1.0f : 0.0f;   // clip_coordinates_set_grad: borders count as clipped
  c.my = (iy > 0.0f && iy < maxy) ? 1.0f : 0.0f;
  ix = fminf(maxx, fmaxf(ix, 0.0f));
  iy = fminf(maxy, fmaxf(iy, 0.0f));
  float xw = floorf(ix), yn = floorf(iy);
  c.x0 = (int)xw; c.y0 = (int)yn;
  c.wx = ix - xw; c.wy = iy - yn;
  return c;
}

__global__ __launch_bounds__(256) void fsv_warp_fwd_kernel(WarpP p) {
  const long long total = (long long)p.B * p.H * p.W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.W);
  const int y = (int)((i / p.W) % p.H);
  const int b = (int)(i / ((long long)p.W * p.H));
  const float* fl = p.flow + b * p.fsb + y * p.fsy + x * p.fsx;
  WarpCoord c = fsv_warp_coord(fl[0], fl[p.fsc], p.lin_x[x], p.lin_y[y], p.W, p.H);
  if (p.taps) { p.taps[i * 2] = c.x0; p.taps[i * 2 + 1] = c.y0; }
  const float w = c.wx, e = 1.0f - c.wx, n = c.wy, s = 1.0f - c.wy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  const bool xe_ok = (c.x0 + 1) < p.W, ys_ok = (c.y0 + 1) < p.H;
  const float* base = p.img + b * p.isb + c.y0 * p.isy + c.x0 * p.isx;
  float* o = p.out + b * p.osb + y * p.osy + x * p.osx;
  for (int ch = 0; ch < p.C; ++ch) {
    const float* q = base + ch * p.isc;
    float vnw = q[0];
    float vne = xe_ok ? q[p.isx] : 0.0f;
    float vsw = ys_ok ? q[p.isy] : 0.0f;
    float vse = (xe_ok && ys_ok) ? q[p.isy + p.isx] : 0.0f;
    o[ch * p.osc] = vnw * wnw + vne * wne + vsw * wsw + vse * wse;
  }
}

struct WarpBwdP {
  const float* img;
  const float* flow;
  const float* lin_x;
  const float* lin_y;
  const float* gout;
  float* gimg;       // zero-initialised by the caller; may be null
  float* gflow;      // [B,2,H,W] with the flow strides; may be null
  int B, C, H, W;
  long long isb, isc, isy, isx;
  long long fsb, fsc, fsy, fsx;
  long long osb, osc, osy, osx;      // strides of gout
  long long gsb, gsc, gsy, gsx;      // strides of gimg
  long long hsb, hsc, hsy, hsx;      // strides of gflow
};

__global__ __launch_bounds__(256) void fsv_warp_bwd_kernel(WarpBwdP p) {
  const long long total = (long long)p.B * p.H * p.W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.W);
  const int y = (int)((i / p.W) % p.H);
  const int b = (int)(i / ((long long)p.W * p.H));
  const float* fl = p.flow + b * p.fsb + y * p.fsy + x * p.fsx;
  WarpCoord c = fsv_warp_coord(fl[0], fl[p.fsc], p.lin_x[x], p.lin_y[y], p.W, p.H);
  const float w = c.wx, e = 1.0f - c.wx, n = c.wy, s = 1.0f - c.wy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  const bool xe_ok = (c.x0 + 1) < p.W, ys_ok = (c.y0 + 1) < p.H;
  const float* base = p.img + b * p.isb + c.y0 * p.isy + c.x0 * p.isx;
  float* gbase = p.gimg ? p.gimg + b * p.gsb + c.y0 * p.gsy + c.x0 * p.gsx : nullptr;
  const float* go = p.gout + b * p.osb + y * p.osy + x * p.osx;
  float gx = 0.0f, gy = 0.0f;
  for (int ch = 0; ch < p.C; ++ch) {
    const float g = go[ch * p.osc];
    const float* q = base + ch * p.isc;
    float vnw = q[0];
    float vne = xe_ok ? q[p.isx] : 0.0f;
    float vsw = ys_ok ? q[p.isy] : 0.0f;
    float vse = (xe_ok && ys_ok) ? q[p.isy + p.isx] : 0.0f;
    gx += g * ((vne - vnw) * s + (vse - vsw) * n);
    gy += g * ((vsw - vnw) * e + (vse - vne) * w);
    if (gbase) {
      float* gq = gbase + ch * p.gsc;
      atomicAdd(gq, g * wnw);
      if (xe_ok) atomicAdd(gq + p.gsx, g * wne);
      if (ys_ok) atomicAdd(gq + p.gsy, g * wsw);
      if (xe_ok && ys_ok) atomicAdd(gq + p.gsy + p.gsx, g * wse);
    }
  }
  if (p.gflow) {
    float* gf = p.gflow + b * p.hsb + y * p.hsy + x * p.hsx;
    gf[0] = gx * c.mx;          // d ix / d flow_x = ((W-1)/2) / ((W-1)/2) = 1 where not clipped
    gf[p.hsc] = gy * c.my;
  }
}


// ---- warp + occlusion-mask compositing in one pass (generator.py:214-227 + base_network.py:28-37) ---------------------------
// The warped image is never used alone: with --spade_combine it is concatenated with the mask into the 4-channel input of the
// image embedding (ds_ref = cat([warp, mask]), generator.py:441-443), otherwise it is blended into the synthesised image
// (img_final = raw * mask + warp * (1 - mask), generator.py:217,224).  One work-item per output pixel does the bilinear taps
// (same fsv_warp_coord, so the tap indices stay bit-identical), writes the warped pixel (the losses read it) and the composite:
//   FSV_COMPOSE_CONCAT : comp = dense NHWC [B][H*W][C + 1] = (warp, mask)   - one 16-byte store per pixel for RGB
//   FSV_COMPOSE_BLEND  : comp[b, c, y, x] = raw * m + warp * (1 - m)           (strides of `comp` given)
#define FSV_COMPOSE_CONCAT 0
#define FSV_COMPOSE_BLEND 1
#define FSV_COMPOSE_MAXC 8
struct WarpCompP {
  const float* img; const float* flow; const float* lin_x; const float* lin_y;
  const float* mask;       // [B, 1, H, W]: strides msb, msy, msx
  const float* raw;        // BLEND only
  float* warp;             // forward: out; backward: upstream gradient w.r.t. warp (may be null)
  float* comp;             // forward: out; backward: upstream gradient w.r.t. comp (may be null)
  float* gimg;             // backward, optional (zero-initialised), strides of img
  float* gflow;            // backward: [B, 2, H, W] dense
  float* gmask;            // backward: [B, H, W] dense
  float* graw;             // backward, BLEND: [B, C, H, W] dense
  int mode, B, C, H, W;
  long long isb, isc, isy, isx;
  long long fsb, fsc, fsy, fsx;
  long long msb, msy, msx;
  long long rsb, rsc, rsy, rsx;
  long long wsb, wsc, wsy, wsx;
  long long csb, csc, csy, csx;
  long long gsb, gsc, gsy, gsx;      // gimg
};

__global__ __launch_bounds__(256) void fsv_warp_compose_fwd_kernel(WarpCompP p) {
  const long long total = (long long)p.B * p.H * p.W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.W);
  const int y = (int)((i / p.W) % p.H);
  const int b = (int)(i / ((long long)p.W * p.H));
  const float* fl = p.flow + b * p.fsb + y * p.fsy + x * p.fsx;
  WarpCoord c = fsv_warp_coord(fl[0], fl[p.fsc], p.lin_x[x], p.lin_y[y], p.W, p.H);
  const float w = c.wx, e = 1.0f - c.wx, n = c.wy, s = 1.0f - c.wy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  const bool xe_ok = (c.x0 + 1) < p.W, ys_ok = (c.y0 + 1) < p.H;
  const float* base = p.img + b * p.isb + c.y0 * p.isy + c.x0 * p.isx;
  const float m = p.mask[b * p.msb + y * p.msy + x * p.msx];
  float* wo = p.warp + b * p.wsb + y * p.wsy + x * p.wsx;
  float vals[FSV_COMPOSE_MAXC];
#pragma unroll
  for (int ch = 0; ch < FSV_COMPOSE_MAXC; ++ch) {
    if (ch < p.C) {
      const float* q = base + ch * p.isc;
      float vnw = q[0];
      float vne = xe_ok ? q[p.isx] : 0.0f;
      float vsw = ys_ok ? q[p.isy] : 0.0f;
      float vse = (xe_ok && ys_ok) ? q[p.isy + p.isx] : 0.0f;
      vals[ch] = vnw * wnw + vne * wne + vsw * wsw + vse * wse;
      wo[ch * p.wsc] = vals[ch];
    }
  }
  if (p.mode == FSV_COMPOSE_CONCAT) {
    float* co = p.comp + i * (p.C + 1);
    if (p.C == 3) {
      *reinterpret_cast<float4*>(co) = make_float4(vals[0], vals[1], vals[2], m);
    } else {
#pragma unroll
      for (int ch = 0; ch < FSV_COMPOSE_MAXC; ++ch) if (ch < p.C) co[ch] = vals[ch];
      co[p.C] = m;
    }
  } else {
    const float* ro = p.raw + b * p.rsb + y * p.rsy + x * p.rsx;
    float* co = p.comp + b * p.csb + y * p.csy + x * p.csx;
#pragma unroll
    for (int ch = 0; ch < FSV_COMPOSE_MAXC; ++ch)
      if (ch < p.C) co[ch * p.csc] = ro[ch * p.rsc] * m + vals[ch] * (1.f - m);
  }
}

// p.warp / p.comp carry the upstream gradients (strides ws* / cs*; CONCAT: comp gradient dense NHWC [B][H*W][C+1])
__global__ __launch_bounds__(256) void fsv_warp_compose_bwd_kernel(WarpCompP p) {
  const long long total = (long long)p.B * p.H * p.W;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % p.W);
  const int y = (int)((i / p.W) % p.H);
  const int b = (int)(i / ((long long)p.W * p.H));
  const float* fl = p.flow + b * p.fsb + y * p.fsy + x * p.fsx;
  WarpCoord c = fsv_warp_coord(fl[0], fl[p.fsc], p.lin_x[x], p.lin_y[y], p.W, p.H);
  const float w = c.wx, e = 1.0f - c.wx, n = c.wy, s = 1.0f - c.wy;
  const float wnw = s * e, wne = s * w, wsw = n * e, wse = n * w;
  const bool xe_ok = (c.x0 + 1) < p.W, ys_ok = (c.y0 + 1) < p.H;
  const float* base = p.img + b * p.isb + c.y0 * p.isy + c.x0 * p.isx;
  float* gbase = p.gimg ? p.gimg + b * p.gsb + c.y0 * p.gsy + c.x0 * p.gsx : nullptr;
  const float m = p.mask[b * p.msb + y * p.msy + x * p.msx];
  const float* gw = p.warp ? p.warp + b * p.wsb + y * p.wsy + x * p.wsx : nullptr;
  const float* gc = nullptr;
  if (p.comp) gc = (p.mode == FSV_COMPOSE_CONCAT) ? p.comp + i * (p.C + 1) : p.comp + b * p.csb + y * p.csy + x * p.csx;
  const float* ro = (p.mode == FSV_COMPOSE_BLEND) ? p.raw + b * p.rsb + y * p.rsy + x * p.rsx : nullptr;
  float gx = 0.0f, gy = 0.0f, gm = 0.0f;
  for (int ch = 0; ch < p.C; ++ch) {
    const float* q = base + ch * p.isc;
    float vnw = q[0];
    float vne = xe_ok ? q[p.isx] : 0.0f;
    float vsw = ys_ok ? q[p.isy] : 0.0f;
    float vse = (xe_ok && ys_ok) ? q[p.isy + p.isx] : 0.0f;
    float g = gw ? gw[ch * p.wsc] : 0.0f;
    if (p.mode == FSV_COMPOSE_CONCAT) {
      if (gc) g = g + gc[ch];
    } else {
      const float go = gc ? gc[ch * p.csc] : 0.0f;
      const float val = vnw * wnw + vne * wne + vsw * wsw + vse * wse;
      g = g + go * (1.f - m);
      if (p.graw) p.graw[((long long)b * p.C + ch) * p.H * p.W + (long long)y * p.W + x] = go * m;
      gm += go * (ro[ch * p.rsc] - val);
    }
    gx += g * ((vne - vnw) * s + (vse - vsw) * n);
    gy += g * ((vsw - vnw) * e + (vse - vne) * w);
    if (gbase) {
      float* gq = gbase + ch * p.gsc;
      atomicAdd(gq, g * wnw);
      if (xe_ok) atomicAdd(gq + p.gsx, g * wne);
      if (ys_ok) atomicAdd(gq + p.gsy, g * wsw);
      if (xe_ok && ys_ok) atomicAdd(gq + p.gsy + p.gsx, g * wse);
    }
  }
  if (p.mode == FSV_COMPOSE_CONCAT) gm = gc ? gc[p.C] : 0.0f;
  if (p.gmask) p.gmask[i] = gm;
  if (p.gflow) {
    float* gf = p.gflow + ((long long)b * 2) * p.H * p.W + (long long)y * p.W + x;
    gf[0] = gx * c.mx;
    gf[(long long)p.H * p.W] = gy * c.my;
  }
}

extern "C" {

// strides are in elements, order (batch, channel, y, x)
int fsv_warp_fwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, float* out, int* taps,
                 int B, int C, int H, int W, const long long* img_strides, const long long* flow_strides,
                 const long long* out_strides, hipStream_t stream) {
  if (!img || !flow || !lin_x || !lin_y || !out || B < 1 || C < 1 || H < 2 || W < 2) return FSV_ERR_BAD_ARG;
  WarpP p;
  p.img = img; p.flow = flow; p.lin_x = lin_x; p.lin_y = lin_y; p.out = out; p.taps = taps;
  p.B = B; p.C = C; p.H = H; p.W = W;
  p.isb = img_strides[0]; p.isc = img_strides[1]; p.isy = img_strides[2]; p.isx = img_strides[3];
  p.fsb = flow_strides[0]; p.fsc = flow_strides[1]; p.fsy = flow_strides[2]; p.fsx = flow_strides[3];
  p.osb = out_strides[0]; p.osc = out_strides[1]; p.osy = out_strides[2]; p.osx = out_strides[3];
  long long total = (long long)B * H * W;
  FSV_LAUNCH(fsv_warp_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, p);
  return fsv_check_launch();
}

int fsv_warp_bwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* gout,
                 float* gimg, float* gflow, int B, int C, int H, int W, const long long* img_strides,
                 const long long* flow_strides, const long long* gout_strides, const long long* gimg_strides,
                 const long long* gflow_strides, hipStream_t stream) {
  if (!img || !flow || !lin_x || !lin_y || !gout || B < 1 || C < 1 || H < 2 || W < 2) return FSV_ERR_BAD_ARG;
  WarpBwdP p;
  p.img = img; p.flow = flow; p.lin_x = lin_x; p.lin_y = lin_y; p.gout = gout; p.gimg = gimg; p.gflow = gflow;
  p.B = B; p.C = C; p.H = H; p.W = W;
  p.isb = img_strides[0]; p.isc = img_strides[1]; p.isy = img_strides[2]; p.isx = img_strides[3];
  p.fsb = flow_strides[0]; p.fsc = flow_strides[1]; p.fsy = flow_strides[2]; p.fsx = flow_strides[3];
  p.osb = gout_strides[0]; p.osc = gout_strides[1]; p.osy = gout_strides[2]; p.osx = gout_strides[3];
  if (gimg) { p.gsb = gimg_strides[0]; p.gsc = gimg_strides[1]; p.gsy = gimg_strides[2]; p.gsx = gimg_strides[3]; }
  else { p.gsb = p.gsc = p.gsy = p.gsx = 0; }
  if (gflow) { p.hsb = gflow_strides[0]; p.hsc = gflow_strides[1]; p.hsy = gflow_strides[2]; p.hsx = gflow_strides[3]; }
  else { p.hsb = p.hsc = p.hsy = p.hsx = 0; }
  long long total = (long long)B * H * W;
  FSV_LAUNCH(fsv_warp_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, p);
  return fsv_check_launch();
}

static inline void fsv_wc_common(WarpCompP& p, const float* img, const float* flow, const float* lin_x, const float* lin_y,
                                 const float* mask, const float* raw, int mode, int B, int C, int H, int W,
                                 const long long* img_strides, const long long* flow_strides, const long long* mask_strides,
                                 const long long* raw_strides, const long long* warp_strides, const long long* comp_strides) {
  p.img = img; p.flow = flow; p.lin_x = lin_x; p.lin_y = lin_y; p.mask = mask; p.raw = raw;
  p.mode = mode; p.B = B; p.C = C; p.H = H; p.W = W;
  p.isb = img_strides[0]; p.isc = img_strides[1]; p.isy = img_strides[2]; p.isx = img_strides[3];
  p.fsb = flow_strides[0]; p.fsc = flow_strides[1]; p.fsy = flow_strides[2]; p.fsx = flow_strides[3];
  p.msb = mask_strides[0]; p.msy = mask_strides[1]; p.msx = mask_strides[2];
  p.rsb = p.rsc = p.rsy = p.rsx = 0;
  if (raw_strides) { p.rsb = raw_strides[0]; p.rsc = raw_strides[1]; p.rsy = raw_strides[2]; p.rsx = raw_strides[3]; }
  p.wsb = p.wsc = p.wsy = p.wsx = 0;
  if (warp_strides) { p.wsb = warp_strides[0]; p.wsc = warp_strides[1]; p.wsy = warp_strides[2]; p.wsx = warp_strides[3]; }
  p.csb = p.csc = p.csy = p.csx = 0;
  if (comp_strides) { p.csb = comp_strides[0]; p.csc = comp_strides[1]; p.csy = comp_strides[2]; p.csx = comp_strides[3]; }
  p.gimg = nullptr; p.gflow = nullptr; p.gmask = nullptr; p.graw = nullptr; p.warp = nullptr; p.comp = nullptr;
  p.gsb = p.gsc = p.gsy = p.gsx = 0;
}

// mode 0 (concat): comp = dense NHWC [B][H*W][C+1] (comp_strides ignored); mode 1 (blend): comp strided, raw required.
// mask_strides = (batch, y, x).  C <= 8.
int fsv_warp_compose_fwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* mask,
                         const float* raw, float* warp, float* comp, int mode, int B, int C, int H, int W,
                         const long long* img_strides, const long long* flow_strides, const long long* mask_strides,
                         const long long* raw_strides, const long long* warp_strides, const long long* comp_strides,
                         hipStream_t stream) {
  if (!img || !flow || !lin_x || !lin_y || !mask || !warp || !comp || B < 1 || C < 1 || C > FSV_COMPOSE_MAXC || H < 2 || W < 2)
    return FSV_ERR_BAD_ARG;
  if (mode != FSV_COMPOSE_CONCAT && mode != FSV_COMPOSE_BLEND) return FSV_ERR_BAD_ARG;
  if (mode == FSV_COMPOSE_BLEND && (!raw || !raw_strides || !comp_strides)) return FSV_ERR_BAD_ARG;
  if (!img_strides || !flow_strides || !mask_strides || !warp_strides) return FSV_ERR_BAD_ARG;
  WarpCompP p;
  fsv_wc_common(p, img, flow, lin_x, lin_y, mask, raw, mode, B, C, H, W, img_strides, flow_strides, mask_strides, raw_strides,
                warp_strides, comp_strides);
  p.warp = warp; p.comp = comp;
  long long total = (long long)B * H * W;
  FSV_LAUNCH(fsv_warp_compose_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, p);
  return fsv_check_launch();
}

// g_warp / g_comp: upstream gradients (either may be null); gimg (optional, zero-initialised, own strides), gflow
// [B,2,H,W] dense, gmask [B,H,W] dense, graw [B,C,H,W] dense (blend only) - each optional.
int fsv_warp_compose_bwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* mask,
                         const float* raw, const float* g_warp, const float* g_comp, float* gimg, float* gflow, float* gmask,
                         float* graw, int mode, int B, int C, int H, int W, const long long* img_strides,
                         const long long* flow_strides, const long long* mask_strides, const long long* raw_strides,
                         const long long* g_warp_strides, const long long* g_comp_strides, const long long* gimg_strides,
                         hipStream_t stream) {
  if (gimg && !gimg_strides) return FSV_ERR_BAD_ARG;
  if (!img || !flow || !lin_x || !lin_y || !mask || B < 1 || C < 1 || C > FSV_COMPOSE_MAXC || H < 2 || W < 2) return FSV_ERR_BAD_ARG;
  if (mode != FSV_COMPOSE_CONCAT && mode != FSV_COMPOSE_BLEND) return FSV_ERR_BAD_ARG;
  if (mode == FSV_COMPOSE_BLEND && (!raw || !raw_strides || (g_comp && !g_comp_strides))) return FSV_ERR_BAD_ARG;
  if (!img_strides || !flow_strides || !mask_strides || (g_warp && !g_warp_strides)) return FSV_ERR_BAD_ARG;
  WarpCompP p;
  fsv_wc_common(p, img, flow, lin_x, lin_y, mask, raw, mode, B, C, H, W, img_strides, flow_strides, mask_strides, raw_strides,
                g_warp ? g_warp_strides : nullptr, g_comp ? g_comp_strides : nullptr);
  p.warp = const_cast<float*>(g_warp); p.comp = const_cast<float*>(g_comp);
  p.gimg = gimg; p.gflow = gflow; p.gmask = gmask; p.graw = graw;
  if (gimg) { p.gsb = gimg_strides[0]; p.gsc = gimg_strides[1]; p.gsy = gimg_strides[2]; p.gsx = gimg_strides[3]; }
  long long total = (long long)B * H * W;
  FSV_LAUNCH(fsv_warp_compose_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, p);
  return fsv_check_launch();
}

}  // extern "C"
