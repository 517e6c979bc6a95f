// Parameter block and tap decoding shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_np.hip).
#pragma once
#include "fsv_common.h"

#define FSV_BK 32
#define FSV_GROUP_MAX 16      // problems per grouped launch (the table is a kernel argument: 16 x 192 B + header < 4 KB)

// ---- 16-byte loads through a buffer descriptor -------------------------------------------------------------------------
// The gather-GEMM kernels need "this row / tap / column does not exist -> zeros".  A select on the LOADED value makes the
// compiler wait for the load right where it was issued, and a select between two POINTERS is turned into a branch around
// two loads with a vmcnt(0) each (round-2 ISA audit; cdna_hip_programming.md section 5 trap 4c).  A buffer load solves it in
// hardware: the per-lane byte offset is range-checked against the descriptor and an out-of-range lane returns zeros
// without touching memory - so "absent" is just the offset FSV_BUF_OOB, an integer select before the load.  Descriptors
// are built from kernel arguments and blockIdx only (provably wave-uniform: no waterfall loop, guide T20).
#define FSV_BUF_OOB 0x80000000u
#define FSV_BUF_MAX_BYTES 0x80000000ll      // one descriptor covers at most 2 GiB
#ifdef FSV_EMU
struct fsv_buf { const char* base; unsigned bytes; };
static inline fsv_buf fsv_make_buf(const void* p, long long bytes) { fsv_buf b; b.base = (const char*)p; b.bytes = (unsigned)bytes; return b; }
static inline float4 fsv_buf_load4(const fsv_buf& b, unsigned off) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (off < b.bytes && (unsigned long long)off + 16ull <= (unsigned long long)b.bytes) memcpy(&v, b.base + off, 16);
  return v;
}
static inline float fsv_buf_load1(const fsv_buf& b, unsigned off) {
  float v = 0.f;
  if (off < b.bytes && (unsigned long long)off + 4ull <= (unsigned long long)b.bytes) memcpy(&v, b.base + off, 4);
  return v;
}
static inline float fsv_buf_load_h(const fsv_buf& b, unsigned off) {        // one IEEE half, widened
  _Float16 v = (_Float16)0.f;
  if (off < b.bytes && (unsigned long long)off + 2ull <= (unsigned long long)b.bytes) memcpy(&v, b.base + off, 2);
  return (float)v;
}
// LDS-direct form: the lane's 16 bytes land at lds_wave_base + lane * 16 (out-of-range lanes write zeros)
typedef fsv_buf fsv_rawbuf;
static inline fsv_rawbuf fsv_make_rawbuf(const void* p, long long bytes) { return fsv_make_buf(p, bytes); }
static inline void fsv_buf_load4_lds(const fsv_rawbuf& b, unsigned off, float* lds_wave_base) {
  const float4 v = fsv_buf_load4(b, off);
  memcpy(lds_wave_base + (threadIdx.x & 63) * 4, &v, 16);
}
#define FSV_WAIT_VMCNT(n) ((void)0)
#define FSV_SCHED_FENCE() ((void)0)
#else
typedef __amdgpu_buffer_rsrc_t fsv_buf;
typedef unsigned int fsv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fsv_buf fsv_make_buf(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 fsv_buf_load4(fsv_buf b, unsigned off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(b, off, 0, 0));
}
__device__ __forceinline__ float fsv_buf_load1(fsv_buf b, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, off, 0, 0));
}
__device__ __forceinline__ float fsv_buf_load_h(fsv_buf b, unsigned off) {   // one IEEE half, widened
  return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(b, off, 0, 0));
}
// buffer_load_dwordx4 ... lds: no destination registers; the wave's 64 quads are written to 1 KB of LDS starting at the wave-uniform
// address lds_wave_base (M0), lane l at + 16 l; counted by vmcnt like any other buffer load.  Issued through inline assembly on
// purpose: with the builtin the compiler's wait-count pass treats every later ds_read as a possible reader of the data in flight
// and puts s_waitcnt vmcnt(0) in front of it (alias scopes did not survive the loop structure); the kernel that uses this waits for
// its own loads explicitly (FSV_WAIT_VMCNT) in front of the barrier that publishes them.
typedef int fsv_rawbuf __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fsv_rawbuf fsv_make_rawbuf(const void* p, long long bytes) {
  const unsigned long long a = (unsigned long long)p;
  fsv_rawbuf r;
  r.x = (int)(unsigned)a; r.y = (int)(unsigned)((a >> 32) & 0xffffull); r.z = (int)bytes; r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ void fsv_buf_load4_lds(fsv_rawbuf b, unsigned off, float* lds_wave_base) {
  const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);
  // (the descriptor is built from kernel arguments only: uniform, the "s" constraint needs no readfirstlane)
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(l), "v"(off), "s"(b) : "memory");
}
// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
#define FSV_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8))
// nothing is scheduled across this point: keeps the next chunk's loads at the top of the K loop and their LDS stores (with
// the vmcnt waits they carry) behind the MFMAs that cover the latency
#define FSV_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

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
  // optional: per-channel sums of the stored output for the normalisation that follows (BatchNorm / InstanceNorm statistics
  // from the producing convolution's epilogue instead of a read pass over its output): zeroed partials
  // stats[group][slot][Cout][2] = (sum v, sum v^2), the layout fsv_stats_final_kernel (norm.hip) finishes; a pixel tile adds into
  // slot (tile index % stats_slots); group = pixel / stats_ohw (one group for BatchNorm, one per sample for InstanceNorm)
  double* stats;
  int stats_slots, stats_ohw;
  long long res_bytes;                   // extent of the residual tensor when one descriptor covers it (else 0: loaded per element)
  float* part;                           // ordered split-K: split k stores its partial output at part + k * part_stride (else null:
  long long part_stride;                 // splits add into the zeroed output atomically)
  int band;                              // XCD bands: an XCD owns a CONTIGUOUS run of (pixel tile, channel tile) pairs (conv_igemm.hip)
  int up;                                // 1: `in` is stored at HALF the resolution (H / 2 x W / 2) and read through the nearest x2
                                         // up-sampling index (y >> 1, x >> 1) - the up-sampled tensor (generator.py:124, 497-504, 541-572:
                                         // nn.Upsample in front of a 3x3 convolution) is never written; H, W stay the logical size
};

__device__ __forceinline__ void fsv_tap(const ConvP& p, int t, int& ty, int& tx) {
  unsigned long long code = (t < 8) ? p.taps_lo : p.taps_hi;
  int sh = (t & 7) * 8;
  ty = (int)((code >> sh) & 15ull) - 8;
  tx = (int)((code >> (sh + 4)) & 15ull) - 8;
}

// One problem of a grouped launch at the C ABI: the arguments of fsv_conv_gather_fwd / fsv_conv_wgrad as plain structs.
// MUST match include/fsv2v.h (fsv_conv_desc / fsv_wgrad_desc) and the ctypes mirrors in few-shot-vid2vid_amd/conv.py.
struct fsv_conv_desc {
  const float* in; const float* wt; const float* bias; const float* res; float* out; const float* wscale;
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, outH, outW, osy, osx, ooy, oox, ldw;
  int per_sample, act, accumulate;
  float scale;
  long long w_bstride, b_bstride;
};
struct fsv_wgrad_desc {
  const float* in; const float* dout; float* dwt;
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, ldw, Kpad;
  int per_sample, reserved;
  long long w_bstride;
};

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
  int up;                  // as ConvP::up: `in` at half resolution behind a folded nearest x2 up-sampling (shift amount 0 / 1)
};
