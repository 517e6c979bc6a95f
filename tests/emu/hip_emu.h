// SIMT emulator for running the package's HIP kernel sources on a CPU.  TEST INFRASTRUCTURE ONLY.
//
// The GPU box is only reachable for a few minutes per round, so the *same* kernel sources under
// few-shot-vid2vid_amd/csrc/ are also compiled for the host (clang++ -DFSV_EMU) against this header and
// executed by a cooperative fibre scheduler: one fibre per work-item, one workgroup at a time, round-robin
// between barriers.  Wave-level collectives (MFMA, shuffles) rendezvous the 64 fibres of a wave and are
// evaluated with the gfx950 lane layouts documented in /opt/skills/guides/cdna_hip_programming.md section 3.
// This checks indexing, tiling, bounds handling and the maths; it says nothing about performance or
// memory-model races.  Nothing in the product path (bench.py, smoke(), -m gpu tests) loads the emulated
// library: the loader requires FSV2V_EMU=1 to be set explicitly.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

#define __global__
#define __device__
#define __constant__ static const
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

namespace emu {

struct Fiber {
  void* sp = nullptr;       // saved stack pointer
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int lin = 0;              // linear thread id in block
};

struct WaveSlot {           // rendezvous area for one wave
  int arrived = 0;
  unsigned gen = 0;
  float a[64], b[64];
  float a8[64][8], b8[64][8];
  float c[64][16];
  float d[64][16];
  double dv[64];
  long long iv[64];
  int src[64];
};

struct State {
  dim3 grid, block, bid;
  std::vector<Fiber> fibers;
  std::vector<WaveSlot> waves;
  int cur = -1;
  int live = 0;
  int bar_arrived = 0;
  unsigned bar_gen = 0;
  void* sched_sp = nullptr;
  std::function<void()> body;
};

inline State& S() { static State s; return s; }

extern "C" void fsv_emu_switch(void** save_sp, void* load_sp);
// minimal x86-64 SysV context switch: save callee-saved registers on the current stack, swap rsp.
__asm__(
    ".text\n.globl fsv_emu_switch\n.type fsv_emu_switch,@function\nfsv_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size fsv_emu_switch,.-fsv_emu_switch\n");

inline void yield() {
  State& s = S();
  Fiber& f = s.fibers[s.cur];
  fsv_emu_switch(&f.sp, s.sched_sp);
}

inline void fiber_exit_check_barrier() {
  State& s = S();
  // a finished work-item no longer takes part in workgroup barriers
  if (s.live > 0 && s.bar_arrived == s.live) { s.bar_arrived = 0; s.bar_gen++; }
}

extern "C" inline void fsv_emu_entry() {
  State& s = S();
  s.body();
  Fiber& f = s.fibers[s.cur];
  f.done = true;
  s.live--;
  fiber_exit_check_barrier();
  fsv_emu_switch(&f.sp, s.sched_sp);
  abort();
}

inline void barrier_wg() {
  State& s = S();
  unsigned my = s.bar_gen;
  s.bar_arrived++;
  if (s.bar_arrived == s.live) { s.bar_arrived = 0; s.bar_gen++; return; }
  while (s.bar_gen == my) yield();
}

// wave rendezvous: returns true for the LAST arriving lane (which evaluates the collective), after which
// every lane continues once `gen` has moved on.
inline WaveSlot& my_wave() { State& s = S(); return s.waves[s.fibers[s.cur].lin >> 6]; }
inline int my_lane() { State& s = S(); return s.fibers[s.cur].lin & 63; }
inline int wave_width() {
  State& s = S();
  int nthr = (int)(s.block.x * s.block.y * s.block.z);
  int w = s.fibers[s.cur].lin >> 6;
  int rem = nthr - w * 64;
  return rem < 64 ? rem : 64;
}
template <class F>
inline void wave_collective(F&& eval) {
  WaveSlot& w = my_wave();
  unsigned my = w.gen;
  w.arrived++;
  if (w.arrived == wave_width()) { eval(w); w.arrived = 0; w.gen++; return; }
  while (w.gen == my) yield();
}

inline void run_block(const std::function<void()>& body) {
  State& s = S();
  int nthr = (int)(s.block.x * s.block.y * s.block.z);
  const size_t STK = 256 * 1024;
  s.fibers.assign(nthr, Fiber());
  s.waves.assign((nthr + 63) / 64, WaveSlot());
  s.live = nthr; s.bar_arrived = 0; s.bar_gen = 0; s.body = body;
  static std::vector<char*> pool;
  while ((int)pool.size() < nthr) pool.push_back((char*)aligned_alloc(64, STK));
  for (int i = 0; i < nthr; ++i) {
    Fiber& f = s.fibers[i];
    f.lin = i;
    f.tid.x = i % s.block.x; f.tid.y = (i / s.block.x) % s.block.y; f.tid.z = i / (s.block.x * s.block.y);
    f.stack = pool[i];
    // initial frame: 6 callee-saved slots + return address = fsv_emu_entry; keep 16B alignment at entry
    uintptr_t top = ((uintptr_t)(f.stack + STK)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 8);         // so that after `ret` rsp % 16 == 8 like a normal call
    *(--sp) = (void*)&fsv_emu_entry;       // return address
    for (int r = 0; r < 6; ++r) *(--sp) = nullptr;
    f.sp = (void*)sp;
  }
  int remaining = nthr;
  while (remaining > 0) {
    remaining = 0;
    for (int i = 0; i < nthr; ++i) {
      if (s.fibers[i].done) continue;
      s.cur = i;
      fsv_emu_switch(&s.sched_sp, s.fibers[i].sp);
      if (!s.fibers[i].done) remaining++;
    }
  }
  s.cur = -1;
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& body) {
  State& s = S();
  s.grid = grid; s.block = block;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        s.bid = dim3(x, y, z);
        run_block(body);
      }
}

}  // namespace emu

#define threadIdx (emu::S().fibers[emu::S().cur].tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().block)
#define gridDim (emu::S().grid)
static inline void __syncthreads() { emu::barrier_wg(); }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(ptr, order, scope) (*(ptr))
static inline void __threadfence() {}      // one workgroup runs at a time and to completion: every store is already visible

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
// Exact fp32 fma chain over k = 0, 1 (cdna_hip_programming.md section 3 "Numerics").
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
  emu::WaveSlot& w = emu::my_wave();
  int l = emu::my_lane();
  w.a[l] = a; w.b[l] = b;
  for (int r = 0; r < 16; ++r) w.c[l][r] = c[r];
  emu::wave_collective([](emu::WaveSlot& ws) {
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = ws.c[lane][r];
        for (int k = 0; k < 2; ++k) acc = fmaf(ws.a[row + 32 * k], ws.b[col + 32 * k], acc);
        ws.d[lane][r] = acc;
      }
  });
  emu_f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = w.d[l][r];
  return d;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r.
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::WaveSlot& w = emu::my_wave();
  int l = emu::my_lane();
  w.a[l] = a; w.b[l] = b;
  for (int r = 0; r < 4; ++r) w.c[l][r] = c[r];
  emu::wave_collective([](emu::WaveSlot& ws) {
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 4; ++r) {
        int col = lane & 15, row = (lane >> 4) * 4 + r;
        float acc = ws.c[lane][r];
        for (int k = 0; k < 4; ++k) acc = fmaf(ws.a[row + 16 * k], ws.b[col + 16 * k], acc);
        ws.d[lane][r] = acc;
      }
  });
  emu_f32x4 d;
  for (int r = 0; r < 4; ++r) d[r] = w.d[l][r];
  return d;
}

// v_mfma_f32_32x32x16_{f16,bf16} (gfx950): A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e = element 0..7 of the lane's
// operand vector; C/D as the fp32 32x32 form.  Products of two 16-bit floats are exact in fp32; the hardware's internal
// summation order over the 16 k is not documented, the emulator adds them in k order in fp32.
template <class V8>
static inline emu_f32x16 emu_mfma_32x32x16(V8 a, V8 b, emu_f32x16 c) {
  emu::WaveSlot& w = emu::my_wave();
  int l = emu::my_lane();
  for (int e = 0; e < 8; ++e) { w.a8[l][e] = (float)a[e]; w.b8[l][e] = (float)b[e]; }
  for (int r = 0; r < 16; ++r) w.c[l][r] = c[r];
  emu::wave_collective([](emu::WaveSlot& ws) {
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = ws.c[lane][r];
        for (int k = 0; k < 16; ++k) acc += ws.a8[row + 32 * (k >> 3)][k & 7] * ws.b8[col + 32 * (k >> 3)][k & 7];
        ws.d[lane][r] = acc;
      }
  });
  emu_f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = w.d[l][r];
  return d;
}
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x16 c, int, int, int) {
  return emu_mfma_32x32x16(a, b, c);
}
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
  return emu_mfma_32x32x16(a, b, c);
}

template <class T>
static inline T emu_shfl_from(T v, int src_lane) {
  emu::WaveSlot& w = emu::my_wave();
  int l = emu::my_lane();
  static_assert(sizeof(T) <= 8, "shuffle payload");
  long long bits = 0; memcpy(&bits, &v, sizeof(T));
  w.iv[l] = bits; w.src[l] = src_lane;
  emu::wave_collective([](emu::WaveSlot& ws) {
    long long tmp[64];
    for (int i = 0; i < 64; ++i) { int s = ws.src[i]; tmp[i] = (s >= 0 && s < 64) ? ws.iv[s] : ws.iv[i]; }
    for (int i = 0; i < 64; ++i) { double dd; memcpy(&dd, &tmp[i], 8); ws.dv[i] = dd; }
  });
  T out; memcpy(&out, &w.dv[l], sizeof(T));
  return out;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_shfl_from(v, emu::my_lane() ^ mask); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { (void)width; int s = emu::my_lane() + d; return emu_shfl_from(v, s < 64 ? s : emu::my_lane()); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return emu_shfl_from(v, src); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float fminf_(float a, float b) { return a < b ? a : b; }

// number of kernel launches since the library was loaded (the emulated library is one translation unit): lets the CPU tests
// watch the launch count of a training step, which is what bounds the eager step on the real stack
inline long long& fsv_emu_launches() { static long long n = 0; return n; }
extern "C" __attribute__((used, visibility("default"))) long long fsv_emu_launch_count() { return fsv_emu_launches(); }
#include <map>
#include <string>
inline std::map<std::string, long long>& fsv_emu_by_kernel() { static std::map<std::string, long long> m; return m; }
// "name count\n" lines of the launches since the last reset (reset != 0 clears the table after writing it)
extern "C" __attribute__((used, visibility("default"))) int fsv_emu_launch_report(char* buf, int cap, int reset) {
  std::string out;
  for (auto& kv : fsv_emu_by_kernel()) out += kv.first + " " + std::to_string(kv.second) + "\n";
  if (reset) fsv_emu_by_kernel().clear();
  int n = (int)out.size() < cap - 1 ? (int)out.size() : cap - 1;
  if (buf && cap > 0) { memcpy(buf, out.data(), n); buf[n] = 0; }
  return (int)out.size();
}

#define FSV_LAUNCH(kernel, grid, block, stream, ...)                         \
  do { (void)(stream); ++fsv_emu_launches(); ++fsv_emu_by_kernel()[#kernel];                                         \
       emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); }); } while (0)
