// Shared device/host helpers for the fsv2v HIP kernels (gfx950 / CDNA4 only).
//
// Every kernel in this directory is written for 64-wide wavefronts and the gfx950 MFMA lane layouts
// (see /opt/skills/guides/cdna_hip_programming.md section 3).  The only other compilation mode is the CPU
// SIMT emulator used by the `not gpu` tests (tests/emu/hip_emu.h), selected with -DFSV_EMU.
#pragma once
#ifdef FSV_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
// hipGetLastError() is per-thread state shared with everything else in the process (PyTorch's pinned-memory allocator
// leaves hipErrorNotReady there after polling an event): clear it before a launch and record only what the launch
// itself reports, so that fsv_check_launch() speaks for this library's launches alone.
static thread_local int fsv_launch_status = 0;
#define FSV_LAUNCH(kernel, grid, block, stream, ...)                                  \
  do {                                                                                \
    (void)hipGetLastError();                                                          \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (stream), __VA_ARGS__);            \
    if (hipGetLastError() != hipSuccess) fsv_launch_status = -3;                      \
  } while (0)
#endif
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- status codes of the C ABI (include/fsv2v.h) -----------------------------------------------------
#define FSV_OK 0
#define FSV_ERR_BAD_ARG (-1)
#define FSV_ERR_UNSUPPORTED (-2)
#define FSV_ERR_LAUNCH (-3)

// activation codes used by fused epilogues
#define FSV_ACT_NONE 0
#define FSV_ACT_LRELU 1   // leaky_relu(x, 0.2)  (reference models/networks/architecture.py:15-17)
#define FSV_ACT_TANH 2
#define FSV_ACT_SIGMOID 3
#define FSV_ACT_RELU 4      // VGG19 feature stack (models/networks/vgg.py)
#define FSV_ACT_LRELU01 5   // leaky_relu(x, 0.1): FlowNet2 teacher (flownet2_pytorch/networks/submodules.py:16,39)
// gather-GEMM epilogue only (V4 kernel): out = v * leaky_relu'(aux), aux = the `res` operand (the OUTPUT of the layer whose
// pre-activation gradient this launch produces) - the act-backward pass of a Linear + LeakyReLU chain folded into the data
// gradient that feeds it.  Same arithmetic as fsv_act_bwd_kernel: aux > 0 ? v : 0.2 * v.
#define FSV_ACT_DLRELU 6

#ifdef FSV_EMU
static inline int fsv_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? FSV_OK : FSV_ERR_LAUNCH;
}
#else
static inline int fsv_check_launch() {        // status of the launches since the previous check (this thread, this file)
  const int s = fsv_launch_status;
  fsv_launch_status = FSV_OK;
  return s;
}
#endif

__device__ __forceinline__ float fsv_act(float v, int act) {
  if (act == FSV_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  if (act == FSV_ACT_TANH) return tanhf(v);
  if (act == FSV_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == FSV_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == FSV_ACT_LRELU01) return v > 0.f ? v : 0.1f * v;
  return v;
}

static inline int fsv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
