// Device-side time stamps for bench.py's `roofline` object: a one-work-item kernel that writes the GPU's constant-rate wall clock
// (wall_clock64, hipDeviceAttributeWallClockRate) into a slot.  A pair of them around a launch is a graph-capturable bracket: inside
// the REPLAYED iteration - where HIP events cannot be recorded - the pairs time the dominant kernel's launches where they sit, with the
// real neighbours, clocks and caches of the replay (the eager instrumented pass reads 4 - 7 % long against rocprofv3's in-graph
// durations).  A pair brackets [end of whatever ran before the first stamp, end of the launch + the stamp after it]: the per-pair overhead is
// measured by the caller with pairs around nothing (same stream, same graph) and subtracted.  Measurement infrastructure only: the
// product path never launches it.
#include "fsv_common.h"

#ifdef FSV_EMU
static inline long long fsv_emu_clock() { static long long t = 0; return t += 100; }
#endif

__global__ void fsv_stamp_kernel(unsigned long long* slot) {
#ifdef FSV_EMU
  if (threadIdx.x == 0) *slot = (unsigned long long)fsv_emu_clock();
#else
  if (threadIdx.x == 0) *slot = (unsigned long long)wall_clock64();
#endif
}

extern "C" {

// writes the device's wall clock into *slot when the stream reaches this point
int fsv_stamp(unsigned long long* slot, hipStream_t stream) {
  if (!slot) return FSV_ERR_BAD_ARG;
  FSV_LAUNCH(fsv_stamp_kernel, dim3(1), dim3(64), stream, slot);
  return fsv_check_launch();
}

// ticks per millisecond of the clock fsv_stamp reads (kHz), 0 when the runtime does not report it
int fsv_stamp_rate_khz(void) {
#ifdef FSV_EMU
  return 100000;
#else
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
  return khz;
#endif
}

}  // extern "C"
