// Shared helpers for the gfx950 kernels of libtgsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "../../include/tgsr_hip.h"

namespace tgsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;     // CDNA wavefront
constexpr int kConvCK = 4;    // input channels per LDS stage (== the packed-weight chunk)

// Records a launch failure for tgsr_last_error(); returns the ABI status.
int note_launch(hipError_t e, const char* what);
// Records a non-HIP failure (e.g. an RCCL status) for tgsr_last_error(); returns TGSR_ELAUNCH.
int note_error(const char* what, const char* detail, int code);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Row of a 32x32 MFMA accumulator register: D[row][col = lane & 31], lane half h = lane >> 5.
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tile ids so that
// neighbouring tiles (shared halo rows, same weights) hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Experiment knob: TGSR_WGRAD_SPLIT_PCT scales how many partial slabs the split weight-gradient kernels produce
// (100 = the launchers' own choice).
inline int wgrad_split_pct() {
  static int pct = [] {
    const char* e = getenv("TGSR_WGRAD_SPLIT_PCT");
    const int v = e ? atoi(e) : 100;
    return v < 1 ? 100 : v;
  }();
  return pct;
}

}  // namespace tgsr
