// Shared helpers for the gfx950 kernels behind include/inr.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/inr.h"

namespace inr {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(inr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that just happened; never exits the process.
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return INR_ELAUNCH;
  }
  return INR_OK;
}

#define INR_REQUIRE(cond, msg)            \
  do {                                    \
    if (!(cond)) {                        \
      ::inr::set_error("%s: %s", __func__, msg); \
      return INR_EINVAL;                  \
    }                                     \
  } while (0)

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return expand_bits10(x) | (expand_bits10(y) << 1) | (expand_bits10(z) << 2);
}
__device__ __forceinline__ uint32_t compact_bits10(uint32_t x) {
  x &= 0x49249249u;
  x = (x | (x >> 2)) & 0xC30C30C3u;
  x = (x | (x >> 4)) & 0x0F00F00Fu;
  x = (x | (x >> 8)) & 0xFF0000FFu;
  x = (x | (x >> 16)) & 0x0000FFFFu;
  return x;
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// Wave64 inclusive prefix sum (all lanes must participate).
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// Compute units of the current device, queried once per device: hipGetDeviceProperties fills a multi-KB struct
// and costs tens of microseconds - too much for a call that precedes every launch of a 50 us kernel.
inline int cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

// Overlap placement (inr_set_overlap_placement, raymarch.hip): minimum dynamic LDS of the eval field launch and extra
// dynamic LDS of the march launches; both 0 unless a caller overlaps the two on separate streams.
extern int g_field_lds_min;
extern int g_march_lds_pad;

}  // namespace inr
