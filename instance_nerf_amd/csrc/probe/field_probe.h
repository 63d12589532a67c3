// Ablation hooks of the fused field kernel (k_nerf_fwd in ../field_fused.hip) for PROFILING builds only.
// Every variant here produces WRONG or incomplete results; this file is not among the sources instance_nerf_amd/build.py
// compiles or hashes - only tools/build_probe.py adds it, by defining INR_PROBE_BUILD:
//   -DINR_PROBE_MODE=2     no MLP: the gathered features are summed into sigma (what does the gather alone cost?)
//   -DINR_PROBE_MODE=3     per-workgroup start / end time stamps in the geo buffer (uint64 [gridDim.x][2]; geo is not written)
//   -DINR_PROBE_STATIC=1   the static tile deal again (no per-XCD stealing cursors)
//   -DINR_PROBE_SLOW_XCD=1 ~3 us of sleep per tile on XCD 3: what does the dynamic schedule do with a slow XCD?
// Measurements taken with them: profiles/r02_NOTES.txt .. r05_NOTES.txt.
#pragma once
#ifndef INR_PROBE_BUILD
#error "csrc/probe/field_probe.h is for tools/build_probe.py builds only (-DINR_PROBE_BUILD)"
#endif
#ifndef INR_PROBE_MODE
#define INR_PROBE_MODE 0
#endif

#if INR_PROBE_MODE == 3
#define INR_PROBE_PROLOGUE() const unsigned long long probe_t0 = wall_clock64()
#define INR_PROBE_GEO_IS_OUTPUT 0
#define INR_PROBE_EPILOGUE()                                                            \
  do {                                                                                  \
    if (geo && (threadIdx.x & 63) == 0) {                                               \
      unsigned long long* dbg = reinterpret_cast<unsigned long long*>(geo);             \
      atomicMax(&dbg[2 * blockIdx.x + 1], (unsigned long long)wall_clock64());          \
      if (threadIdx.x == 0) dbg[2 * blockIdx.x] = probe_t0;                             \
    }                                                                                   \
  } while (0)
#else
#define INR_PROBE_PROLOGUE()
#define INR_PROBE_GEO_IS_OUTPUT 1
#define INR_PROBE_EPILOGUE()
#endif

#if defined(INR_PROBE_STATIC) && INR_PROBE_STATIC
#define INR_PROBE_SCHEDULE(steal) steal = nullptr
#else
#define INR_PROBE_SCHEDULE(steal)
#endif

#if defined(INR_PROBE_SLOW_XCD) && INR_PROBE_SLOW_XCD
#define INR_PROBE_TILE()                                              \
  do {                                                                \
    if ((blockIdx.x & 7) == 3) __builtin_amdgcn_s_sleep(127);         \
  } while (0)
#else
#define INR_PROBE_TILE()
#endif

#if INR_PROBE_MODE == 2
#define INR_PROBE_AFTER_GATHER()                                                                                      \
  {                                                                                                                   \
    float acc = 0.f;                                                                                                  \
    for (int t = 0; t < 4; ++t) acc += enc[0][t] + enc[1][t];                                                         \
    acc += __shfl_xor(acc, 16);                                                                                       \
    acc += __shfl_xor(acc, 32);                                                                                       \
    if (valid && q == 0) {                                                                                            \
      sigma[m] = acc;                                                                                                 \
      if (rgb) { rgb[m * 3] = acc; rgb[m * 3 + 1] = me.d0; rgb[m * 3 + 2] = me.d1; }                                  \
    }                                                                                                                 \
    continue;                                                                                                         \
  }
#else
#define INR_PROBE_AFTER_GATHER()
#endif
