// Microbenchmark (round 5): does it pay to give every XCD's L2 its OWN hashed level?
// A hashed level is 2^19 rows x 8 B = 4 MiB - exactly one XCD's L2; a bound-4 table has twelve of them, and on its
// fine levels no two samples share a cell, so every (sample, level) costs four random 128-byte lines (x-neighbour rows
// share a line).  The fused field kernel lets every workgroup touch all 16 levels: 48 MB of table behind 4 MB of L2.
//   mode 0 "mixed":     every lane gathers from all NL levels in turn (the fused kernel's access pattern)
//   mode 1 "xcd-owned": workgroup b gathers from level (XCC_ID % NL) only (an XCD's L2 holds one level)
// Same number of loads either way; lanes 2i, 2i+1 read the two rows of one aligned 16-byte pair.
// usage: level_xcd_bench [log2_rows_per_level=19]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

template <int MODE>
__global__ void __launch_bounds__(512) k_gather(const float2* __restrict__ table, uint32_t rows_mask, uint32_t level_rows,
                                                int n_levels, int rounds, float* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t own = xcc_id() % (uint32_t)n_levels;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const uint32_t salt = (uint32_t)(r * 32 + k) * 0x9E3779B9u;
      uint32_t row = ((hash32((tid >> 1) ^ salt) & ~1u) | (lane & 1u)) & rows_mask;
      if (MODE == 2) {                     // the real x-neighbour pair of a hashed level: rows c ^ h and (c + 1) ^ h, c random
        const uint32_t r0 = hash32((tid >> 1) ^ salt), c = hash32(r0 ^ 0x5bd1e995u);
        row = ((c + (lane & 1u)) ^ r0) & rows_mask;
      }
      const uint32_t level = MODE == 0 ? (uint32_t)((k >> 2) % n_levels) : own;     // four loads per (sample, level)
      v[k] = table[(size_t)level * level_rows + row];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += v[k].x + v[k].y;
  }
  if (acc == 12345.678f) out[tid] = acc;
}

template <int MODE>
static int run(const float2* table, uint32_t rows, int n_levels, float* out, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 200, blocks = 512;
  k_gather<MODE><<<blocks, 512>>>(table, rows - 1, rows, n_levels, 8, out);
  CK(hipEventRecord(e0));
  k_gather<MODE><<<blocks, 512>>>(table, rows - 1, rows, n_levels, rounds, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double loads = (double)blocks * 512 * rounds * 32;
  printf("levels %2d x 2^%2d rows  mode %d  %8.3f ms  %7.1f G lane-loads/s  %6.1f G lines/s  %5.2f TB/s of lines   %s\n", n_levels,
         31 - __builtin_clz(rows), MODE, ms, loads / ms / 1e6, loads / 2 / ms / 1e6, loads / 2 * 128 / ms / 1e9, what);
  return 0;
}

int main(int argc, char** argv) {
  const uint32_t lg = argc > 1 ? (uint32_t)atoi(argv[1]) : 19u;
  const uint32_t rows = 1u << lg;
  float2* table; float* out;
  CK(hipMalloc(&table, (size_t)rows * 8 * 16)); CK(hipMemset(table, 0, (size_t)rows * 8 * 16));
  CK(hipMalloc(&out, 512 * 512 * 4));
  for (int nl : {1, 2, 4, 8, 12, 16}) {
    if (run<0>(table, rows, nl, out, "mixed: every lane walks all levels")) return 1;
    if (nl <= 8 || nl == 16) if (run<1>(table, rows, nl, out, "xcd-owned: level = XCC_ID % levels")) return 1;
    if (nl == 8) if (run<2>(table, rows, nl, out, "xcd-owned, pair = rows c^h, (c+1)^h (same line 15/16, same 16 bytes 1/2)")) return 1;
  }
  return 0;
}
