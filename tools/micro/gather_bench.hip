// Microbenchmark: how the vector memory path of an MI355X CU prices 8-byte gathers as a function of WHICH lanes of a
// wave share a 128-byte line.  One 512-thread workgroup per CU (the field kernel's shape), every lane issues 32
// independent buffer-free global loads per round from a table that fits the L2 (or the L1), rows pseudo-random.
//   mode 0: every lane its own random row (distinct lines)
//   mode 1: lanes 2i, 2i+1 read adjacent rows of one aligned 16-byte pair (an x-neighbour pair of a hashed level)
//   mode 2: lanes 4i..4i+3 read one aligned 32-byte quad
//   mode 3: lanes 16i..16i+15 read one 128-byte line
//   mode 4: like mode 1 but the partners are lanes l and l^32 (same line, far-apart lanes)
//   mode 5: like mode 1 with one dwordx4 per PAIR issued by the even lane only (odd lanes idle)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(512) k_gather(const float2* __restrict__ table, uint32_t rows_mask, int rounds, float* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63, wave = tid >> 6;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    float2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const uint32_t salt = (uint32_t)(r * 32 + k) * 0x9E3779B9u;
      uint32_t row;
      if (MODE == 0) row = hash32(tid ^ salt);
      else if (MODE == 1 || MODE == 5) row = (hash32((tid >> 1) ^ salt) & ~1u) | (lane & 1u);
      else if (MODE == 2) row = (hash32((tid >> 2) ^ salt) & ~3u) | (lane & 3u);
      else if (MODE == 3) row = (hash32((tid >> 4) ^ salt) & ~15u) | (lane & 15u);
      else row = (hash32((wave * 32 + (lane & 31)) ^ salt) & ~1u) | (lane >> 5);
      row &= rows_mask;
      if (MODE == 5) {
        if ((lane & 1) == 0) {
          const float4 t = *reinterpret_cast<const float4*>(table + (row & ~1u));
          v[k] = make_float2(t.x + t.z, t.y + t.w);
        } else v[k] = make_float2(0.f, 0.f);
      } else v[k] = table[row];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += v[k].x + v[k].y;
  }
  if (acc == 12345.678f) out[tid] = acc;
}

template <int MODE>
static int run(const float2* table, uint32_t rows, float* out, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = 200, blocks = 256;
  k_gather<MODE><<<blocks, 512>>>(table, rows - 1, 4, out);
  CK(hipEventRecord(e0));
  k_gather<MODE><<<blocks, 512>>>(table, rows - 1, rounds, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double loads = (double)blocks * 512 * rounds * 32;
  printf("rows 2^%2d  mode %d  %7.3f ms  %7.1f G lane-loads/s  %6.2f lane-loads/clk/CU (2.4 GHz)   %s\n", 31 - __builtin_clz(rows),
         MODE, ms, loads / ms / 1e6, loads / ms / 1e6 / 256 / 2.4, what);
  return 0;
}

int main() {
  float2* table; float* out;
  const uint32_t max_rows = 1u << 23;
  CK(hipMalloc(&table, (size_t)max_rows * 8)); CK(hipMemset(table, 0, (size_t)max_rows * 8));
  CK(hipMalloc(&out, 256 * 512 * 4));
  for (uint32_t lg : {11u, 15u, 19u, 23u}) {            // 16 KB (L1), 256 KB, 4 MB (one hashed level), 64 MB (whole table)
    const uint32_t rows = 1u << lg;
    if (run<0>(table, rows, out, "distinct lines")) return 1;
    if (run<1>(table, rows, out, "adjacent lanes share a 16-byte pair")) return 1;
    if (run<2>(table, rows, out, "quads share 32 bytes")) return 1;
    if (run<3>(table, rows, out, "16 lanes share a line")) return 1;
    if (run<4>(table, rows, out, "lanes l, l^32 share a pair")) return 1;
    if (run<5>(table, rows, out, "one dwordx4 per pair (even lanes)")) return 1;
  }
  return 0;
}
