// Microbenchmark: fp32 atomic-add scatter throughput on MI355X, shared table vs one private copy per XCD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

// mode 0: shared table; 1: private copy per XCD (by XCC_ID); 2: shared, but addresses sorted (adjacent lanes adjacent rows)
template <int MODE>
__global__ void k_scatter(float* table, uint32_t rows, uint64_t n_ops, uint32_t seed) {
  float* t = table;
  if (MODE == 1) t += (size_t)xcc_id() * rows * 2;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t r = MODE == 2 ? (uint32_t)(i % rows) : hash32((uint32_t)i ^ seed) % rows;
    atomicAdd(t + 2 * (size_t)r, 1.0f);
    atomicAdd(t + 2 * (size_t)r + 1, 0.5f);
  }
}
__global__ void k_reduce8(const float4* copies, float4* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = copies[i];
    for (int c = 1; c < 8; ++c) { float4 b = copies[i + c * n4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    out[i] = a;
  }
}
int main() {
  const uint32_t rows = 6119864; const uint64_t n_ops = 26800000ull;   // 209k samples x 128 rows
  float *table, *out; CK(hipMalloc(&table, (size_t)rows * 2 * 4 * 8)); CK(hipMalloc(&out, (size_t)rows * 2 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(table, 0, (size_t)rows * 2 * 4 * 8));
      CK(hipEventRecord(e0));
      if (mode == 0) k_scatter<0><<<2048, 256>>>(table, rows, n_ops, rep);
      if (mode == 1) k_scatter<1><<<2048, 256>>>(table, rows, n_ops, rep);
      if (mode == 2) k_scatter<2><<<2048, 256>>>(table, rows, n_ops, rep);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float ms2 = 0;
      if (mode == 1) { CK(hipEventRecord(e0)); k_reduce8<<<2048, 256>>>((const float4*)table, (float4*)out, (size_t)rows * 2 / 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms2, e0, e1)); }
      printf("mode %d rep %d: scatter %.3f ms (%.1f G atomics/s)  reduce %.3f ms\n", mode, rep, ms, 2.0 * n_ops / ms / 1e6, ms2);
    }
  }
  // correctness of mode 1: total must equal n_ops*1.5
  CK(hipMemset(table, 0, (size_t)rows * 2 * 4 * 8));
  k_scatter<1><<<2048, 256>>>(table, rows, n_ops, 7);
  k_reduce8<<<2048, 256>>>((const float4*)table, (float4*)out, (size_t)rows * 2 / 4);
  CK(hipDeviceSynchronize());
  float* h = (float*)malloc((size_t)rows * 2 * 4); CK(hipMemcpy(h, out, (size_t)rows * 2 * 4, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < (size_t)rows * 2; ++i) s += h[i];
  printf("mode 1 checksum %.1f expected %.1f\n", s, 1.5 * n_ops);
  return 0;
}
