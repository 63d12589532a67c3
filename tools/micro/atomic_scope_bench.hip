// Microbenchmark: fp32 atomic-add scatter on MI355X when every address is only ever touched from ONE XCD.
// Agent-scope atomics execute memory-side (the eight L2s are not coherent with each other); workgroup-scope atomics
// execute in the XCD's own L2.  If rows are partitioned by XCD the narrower scope is sufficient for correctness on this
// hardware (kernel boundaries write the L2s back).  Question: how much faster, and up to which slice size?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

// SCOPE 0: agent, 1: workgroup.  SLICED 0: any row of the table; 1: rows of slice xcc_id() only (slice_rows each).
template <int SCOPE, int SLICED>
__global__ void k_scatter(float* table, uint32_t rows, uint32_t slice_rows, uint64_t n_ops, uint32_t seed) {
  const uint32_t x = xcc_id();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)i ^ seed);
    const size_t r = SLICED ? (size_t)x * slice_rows + h % slice_rows : h % rows;
    float* p = table + 2 * r;
    if (SCOPE == 0) {
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(p + 1, 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(p + 1, 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

__global__ void k_xcc_hist(uint32_t* hist, uint32_t* mism) {
  if (threadIdx.x == 0) {
    const uint32_t x = xcc_id();
    atomicAdd(hist + x, 1u);
    if (x != (blockIdx.x & 7u)) atomicAdd(mism, 1u);
  }
}

int main() {
  const uint64_t n_ops = 12500000ull;            // ~191k samples x 16 levels x 4 transactions
  const uint32_t max_slice = 2u << 20;           // rows per XCD slice at most (16 MB)
  const uint32_t rows = 8 * max_slice;
  float* table; CK(hipMalloc(&table, (size_t)rows * 2 * 4));
  uint32_t* hist; CK(hipMalloc(&hist, 64)); CK(hipMemset(hist, 0, 64));
  k_xcc_hist<<<2048, 64>>>(hist, hist + 8); CK(hipDeviceSynchronize());
  uint32_t hh[9]; CK(hipMemcpy(hh, hist, 36, hipMemcpyDeviceToHost));
  printf("workgroups per XCC:"); for (int i = 0; i < 8; ++i) printf(" %u", hh[i]); printf("  (blockIdx&7 != xcc: %u of 2048)\n", hh[8]);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float* h = (float*)malloc((size_t)rows * 2 * 4);
  const uint32_t slices[] = {1u << 14, 1u << 16, 1u << 17, 1u << 18, 1u << 19, 1u << 20, 2u << 20};
  for (int mode = 0; mode < 3; ++mode) {
    for (uint32_t sr : slices) {
      if (mode == 0 && sr != slices[0]) continue;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(table, 0, (size_t)rows * 2 * 4));
        CK(hipEventRecord(e0));
        if (mode == 0) k_scatter<0, 0><<<2048, 256>>>(table, rows, sr, n_ops, rep);
        if (mode == 1) k_scatter<0, 1><<<2048, 256>>>(table, rows, sr, n_ops, rep);
        if (mode == 2) k_scatter<1, 1><<<2048, 256>>>(table, rows, sr, n_ops, rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(h, table, (size_t)rows * 2 * 4, hipMemcpyDeviceToHost));
      double s = 0; for (size_t i = 0; i < (size_t)rows * 2; ++i) s += h[i];
      printf("%s scope, %s, slice %8u rows (%6.2f MB per XCD): %.3f ms  %.1f G atomics/s  checksum %s\n",
             mode == 2 ? "workgroup" : "agent    ", mode == 0 ? "whole table " : "rows by XCD ", sr, sr * 8.0 / 1048576.0,
             best, 2.0 * n_ops / best / 1e6, s == 1.5 * n_ops ? "ok" : "WRONG");
      if (s != 1.5 * n_ops) printf("    got %.1f expected %.1f\n", s, 1.5 * n_ops);
    }
  }
  return 0;
}
