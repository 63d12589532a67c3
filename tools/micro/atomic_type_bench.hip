// Microbenchmark (round 6, round-5 verdict item 6): does ANY atomic type execute in the XCD's own L2 on MI355X?
// The table-gradient scatter (k_grid_bwd) is bound by the memory-side atomic unit: every fp32 atomic is forwarded
// (TCC_EA0_WRREQ_ATOMIC_DRAM == TCC_ATOMIC) at ~21 G requests/s whatever the footprint or the scope
// (atomic_scope_bench.hip, atomic_width_bench.hip).  Here the same random scatter with other operand types - u32, u64,
// packed bf16, packed f16, f64 - over the whole table and over 4 MiB slices that only one XCD ever touches (a hashed
// level is exactly that size).  Rates from events; run under `rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum
// TCC_EA0_WRREQ_sum` (tools/pmc_atomic_types.sh) to see, per kernel, how many of the L2's atomics went on to the memory
// side.  If one type stayed in the L2 its rate on the XCD-owned slice would rise above the ~21 G/s of the unit, and a
// fixed-point accumulation of the table gradient would be worth pricing; if none does the scatter is closed for good.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

enum { T_F32 = 0, T_U32 = 1, T_U64 = 2, T_PKBF16 = 3, T_PKF16 = 4, T_F64 = 5, T_U32_RET = 6, T_F32_WG = 7, T_U32_WG = 8 };

// one atomic per operation on the 8-byte row r (the scatter's row: two fp32 features)
template <int TYPE>
__device__ __forceinline__ void op(char* row, uint32_t& sink) {
  if (TYPE == T_F32) __hip_atomic_fetch_add(reinterpret_cast<float*>(row), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TYPE == T_F32_WG) __hip_atomic_fetch_add(reinterpret_cast<float*>(row), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (TYPE == T_U32) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(row), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TYPE == T_U32_WG) __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(row), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (TYPE == T_U32_RET) sink += __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(row), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TYPE == T_U64) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(row), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TYPE == T_F64) __hip_atomic_fetch_add(reinterpret_cast<double*>(row), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (TYPE == T_PKBF16) {                     // two bf16 ones: 0x3F80 each
    const uint32_t v = 0x3F803F80u;
    asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(row), "v"(v) : "memory");
  }
  if (TYPE == T_PKF16) {                      // two f16 ones: 0x3C00 each
    const uint32_t v = 0x3C003C00u;
    asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(row), "v"(v) : "memory");
  }
}

// SLICED 0: any row of the table; 1: rows of slice xcc_id() only (slice_rows each)
template <int TYPE, int SLICED>
__global__ void k_atomic(char* table, uint32_t rows, uint32_t slice_rows, uint64_t n_ops, uint32_t seed, uint32_t* sink_out) {
  const uint32_t x = xcc_id();
  uint32_t sink = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)i ^ seed);
    const size_t r = SLICED ? (size_t)x * slice_rows + h % slice_rows : h % rows;
    op<TYPE>(table + 8 * r, sink);
  }
  if (TYPE == T_U32_RET && sink == 0xFFFFFFFFu) *sink_out = sink;
}

// The scatter's REAL request shape (k_grid_bwd: 4 lanes per sample = (x side) x (feature) -> the two 8-byte rows of an
// x-adjacent corner pair = 16 contiguous bytes per 4 lanes, one memory-side request): each group of 4 lanes adds to the 4
// dwords of one random 16-byte unit of a table of the product's size (6 119 864 rows = 49 MB), f32 against u32.
template <int TYPE>
__global__ void k_atomic_quad(char* table, uint32_t units, uint64_t n_ops, uint32_t seed) {
  uint32_t sink = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)(i >> 2) ^ seed);
    char* p = table + 16 * (size_t)(h % units) + 4 * (i & 3);
    op<TYPE>(p, sink);
  }
}

// The same with 8-byte operands (what a 64-bit fixed-point sum would need): 4 lanes x 8 B = one 32-byte request per group,
// both features of a row pair widened to 64 bits (the table gradient would then take 98 MB).
template <int TYPE>
__global__ void k_atomic_quad8(char* table, uint32_t units, uint64_t n_ops, uint32_t seed) {
  uint32_t sink = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32((uint32_t)(i >> 2) ^ seed);
    char* p = table + 32 * (size_t)(h % units) + 8 * (i & 3);
    op<TYPE>(p, sink);
  }
}

template <int TYPE>
static int run_quad8(const char* name, char* table, uint64_t n_ops) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t units = 6119864u / 2;                       // 32-byte units: a table of the product's row count, 8 B per feature
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(table, 0, (size_t)units * 32));
    CK(hipEventRecord(e0));
    k_atomic_quad8<TYPE><<<2048, 256>>>(table, units, n_ops, rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-22s 32-byte requests (4 lanes x 8 B) into a 98 MB table: %.3f ms  %6.2f G lane-atomics/s = %5.2f G requests/s\n", name, best,
         n_ops / best / 1e6, n_ops / 4.0 / best / 1e6);
  return 0;
}

template <int TYPE>
static int run_quad(const char* name, char* table, uint64_t n_ops) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t units = 6119864u / 2;                       // 16-byte units of the product's table
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(table, 0, (size_t)units * 16));
    CK(hipEventRecord(e0));
    k_atomic_quad<TYPE><<<2048, 256>>>(table, units, n_ops, rep);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-22s 16-byte requests (4 lanes x 4 B) into a 49 MB table: %.3f ms  %6.2f G lane-atomics/s = %5.2f G requests/s\n", name, best,
         n_ops / best / 1e6, n_ops / 4.0 / best / 1e6);
  return 0;
}

template <int TYPE>
static int run(const char* name, char* table, uint32_t rows, uint64_t n_ops, uint32_t* sink, char* host) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t slice_rows = 512u << 10;                   // 4 MiB per XCD: one hashed level
  for (int sliced = 0; sliced < 2; ++sliced) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(table, 0, (size_t)rows * 8));
      CK(hipEventRecord(e0));
      if (sliced) k_atomic<TYPE, 1><<<2048, 256>>>(table, rows, slice_rows, n_ops, rep, sink);
      else k_atomic<TYPE, 0><<<2048, 256>>>(table, rows, slice_rows, n_ops, rep, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    // checksum: every operation added "one" to the first word(s) of its row
    CK(hipMemcpy(host, table, (size_t)rows * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (size_t r = 0; r < rows; ++r) {
      const char* p = host + 8 * r;
      if (TYPE == T_F32 || TYPE == T_F32_WG) { float v; memcpy(&v, p, 4); s += v; }
      else if (TYPE == T_U32 || TYPE == T_U32_RET || TYPE == T_U32_WG) { uint32_t v; memcpy(&v, p, 4); s += v; }
      else if (TYPE == T_U64) { unsigned long long v; memcpy(&v, p, 8); s += (double)v; }
      else if (TYPE == T_F64) { double v; memcpy(&v, p, 8); s += v; }
      else if (TYPE == T_PKBF16) { uint16_t v; memcpy(&v, p, 2); uint32_t w = (uint32_t)v << 16; float f; memcpy(&f, &w, 4); s += f; }
      else if (TYPE == T_PKF16) { _Float16 v; memcpy(&v, p, 2); s += (float)v; }
    }
    // bf16 / f16 saturate in precision once a row holds more than 256 / 2048 ones: only exact for sparse tables
    const bool exact = !(TYPE == T_PKBF16 || TYPE == T_PKF16);
    printf("%-22s %s: %.3f ms  %6.2f G atomics/s   checksum %s\n", name,
           sliced ? "4 MiB slice per XCD (rows owned by one XCD)" : "whole table (1 GiB, random rows)          ", best,
           n_ops / best / 1e6, !exact ? "(low-precision adds: not exact)" : s == (double)n_ops ? "ok" : "WRONG");
  }
  return 0;
}

int main() {
  const uint64_t n_ops = 25000000ull;            // ~191k samples x 16 levels x 8 corners
  const uint32_t rows = 128u << 20;              // 1 GiB of 8-byte rows
  char* table; CK(hipMalloc(&table, (size_t)rows * 8));
  uint32_t* sink; CK(hipMalloc(&sink, 4));
  char* host = (char*)malloc((size_t)rows * 8);
  if (!host) return 1;
  if (run<T_F32>("f32 add (agent)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_F32_WG>("f32 add (workgroup)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_U32>("u32 add (agent)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_U32_WG>("u32 add (workgroup)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_U32_RET>("u32 add, returning", table, rows, n_ops, sink, host)) return 1;
  if (run<T_U64>("u64 add (agent)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_F64>("f64 add (agent)", table, rows, n_ops, sink, host)) return 1;
  if (run<T_PKBF16>("pk_add_bf16", table, rows, n_ops, sink, host)) return 1;
  if (run<T_PKF16>("pk_add_f16", table, rows, n_ops, sink, host)) return 1;
  const uint64_t n_quad = 64000000ull;           // 16 M requests: ~440 k samples x 38 requests, the trained scene's step
  if (run_quad<T_F32>("f32 add", table, n_quad)) return 1;
  if (run_quad<T_U32>("u32 add", table, n_quad)) return 1;
  if (run_quad<T_F32>("f32 add (again)", table, n_quad)) return 1;
  if (run_quad<T_U32>("u32 add (again)", table, n_quad)) return 1;
  if (run_quad8<T_U64>("u64 add", table, n_quad)) return 1;
  if (run_quad8<T_F64>("f64 add", table, n_quad)) return 1;
  return 0;
}
