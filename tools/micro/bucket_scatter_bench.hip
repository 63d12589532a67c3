// Microbenchmark (round 3): would a BUCKETED table-gradient scatter beat the atomic one?
// Atomic scatter today: ~7.1 M memory-side atomic requests per step at ~18-21 G/s = 370-400 us (profiles/r03_NOTES.txt).
// Alternative without global atomics: route every (row, value) record to the owner of its table slice, accumulate the
// slice in LDS, write it out with plain stores:
//   K1  count records per (workgroup, bin)         bin = (level, row >> 14): 32 slices of 16384 rows (128 KB) per level
//   K2  exclusive scan per bin over the workgroups
//   K3  write the records to their positions       (12 B each: row, 2 floats)
//   K4  one workgroup per bin: ds_add_f32 into a 128 KB LDS slice, then 16-byte stores of the slice
// Synthetic input with the statistics of the hashed levels: L levels x M samples x 8 uniformly random rows of 2^19.
// build: hipcc --offload-arch=gfx950 -O3 -o bucket_scatter_bench bucket_scatter_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

constexpr int kRowsLog2 = 19, kSliceLog2 = 14, kSlices = 1 << (kRowsLog2 - kSliceLog2);   // 32 slices per level
constexpr int kSamplesPerWG = 256;      // K1 / K3 workgroup: 256 samples x 8 corners = 2048 records of one level
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t row_of(int level, int64_t m, int k) {
  return hash32((uint32_t)(m * 8 + k) * 2654435761u ^ (uint32_t)level * 805459861u) & ((1u << kRowsLog2) - 1u);
}

// K1: counts[level][slice][wg]
__global__ void __launch_bounds__(kThreads) k_count(int64_t M, int n_wg, int32_t* __restrict__ counts) {
  __shared__ int hist[kSlices];
  const int level = blockIdx.y, wg = blockIdx.x;
  if (threadIdx.x < kSlices) hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t m = (int64_t)wg * kSamplesPerWG + threadIdx.x;
  if (m < M)
    for (int k = 0; k < 8; ++k) atomicAdd(&hist[row_of(level, m, k) >> kSliceLog2], 1);
  __syncthreads();
  if (threadIdx.x < kSlices) counts[((int64_t)level * kSlices + threadIdx.x) * n_wg + wg] = hist[threadIdx.x];
}

// K2: exclusive scan of counts[bin][0..n_wg) per bin (one wave per bin), bin totals to totals[bin]
__global__ void __launch_bounds__(64) k_scan(int32_t* __restrict__ counts, int n_wg, int32_t* __restrict__ totals) {
  const int bin = blockIdx.x, lane = threadIdx.x;
  int32_t* c = counts + (int64_t)bin * n_wg;
  int carry = 0;
  for (int base = 0; base < n_wg; base += 64) {
    const int i = base + lane;
    const int v = i < n_wg ? c[i] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (i < n_wg) c[i] = carry + incl - v;
    carry += __shfl(incl, 63, 64);
  }
  if (lane == 0) totals[bin] = carry;
}

// K3: records to recs[bin_base[bin] + counts[bin][wg] + local rank]; local rank through LDS cursors
struct Rec { uint32_t row; float g0, g1; };
__global__ void __launch_bounds__(kThreads) k_write(int64_t M, int n_wg, const int32_t* __restrict__ counts,
                                                    const int64_t* __restrict__ bin_base, Rec* __restrict__ recs) {
  __shared__ int cursor[kSlices];
  const int level = blockIdx.y, wg = blockIdx.x;
  if (threadIdx.x < kSlices)
    cursor[threadIdx.x] = counts[((int64_t)level * kSlices + threadIdx.x) * n_wg + wg];
  __syncthreads();
  const int64_t m = (int64_t)wg * kSamplesPerWG + threadIdx.x;
  if (m < M)
    for (int k = 0; k < 8; ++k) {
      const uint32_t row = row_of(level, m, k);
      const int s = row >> kSliceLog2;
      const int pos = atomicAdd(&cursor[s], 1);
      Rec r; r.row = row & ((1u << kSliceLog2) - 1u); r.g0 = 1.0f; r.g1 = 0.5f;
      recs[bin_base[level * kSlices + s] + pos] = r;
    }
}

// K4: one workgroup per bin
__global__ void __launch_bounds__(512) k_accum(const Rec* __restrict__ recs, const int64_t* __restrict__ bin_base,
                                               const int32_t* __restrict__ totals, float* __restrict__ table) {
  extern __shared__ float slice[];                      // 2 * 16384 floats = 128 KB
  const int bin = blockIdx.x;
  for (int i = threadIdx.x; i < 2 << kSliceLog2; i += 512) slice[i] = 0.f;
  __syncthreads();
  const Rec* r = recs + bin_base[bin];
  const int n = totals[bin];
  // eight records in flight per thread: the loop is a chain of global loads otherwise (267 us -> see the printout)
  int i = threadIdx.x;
  for (; i + 7 * 512 < n; i += 8 * 512) {
    Rec v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r[i + u * 512];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      atomicAdd(&slice[2 * v[u].row], v[u].g0);
      atomicAdd(&slice[2 * v[u].row + 1], v[u].g1);
    }
  }
  for (; i < n; i += 512) {
    const Rec v = r[i];
    atomicAdd(&slice[2 * v.row], v.g0);
    atomicAdd(&slice[2 * v.row + 1], v.g1);
  }
  __syncthreads();
  float4* out = reinterpret_cast<float4*>(table + ((size_t)bin << (kSliceLog2 + 1)));
  const float4* s4 = reinterpret_cast<const float4*>(slice);
  for (int i = threadIdx.x; i < (2 << kSliceLog2) / 4; i += 512) out[i] = s4[i];
}

// baseline: the same records with global atomics (x-pair merging not modelled: 2 lanes per record, 8-byte groups)
__global__ void __launch_bounds__(kThreads) k_atomic(int64_t M, float* __restrict__ table) {
  const int level = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * kSamplesPerWG + threadIdx.x;
  if (m >= M) return;
  for (int k = 0; k < 8; ++k) {
    const uint32_t row = row_of(level, m, k);
    float* p = table + (((size_t)level << kRowsLog2) + row) * 2;
    atomicAdd(p, 1.0f);
    atomicAdd(p + 1, 0.5f);
  }
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int L = 11;                                  // the hashed levels of the 16
  const int64_t M = 200000;
  const int n_wg = (int)((M + kSamplesPerWG - 1) / kSamplesPerWG);
  const int n_bins = L * kSlices;
  int32_t *counts, *totals; int64_t* bin_base; Rec* recs; float* table;
  CK(hipMalloc(&counts, (size_t)n_bins * n_wg * 4)); CK(hipMalloc(&totals, n_bins * 4)); CK(hipMalloc(&bin_base, n_bins * 8));
  CK(hipMalloc(&recs, (size_t)L * M * 8 * sizeof(Rec) + 4096)); CK(hipMalloc(&table, ((size_t)L << kRowsLog2) * 8));
  hipEvent_t e[6]; for (auto& x : e) CK(hipEventCreate(&x));
  int32_t* h_tot = (int32_t*)malloc(n_bins * 4); int64_t* h_base = (int64_t*)malloc(n_bins * 8);
  CK(hipFuncSetAttribute((const void*)k_accum, hipFuncAttributeMaxDynamicSharedMemorySize, (2 << kSliceLog2) * 4));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e[0]));
    k_count<<<dim3(n_wg, L), kThreads>>>(M, n_wg, counts);
    CK(hipEventRecord(e[1]));
    k_scan<<<n_bins, 64>>>(counts, n_wg, totals);
    CK(hipEventRecord(e[2]));
    if (rep == 0) {          // bin bases from the totals (on the host here; a third tiny scan kernel in a real version)
      CK(hipMemcpy(h_tot, totals, n_bins * 4, hipMemcpyDeviceToHost));
      int64_t acc = 0; for (int b = 0; b < n_bins; ++b) { h_base[b] = acc; acc += h_tot[b]; }
      CK(hipMemcpy(bin_base, h_base, n_bins * 8, hipMemcpyHostToDevice));
      printf("records %lld, per bin %d..%d\n", (long long)acc, h_tot[0], h_tot[n_bins - 1]);
      CK(hipEventRecord(e[2]));
    }
    k_write<<<dim3(n_wg, L), kThreads>>>(M, n_wg, counts, bin_base, recs);
    CK(hipEventRecord(e[3]));
    k_accum<<<n_bins, 512, (2 << kSliceLog2) * 4>>>(recs, bin_base, totals, table);
    CK(hipEventRecord(e[4]));
    CK(hipEventSynchronize(e[4]));
    float t[4]; for (int i = 0; i < 4; ++i) CK(hipEventElapsedTime(&t[i], e[i], e[i + 1]));
    printf("rep %d: count %.1f us  scan %.1f us  write %.1f us  accumulate %.1f us  total %.1f us\n", rep, t[0] * 1e3, t[1] * 1e3,
           t[2] * 1e3, t[3] * 1e3, (t[0] + t[1] + t[2] + t[3]) * 1e3);
  }
  float* h = (float*)malloc(((size_t)L << kRowsLog2) * 8);
  CK(hipMemcpy(h, table, ((size_t)L << kRowsLog2) * 8, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < ((size_t)L << kRowsLog2) * 2; ++i) s += h[i];
  printf("checksum %.1f expected %.1f\n", s, 1.5 * L * M * 8);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(table, 0, ((size_t)L << kRowsLog2) * 8));
    CK(hipEventRecord(e[0]));
    k_atomic<<<dim3(n_wg, L), kThreads>>>(M, table);
    CK(hipEventRecord(e[1])); CK(hipEventSynchronize(e[1]));
    float t; CK(hipEventElapsedTime(&t, e[0], e[1]));
    printf("atomic baseline (2 x %lld lane-atomics, 8-byte groups): %.1f us\n", (long long)L * M * 8, t * 1e3);
  }
  return 0;
}
