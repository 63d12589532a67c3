// Microbenchmark (round 3): what prices the table-gradient scatter on MI355X?
//   A  fp32 atomic adds in GROUPS of W adjacent floats (W = 1..32), groups scattered over a 49 MB table:
//      requests/s and lanes/s as a function of the request width;
//   B  the same addresses with plain 16-byte stores and 16-byte read-modify-write (no atomic unit involved):
//      the ceiling an ownership scheme (no two lanes share a row) would have;
//   C  cache-policy bits on the atomic (sc1 = system scope, nt) through inline asm;
//   D  occupancy: the same scatter from 256 / 1024 / 4096 / 16384 workgroups;
//   E  ds_add_f32 into a 64 KB LDS slice (what an LDS-privatised coarse level would pay).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_width_bench atomic_width_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// A: lane i belongs to group i / W and adds to float (base(group) + i % W); base is W-aligned.
template <int W, int POLICY>
__global__ void k_atomic_groups(float* table, uint32_t n_floats, uint64_t n_lanes, uint32_t seed) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lanes; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t grp = (uint32_t)(i / W), sub = (uint32_t)(i % W);
    const uint32_t base = (hash32(grp ^ seed) % (n_floats / W)) * W;
    float* p = table + base + sub;
    if (POLICY == 0) atomicAdd(p, 1.0f);
    else if (POLICY == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(1.0f) : "memory");
    else if (POLICY == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(1.0f) : "memory");
    else asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(1.0f) : "memory");
  }
}

// B: one lane per 16-byte piece.  MODE 0 store, 1 load + add + store (racy on purpose: rate only), 2 load only
template <int MODE>
__global__ void k_plain16(float4* table, uint32_t n_vec, uint64_t n_ops, uint32_t seed, float4* sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ops; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = hash32((uint32_t)i ^ seed) % n_vec;
    if (MODE == 0) table[r] = make_float4(1, 2, 3, 4);
    else if (MODE == 1) { float4 v = table[r]; v.x += 1; v.y += 1; v.z += 1; v.w += 1; table[r] = v; }
    else { float4 v = table[r]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  if (MODE == 2 && acc.x == 12345.678f) sink[0] = acc;
}

// E: every workgroup adds n_per_wg values into its own 64 KB LDS slice, then writes the slice out.
__global__ void __launch_bounds__(512) k_lds_add(float* out, uint32_t n_per_wg, uint32_t seed, int conflict_free) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_per_wg; i += blockDim.x) {
    uint32_t a = hash32((i + blockIdx.x * n_per_wg) ^ seed) & 16383u;
    if (conflict_free) a = (a & ~63u) | (threadIdx.x & 63u);     // one lane per bank
    atomicAdd(&lds[a], 1.0f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) out[(size_t)blockIdx.x * 16384 + i] = lds[i];
}

template <typename F>
static float best_of(int reps, hipEvent_t e0, hipEvent_t e1, F&& launch) {
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    launch(r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const bool with_asm = argc > 1 && argv[1][0] == 'c';           // section C (inline-asm cache-policy bits) on request
  const uint32_t rows = 6119864, n_floats = rows * 2;       // the hash table of one field
  float* table; CK(hipMalloc(&table, (size_t)n_floats * 4 + 4096));
  float4* sink; CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 0, (size_t)n_floats * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t lanes = 48ull << 20;                        // ~ 190k samples x 16 levels x 16 lane-atomics
  printf("A: atomics in groups of W adjacent floats, %llu M lane-atomics, 2048 x 256 threads\n", (unsigned long long)(lanes >> 20));
#define RUN_A(W) { float ms = best_of(3, e0, e1, [&](int r) { k_atomic_groups<W, 0><<<2048, 256>>>(table, n_floats, lanes, r); }); \
    printf("  W=%2d (%3d B): %.3f ms  %6.1f G lanes/s  %6.2f G requests/s\n", W, W * 4, ms, lanes / ms / 1e6, lanes / (double)W / ms / 1e6); }
  RUN_A(1) RUN_A(2) RUN_A(4) RUN_A(8) RUN_A(16) RUN_A(32)
  printf("A': the same with a FIXED number of requests (12 M groups)\n");
#define RUN_A2(W) { const uint64_t ln = (12ull << 20) * W; float ms = best_of(3, e0, e1, [&](int r) { k_atomic_groups<W, 0><<<2048, 256>>>(table, n_floats, ln, r); }); \
    printf("  W=%2d: %.3f ms  %6.1f G lanes/s  %6.2f G requests/s\n", W, ms, ln / ms / 1e6, ln / (double)W / ms / 1e6); }
  RUN_A2(1) RUN_A2(2) RUN_A2(4) RUN_A2(8) RUN_A2(16)
  printf("B: plain 16-byte accesses at the same scattered addresses (12 M pieces)\n");
  {
    const uint64_t n = 12ull << 20;
    float ms0 = best_of(3, e0, e1, [&](int r) { k_plain16<0><<<2048, 256>>>((float4*)table, n_floats / 4, n, r, sink); });
    float ms1 = best_of(3, e0, e1, [&](int r) { k_plain16<1><<<2048, 256>>>((float4*)table, n_floats / 4, n, r, sink); });
    float ms2 = best_of(3, e0, e1, [&](int r) { k_plain16<2><<<2048, 256>>>((float4*)table, n_floats / 4, n, r, sink); });
    printf("  store %.3f ms (%.1f G/s)   load+add+store %.3f ms (%.1f G/s)   load %.3f ms (%.1f G/s)\n", ms0, n / ms0 / 1e6,
           ms1, n / ms1 / 1e6, ms2, n / ms2 / 1e6);
  }
  printf("D: occupancy (W=4, 48 M lanes)\n");
  for (int nb : {256, 512, 1024, 4096, 16384}) {
    float ms = best_of(3, e0, e1, [&](int r) { k_atomic_groups<4, 0><<<nb, 256>>>(table, n_floats, lanes, r); });
    printf("  %5d workgroups: %.3f ms  %6.1f G lanes/s\n", nb, ms, lanes / ms / 1e6);
  }
  printf("D': smaller tables (W=4, 48 M lanes): is it capacity?\n");
  for (uint32_t nf : {1u << 16, 1u << 20, 1u << 22, 1u << 24}) {
    float ms = best_of(3, e0, e1, [&](int r) { k_atomic_groups<4, 0><<<2048, 256>>>(table, nf, lanes, r); });
    printf("  %8.2f MB: %.3f ms  %6.1f G lanes/s\n", nf * 4.0 / 1048576.0, ms, lanes / ms / 1e6);
  }
  printf("E: ds_add_f32 into a 64 KB slice per workgroup (256 workgroups x 512 threads, 188 k adds each = 48 M)\n");
  {
    float* out; CK(hipMalloc(&out, (size_t)256 * 16384 * 4));
    for (int cf = 0; cf < 2; ++cf) {
      float ms = best_of(3, e0, e1, [&](int r) { k_lds_add<<<256, 512>>>(out, 188 * 1024, r, cf); });
      printf("  %s: %.3f ms  %6.1f G adds/s\n", cf ? "one lane per bank" : "random banks   ", ms, 256.0 * 188 * 1024 / ms / 1e6);
    }
  }
  CK(hipDeviceSynchronize());
  if (with_asm) {
  printf("C: cache-policy bits on the atomic (W=4, 48 M lanes)\n");
#define RUN_C(P, name) { float ms = best_of(3, e0, e1, [&](int r) { k_atomic_groups<4, P><<<2048, 256>>>(table, n_floats, lanes, r); }); \
    printf("  %-10s %.3f ms  %6.1f G lanes/s\n", name, ms, lanes / ms / 1e6); }
  RUN_C(0, "atomicAdd") RUN_C(3, "asm plain") RUN_C(1, "asm sc1") RUN_C(2, "asm nt")
  }
  CK(hipDeviceSynchronize());
  return 0;
}
