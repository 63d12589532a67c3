// Does `v_add_f32_dpp ... wave_shr:1` on gfx950 give lane j the sum after j sequential additions?
// Compares against the select-and-add loop of march_ray_coop, bit for bit.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
__global__ void k(float* a, float* b, const float* ts, const float* ds) {
  const int lane = threadIdx.x & 63;
  const float t = ts[blockIdx.x], d = ds[blockIdx.x];
  float c = t;
  for (int j = 0; j < 63; ++j) c = c + (j < lane ? d : 0.0f);
  float e = t;
  asm volatile(
      ".rept 63\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
      ".endr\n"
      : "+v"(e)
      : "v"(d));
  a[blockIdx.x * 64 + lane] = c;
  b[blockIdx.x * 64 + lane] = e;
}
int main() {
  const int n = 1 << 16;
  float *ts, *ds, *a, *b;
  hipMallocManaged(&ts, n * 4); hipMallocManaged(&ds, n * 4);
  hipMallocManaged(&a, n * 256); hipMallocManaged(&b, n * 256);
  srand(1);
  for (int i = 0; i < n; ++i) {
    ts[i] = 0.01f + 6.0f * (rand() / (float)RAND_MAX);
    ds[i] = 0.0005f + 0.02f * (rand() / (float)RAND_MAX);
  }
  k<<<n, 64>>>(a, b, ts, ds);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  long bad = 0;
  for (long i = 0; i < (long)n * 64; ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
  printf("dpp chain: %ld of %ld lanes differ; sample lane63 %.9g vs %.9g\n", bad, (long)n * 64, a[63], b[63]);
  return bad != 0;
}
