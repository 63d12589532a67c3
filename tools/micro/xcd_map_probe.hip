// Which XCD does workgroup b of a launch run on?  The field kernels' schedules take b % 8 for it (speed only, never
// results).  Built as a shared library (tools/_probe/libxcdmap.so) so that bench.py can ask INSIDE its own process, on
// its own stream, right after the timed region: profiles/r03_NOTES.txt 21 - is the slow mode of some calls a launch
// whose workgroups are not dealt round robin?
// hipcc --offload-arch=gfx950 -O3 -shared -fPIC xcd_map_probe.hip -o ../_probe/libxcdmap.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(512) k_xcd_of_block(int32_t* out, int spin) {
  extern __shared__ float pad[];
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  float acc = 0.f;                                   // stay resident for a while, like a persistent workgroup
  for (int i = 0; i < spin; ++i) acc += __sinf(acc + (float)i);
  if (threadIdx.x == 0) {
    pad[0] = acc;
    out[blockIdx.x] = (int32_t)(v & 15u) | (acc == 12345.f ? 16 : 0);
  }
}
extern "C" int xcd_map(int blocks, int threads, int lds_bytes, int spin, int32_t* out_host, void* stream) {
  int32_t* dev = nullptr;
  if (hipMalloc(&dev, sizeof(int32_t) * blocks) != hipSuccess) return -1;
  hipStream_t s = (hipStream_t)stream;
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)k_xcd_of_block, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  k_xcd_of_block<<<blocks, threads, lds_bytes, s>>>(dev, spin);
  int rc = hipGetLastError() == hipSuccess ? 0 : -2;
  if (rc == 0 && hipMemcpyAsync(out_host, dev, sizeof(int32_t) * blocks, hipMemcpyDeviceToHost, s) != hipSuccess) rc = -3;
  if (hipStreamSynchronize(s) != hipSuccess) rc = -4;
  (void)hipFree(dev);
  return rc;
}
