// What do the atomics of the RoIAlign-3D backward cost by themselves, and what would another layout of the accumulation
// target cost?  BASELINE configs[4]: 256 boxes on [256, 40, 40, 40]; every box adds one value to every (channel, cell) of
// its region.  No contraction passes - only the address pattern of the final pass:
//   planes     target [C][W][L][H] (the gradient's own layout): lanes run along z then over (x, y) rows, four channel
//              planes per thread - what k_roi_align3d_sep_bwd issues
//   channels   target [W][L][H][C]: 16 lanes = 16 consecutive channels of one voxel (one 64-byte segment per request)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/micro/atomic_layout_bench.hip -o tools/micro/atomic_layout_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <random>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Box { int x0, y0, z0, sx, sy, sz; };
constexpr int C = 256, W = 40, L = 40, H = 40, K = 256, WLH = W * L * H;

// workgroup = (box, 16 channels); 4 channel runs of 4 planes each, lanes along z then rows
__global__ void __launch_bounds__(256) k_planes(const Box* boxes, float* g) {
  const int k = blockIdx.x % K, grp = blockIdx.x / K;
  const Box b = boxes[k];
  const int t = threadIdx.x;
  const int NS = 256 / b.sz, slot = t / b.sz, z = t - slot * b.sz;
  if (slot >= NS) return;
  for (int run = 0; run < 4; ++run) {
    const int c = grp * 16 + run * 4;
    for (int row = slot; row < b.sx * b.sy; row += NS) {
      const int x = row / b.sy, y = row - x * b.sy;
      float* dst = g + (int64_t)c * WLH + ((b.x0 + x) * L + (b.y0 + y)) * H + b.z0 + z;
      atomicAdd(dst, 1.0f); atomicAdd(dst + WLH, 1.0f); atomicAdd(dst + 2 * WLH, 1.0f); atomicAdd(dst + 3 * WLH, 1.0f);
    }
  }
}

// workgroup = (box, 16 channels); lane = (channel, voxel slot)
__global__ void __launch_bounds__(256) k_channels(const Box* boxes, float* g) {
  const int k = blockIdx.x % K, grp = blockIdx.x / K;
  const Box b = boxes[k];
  const int t = threadIdx.x, ch = t & 15, vs = t >> 4;
  const int n = b.sx * b.sy * b.sz;
  for (int v = vs; v < n; v += 16) {
    const int row = v / b.sz, z = v - row * b.sz;
    const int x = row / b.sy, y = row - x * b.sy;
    atomicAdd(g + ((int64_t)((b.x0 + x) * L + (b.y0 + y)) * H + b.z0 + z) * C + grp * 16 + ch, 1.0f);
  }
}

// the same with 32 channels per workgroup: lane = (channel of 32, voxel slot of 8) - one 128-byte run per voxel
__global__ void __launch_bounds__(256) k_channels32(const Box* boxes, float* g) {
  const int k = blockIdx.x % K, grp = blockIdx.x / K;
  const Box b = boxes[k];
  const int t = threadIdx.x, ch = t & 31, vs = t >> 5;
  const int n = b.sx * b.sy * b.sz;
  for (int v = vs; v < n; v += 8) {
    const int row = v / b.sz, z = v - row * b.sz;
    const int x = row / b.sy, y = row - x * b.sy;
    atomicAdd(g + ((int64_t)((b.x0 + x) * L + (b.y0 + y)) * H + b.z0 + z) * C + grp * 32 + ch, 1.0f);
  }
}

int main() {
  std::mt19937 rng(0);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::vector<Box> hb(K);
  double cells = 0, rows = 0;
  for (auto& b : hb) {
    int lo[3], n[3];
    for (int a = 0; a < 3; ++a) {
      const float s = U(rng) * 100 * 0.25f, e = s + (10 + U(rng) * 50) * 0.25f;      // the probe's boxes at scale 0.25
      lo[a] = std::max(0, (int)s);
      const int hi = std::min(39, (int)e + 1);
      n[a] = std::max(1, hi - lo[a] + 1);
    }
    b = Box{lo[0], lo[1], lo[2], n[0], n[1], n[2]};
    cells += (double)n[0] * n[1] * n[2];
    rows += (double)n[0] * n[1];
  }
  printf("%d boxes, %.0f cells and %.0f (x, y) rows per box on average; %.1f M float adds per call\n", K, cells / K, rows / K,
         cells * C / 1e6);
  Box* boxes; float* g;
  CK(hipMalloc(&boxes, K * sizeof(Box))); CK(hipMalloc(&g, (size_t)C * WLH * 4));
  CK(hipMemcpy(boxes, hb.data(), K * sizeof(Box), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* what) -> int {
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      CK(hipMemset(g, 0, (size_t)C * WLH * 4));
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) best = std::min(best, ms);
    }
    printf("%-64s %.4f ms\n", what, best);
    return 0;
  };
  time([&] { k_planes<<<K * (C / 16), 256>>>(boxes, g); }, "planes   [C][W][L][H], lanes along z, 4 planes per thread");
  time([&] { k_channels<<<K * (C / 16), 256>>>(boxes, g); }, "channels [W][L][H][C], 16 channels x 16 voxels per instruction");
  time([&] { k_channels32<<<K * (C / 32), 256>>>(boxes, g); }, "channels [W][L][H][C], 32 channels x 8 voxels per instruction");
  return 0;
}
