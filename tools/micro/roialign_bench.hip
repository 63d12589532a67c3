// Sweeps the launch parameters of the separable RoIAlign-3D forward kernel on BASELINE configs[4]
// (channels per workgroup, LDS budget of the slab intermediates) and times the lane-per-output kernel beside it.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/roialign_bench.hip -o tools/micro/roialign_bench
#include "../../instance_nerf_amd/csrc/roialign.hip"

#include <random>
#include <string>
#include <vector>

namespace inr {
void set_error(const char* fmt, ...) { (void)fmt; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const bool only_full = argc > 1;
  const int C = 256, W = 40, L = 40, H = 40, K = 256, o = 10;
  std::mt19937 rng(0);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::vector<float> hin((size_t)C * W * L * H), hro(K * 6);
  for (auto& v : hin) v = U(rng) * 2 - 1;
  for (int k = 0; k < K; ++k)
    for (int a = 0; a < 3; ++a) { hro[k * 6 + a] = U(rng) * 100; hro[k * 6 + 3 + a] = hro[k * 6 + a] + 10 + U(rng) * 50; }
  std::vector<int32_t> hind(K, 0);
  float *in, *rois, *out;
  int32_t* inds;
  const size_t nout = (size_t)K * C * o * o * o;
  CK(hipMalloc(&in, hin.size() * 4)); CK(hipMalloc(&rois, hro.size() * 4)); CK(hipMalloc(&inds, K * 4)); CK(hipMalloc(&out, nout * 4));
  CK(hipMemcpy(in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(rois, hro.data(), hro.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(inds, hind.data(), K * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* what, int cpb, int tmp) -> int {
    SepArgs A;
    A.C = C; A.W = W; A.L = L; A.H = H; A.ow = o; A.ol = o; A.oh = o; A.scale = 0.25f; A.cpb = cpb;
    A.ngroups = (C + cpb - 1) / cpb; A.K = K; A.tmp_floats = tmp;
    const int lds = (o * W * 3 + 6 * o) * 4 + (int)sizeof(SepRoi) + 16 + tmp * 4;
    const unsigned grid = 8u * K * ((A.ngroups + 7) / 8);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      CK(hipEventRecord(e0));
      kern<<<grid, SEP_THREADS, lds>>>(in, rois, inds, A, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) best = std::min(best, ms);
    }
    printf("separable %-28s cpb %2d tmp %5d floats (lds %6d B): %.4f ms\n", what, cpb, tmp, lds, best);
    return 0;
  };
  if (argc > 1 && std::string(argv[1]) == "bwd") {
    // the backward's launch parameters: channels per workgroup x LDS budget of the slab intermediates (gout rides on top);
    // `out` holds the output gradient (whatever the forward sweep left there or zeros), gin is accumulated into, never read
    float* gin;
    CK(hipMalloc(&gin, hin.size() * 4));
    CK(hipMemset(gin, 0, hin.size() * 4));
    CK(hipMemset(out, 0, nout * 4));
    for (int cpb : {8, 16, 32, 64})
      for (int tmp : {2048, 3072, 4096, 6144, 10240}) {
        SepArgs A;
        A.C = C; A.W = W; A.L = L; A.H = H; A.ow = o; A.ol = o; A.oh = o; A.scale = 0.25f; A.cpb = cpb;
        A.ngroups = (C + cpb - 1) / cpb; A.K = K; A.tmp_floats = tmp + SEP_CH * o * o * o;
        const int lds = (o * W * 3 + 6 * o + 2 * (W + L + H)) * 4 + (int)sizeof(SepRoi) + 16 + A.tmp_floats * 4;
        const unsigned grid = 8u * K * ((A.ngroups + 7) / 8);
        float best = 1e9f;
        for (int it = 0; it < 6; ++it) {
          CK(hipEventRecord(e0));
          k_roi_align3d_sep_bwd<<<grid, SEP_THREADS, lds>>>(out, rois, inds, A, gin);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it) best = std::min(best, ms);
        }
        printf("separable backward cpb %2d slab budget %5d floats (lds %6d B): %.4f ms (without the zero fill)\n", cpb, tmp, lds, best);
      }
    {   // the workspace form: channels-fastest scratch (zero fill timed with it) + transposing copy
      float* gt;
      CK(hipMalloc(&gt, hin.size() * 4));
      SepArgs A;
      A.C = C; A.W = W; A.L = L; A.H = H; A.ow = o; A.ol = o; A.oh = o; A.scale = 0.25f; A.cpb = 16;
      A.ngroups = C / 16; A.K = K;
      for (int tmp : {4096, 6144, 8192, 10240}) {
        A.tmp_floats = tmp;
        const int lds = (o * W * 3 + 6 * o + 2 * (W + L + H)) * 4 + (int)sizeof(SepRoi) + 16 + tmp * 4;
        const unsigned grid = 8u * K * ((A.ngroups + 7) / 8);
        float best = 1e9f, best_k = 1e9f;
        hipEvent_t ea, eb;
        CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        for (int it = 0; it < 6; ++it) {
          CK(hipEventRecord(e0));
          CK(hipMemsetAsync(gt, 0, hin.size() * 4));
          CK(hipEventRecord(ea));
          k_roi_align3d_sep_bwd_cl<<<grid, SEP_THREADS, lds>>>(out, rois, inds, A, gt);
          CK(hipEventRecord(eb));
          k_channels_last_to_planes<<<dim3((unsigned)((W * L * H + 31) / 32), C / 32, 1), 256>>>(gt, gin, (int64_t)W * L * H, C);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms, mk;
          CK(hipEventElapsedTime(&ms, e0, e1));
          CK(hipEventElapsedTime(&mk, ea, eb));
          if (it) { best = std::min(best, ms); best_k = std::min(best_k, mk); }
        }
        printf("workspace form (16 channels per workgroup) slab budget %5d floats (lds %6d B): %.4f ms, of which the kernel %.4f\n",
               tmp, lds, best, best_k);
      }
    }
    return 0;
  }
  if (only_full) { run(k_roi_align3d_sep_fwd<4, false>, "full", 16, 10240); return 0; }
  for (int cpb : {8, 16, 32})
    for (int tmp : {6144, 8192, 10240}) run(k_roi_align3d_sep_fwd<4, false>, "full", cpb, tmp);
  {
    float best = 1e9f;
    const int64_t total = (int64_t)nout;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      k_roi_align3d_fwd<<<blocks_for(total, 256), 256>>>(in, rois, inds, C, W, L, H, total, o, o, o, 0.25f, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) best = std::min(best, ms);
    }
    printf("lane per output: %.4f ms\n", best);
  }
  {  // the floor of the write alone: 262 MB fill
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0));
      CK(hipMemsetAsync(out, 0, nout * 4));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) best = std::min(best, ms);
    }
    printf("memset of the output (262 MB): %.4f ms\n", best);
  }
  return 0;
}
