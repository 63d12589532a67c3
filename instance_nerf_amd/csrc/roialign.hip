// 3-D RoIAlign for gfx950: the op behind the reference's one in-tree FFI call,
// roi_align.roi_align.roi_align_3d (/root/reference/nerf_rcnn/model/utils.py:604-609; callers
// model/poolers.py:144,174 and model/nerf_rcnn.py:831).  Semantics = torchvision roi_align
// (aligned=False, adaptive ceil(roi/out) sampling grid, average pooling) on three axes, as the
// reference's wrapper documents (utils.py:556-592); the extension itself is an un-vendored submodule.
// HBM-bound trilinear gathers: one lane per output voxel, h (the contiguous axis of [N,C,W,L,H]) fastest.
#include "common.h"

namespace inr {

struct RoiGeom {
  float start[3], bin[3];
  int grid[3];
  float inv_count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float scale, int ow, int ol, int oh) {
  RoiGeom g;
  const int osz[3] = {ow, ol, oh};
  int count = 1;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float s = roi[a] * scale, e = roi[a + 3] * scale;
    const float size = fmaxf(e - s, 1.0f);
    g.start[a] = s;
    g.bin[a] = size / (float)osz[a];
    g.grid[a] = (int)ceilf(size / (float)osz[a]);
    count *= g.grid[a];
  }
  g.inv_count = 1.0f / (float)max(count, 1);
  return g;
}

struct Tri {
  int lo[3], hi[3];
  float fr[3];
  bool inside;
};
__device__ __forceinline__ Tri tri_setup(float x, float y, float z, int W, int L, int H) {
  Tri t;
  t.inside = !(x < -1.0f || x > (float)W || y < -1.0f || y > (float)L || z < -1.0f || z > (float)H);
  const float v[3] = {x, y, z};
  const int n[3] = {W, L, H};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float c = fmaxf(v[a], 0.0f);
    int l = (int)c;
    int h;
    if (l >= n[a] - 1) { l = h = n[a] - 1; c = (float)l; } else { h = l + 1; }
    t.lo[a] = l; t.hi[a] = h; t.fr[a] = c - (float)l;
  }
  return t;
}

__global__ void __launch_bounds__(256) k_roi_align3d_fwd(const float* __restrict__ in, const float* __restrict__ rois,
                                                         const int32_t* __restrict__ roi_inds, int C, int W, int L,
                                                         int H, int64_t total, int ow, int ol, int oh, float scale,
                                                         float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ph = (int)(idx % oh), pl = (int)((idx / oh) % ol), pw = (int)((idx / ((int64_t)oh * ol)) % ow);
  const int c = (int)((idx / ((int64_t)oh * ol * ow)) % C);
  const int64_t k = idx / ((int64_t)oh * ol * ow * C);
  const RoiGeom g = roi_geom(rois + k * 6, scale, ow, ol, oh);
  const float* vol = in + ((int64_t)roi_inds[k] * C + c) * (int64_t)W * L * H;
  float acc = 0.0f;
  for (int ix = 0; ix < g.grid[0]; ++ix) {
    const float x = g.start[0] + pw * g.bin[0] + ((float)ix + 0.5f) * g.bin[0] / (float)g.grid[0];
    for (int iy = 0; iy < g.grid[1]; ++iy) {
      const float y = g.start[1] + pl * g.bin[1] + ((float)iy + 0.5f) * g.bin[1] / (float)g.grid[1];
      for (int iz = 0; iz < g.grid[2]; ++iz) {
        const float z = g.start[2] + ph * g.bin[2] + ((float)iz + 0.5f) * g.bin[2] / (float)g.grid[2];
        const Tri t = tri_setup(x, y, z, W, L, H);
        if (!t.inside) continue;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int cx = (cc & 1) ? t.hi[0] : t.lo[0], cy = (cc & 2) ? t.hi[1] : t.lo[1], cz = (cc & 4) ? t.hi[2] : t.lo[2];
          const float w = ((cc & 1) ? t.fr[0] : 1.0f - t.fr[0]) * ((cc & 2) ? t.fr[1] : 1.0f - t.fr[1]) *
                          ((cc & 4) ? t.fr[2] : 1.0f - t.fr[2]);
          acc += w * vol[((int64_t)cx * L + cy) * H + cz];
        }
      }
    }
  }
  out[idx] = acc * g.inv_count;
}

__global__ void __launch_bounds__(256) k_roi_align3d_bwd(const float* __restrict__ gout, const float* __restrict__ rois,
                                                         const int32_t* __restrict__ roi_inds, int C, int W, int L,
                                                         int H, int64_t total, int ow, int ol, int oh, float scale,
                                                         float* __restrict__ gin) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ph = (int)(idx % oh), pl = (int)((idx / oh) % ol), pw = (int)((idx / ((int64_t)oh * ol)) % ow);
  const int c = (int)((idx / ((int64_t)oh * ol * ow)) % C);
  const int64_t k = idx / ((int64_t)oh * ol * ow * C);
  const RoiGeom g = roi_geom(rois + k * 6, scale, ow, ol, oh);
  float* vol = gin + ((int64_t)roi_inds[k] * C + c) * (int64_t)W * L * H;
  const float go = gout[idx] * g.inv_count;
  if (go == 0.0f) return;
  for (int ix = 0; ix < g.grid[0]; ++ix) {
    const float x = g.start[0] + pw * g.bin[0] + ((float)ix + 0.5f) * g.bin[0] / (float)g.grid[0];
    for (int iy = 0; iy < g.grid[1]; ++iy) {
      const float y = g.start[1] + pl * g.bin[1] + ((float)iy + 0.5f) * g.bin[1] / (float)g.grid[1];
      for (int iz = 0; iz < g.grid[2]; ++iz) {
        const float z = g.start[2] + ph * g.bin[2] + ((float)iz + 0.5f) * g.bin[2] / (float)g.grid[2];
        const Tri t = tri_setup(x, y, z, W, L, H);
        if (!t.inside) continue;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          const int cx = (cc & 1) ? t.hi[0] : t.lo[0], cy = (cc & 2) ? t.hi[1] : t.lo[1], cz = (cc & 4) ? t.hi[2] : t.lo[2];
          const float w = ((cc & 1) ? t.fr[0] : 1.0f - t.fr[0]) * ((cc & 2) ? t.fr[1] : 1.0f - t.fr[1]) *
                          ((cc & 4) ? t.fr[2] : 1.0f - t.fr[2]);
          atomicAdd(vol + ((int64_t)cx * L + cy) * H + cz, w * go);
        }
      }
    }
  }
}

}  // namespace inr

using namespace inr;

extern "C" {

int inr_roi_align_3d_forward(const float* input, const float* rois, const int32_t* roi_inds, int32_t N, int32_t C,
                             int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w, int32_t out_l, int32_t out_h,
                             float spatial_scale, float* out, inr_stream_t s) {
  INR_REQUIRE(K >= 0 && N >= 0 && C > 0 && W > 0 && L > 0 && H > 0 && out_w > 0 && out_l > 0 && out_h > 0, "bad sizes");
  if (K == 0) return INR_OK;
  INR_REQUIRE(input && rois && roi_inds && out && N > 0, "null pointer");
  const int64_t total = K * C * out_w * out_l * out_h;
  k_roi_align3d_fwd<<<blocks_for(total, 256), 256, 0, as_stream(s)>>>(input, rois, roi_inds, C, W, L, H, total, out_w,
                                                                      out_l, out_h, spatial_scale, out);
  return check_launch("roi_align_3d_forward");
}

int inr_roi_align_3d_backward(const float* grad_out, const float* rois, const int32_t* roi_inds, int32_t N, int32_t C,
                              int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w, int32_t out_l, int32_t out_h,
                              float spatial_scale, float* grad_input, inr_stream_t s) {
  INR_REQUIRE(K >= 0 && N >= 0 && C > 0 && W > 0 && L > 0 && H > 0 && out_w > 0 && out_l > 0 && out_h > 0, "bad sizes");
  if (K == 0) return INR_OK;
  INR_REQUIRE(grad_out && rois && roi_inds && grad_input && N > 0, "null pointer");
  const int64_t total = K * C * out_w * out_l * out_h;
  k_roi_align3d_bwd<<<blocks_for(total, 256), 256, 0, as_stream(s)>>>(grad_out, rois, roi_inds, C, W, L, H, total,
                                                                      out_w, out_l, out_h, spatial_scale, grad_input);
  return check_launch("roi_align_3d_backward");
}

}  // extern "C"
