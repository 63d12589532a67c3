// Hash-grid cell location / indexing and SH-4 evaluation shared by encoders.hip and
// field_fused.hip.  Index-determining arithmetic follows oracle/hashgrid.py operation by
// operation (files including this header are built with -ffp-contract=off).
#pragma once
#include "common.h"

namespace inr {

// Kernel-argument copy of inr_grid_desc (passed by value: lives in SGPRs / the kernarg segment).
struct GridDesc {
  int num_levels;
  uint32_t offsets[INR_MAX_LEVELS + 1];
  float scales[INR_MAX_LEVELS];
  uint32_t res1[INR_MAX_LEVELS];      // resolution + 1 (dense stride)
  uint32_t mask[INR_MAX_LEVELS];      // rows - 1 for hashed levels (rows is a power of two), 0 = dense
};

inline int make_grid_desc(const inr_grid_desc* d, GridDesc& G) {
  if (d->num_levels < 1 || d->num_levels > INR_MAX_LEVELS) {
    set_error("grid desc: num_levels %d out of range", d->num_levels);
    return INR_EINVAL;
  }
  if (d->level_dim != 2) {
    set_error("grid desc: level_dim %d unsupported (2 only)", d->level_dim);
    return INR_EINVAL;
  }
  G.num_levels = d->num_levels;
  for (int l = 0; l <= INR_MAX_LEVELS; ++l) G.offsets[l] = l <= d->num_levels ? d->offsets[l] : 0;
  for (int l = 0; l < INR_MAX_LEVELS; ++l) {
    const bool live = l < d->num_levels;
    G.scales[l] = live ? d->scales[l] : 0.f;
    G.res1[l] = live ? d->resolutions[l] + 1 : 1;
    G.mask[l] = 0;
    if (live && d->hashed[l]) {
      const uint32_t rows = d->offsets[l + 1] - d->offsets[l];
      if (rows == 0 || (rows & (rows - 1)) != 0) {
        set_error("grid desc: hashed level %d has %u rows (must be a power of two)", l, rows);
        return INR_EINVAL;
      }
      G.mask[l] = rows - 1;
    }
  }
  return INR_OK;
}

// Upstream's gridencoder flags a sample whose normalised coordinate leaves [0,1] on any axis (flag_oob): its
// features are zero and it adds nothing to the table gradient.  NaN counts as outside.
__device__ __forceinline__ bool oob01(float x0, float x1, float x2) {
  return !(x0 >= 0.0f && x0 <= 1.0f && x1 >= 0.0f && x1 <= 1.0f && x2 >= 0.0f && x2 <= 1.0f);
}

struct Cell {
  uint32_t gx, gy, gz;   // lower corner
  float fx, fy, fz;      // fractional position inside the cell
};

// x01 in [0,1]^3 -> cell of level l.  pos = x01*scale + 0.5 as a separate mul and add.
__device__ __forceinline__ void locate(const GridDesc& G, int l, float x0, float x1, float x2, Cell& c) {
  const float s = G.scales[l];
  const float px = x0 * s + 0.5f, py = x1 * s + 0.5f, pz = x2 * s + 0.5f;
  const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
  c.fx = px - flx; c.fy = py - fly; c.fz = pz - flz;
  c.gx = (uint32_t)flx; c.gy = (uint32_t)fly; c.gz = (uint32_t)flz;
}

// corner k: bit d of k selects the +1 neighbour on axis d; weight = (wx*wy)*wz
__device__ __forceinline__ float corner_weight(const Cell& c, int k) {
  const float wx = (k & 1) ? c.fx : 1.0f - c.fx;
  const float wy = (k & 2) ? c.fy : 1.0f - c.fy;
  const float wz = (k & 4) ? c.fz : 1.0f - c.fz;
  return (wx * wy) * wz;
}

// row inside level l (add G.offsets[l] for the absolute row)
__device__ __forceinline__ uint32_t corner_index(const GridDesc& G, int l, const Cell& c, int k) {
  const uint32_t cx = c.gx + (k & 1), cy = c.gy + ((k >> 1) & 1), cz = c.gz + ((k >> 2) & 1);
  const uint32_t m = G.mask[l];
  if (m) return (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) & m;
  const uint32_t s = G.res1[l];
  return cx + cy * s + cz * s * s;
}

// degree-4 real spherical harmonics
__device__ __forceinline__ void sh4(float x, float y, float z, float* v) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  v[0] = 0.28209479177387814f;
  v[1] = -0.48860251190291987f * y;
  v[2] = 0.48860251190291987f * z;
  v[3] = -0.48860251190291987f * x;
  v[4] = 1.0925484305920792f * xy;
  v[5] = -1.0925484305920792f * yz;
  v[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  v[7] = -1.0925484305920792f * xz;
  v[8] = 0.54627421529603959f * (x2 - y2);
  v[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
  v[10] = 2.8906114426405538f * xy * z;
  v[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
  v[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
  v[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
  v[14] = 1.4453057213202769f * z * (x2 - y2);
  v[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}

}  // namespace inr
