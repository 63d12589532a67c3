// Ray / occupancy-grid kernels for gfx950.
//
// This file is compiled with -ffp-contract=off: the ray-marching arithmetic is a
// bit-exact contract with oracle/march.py (one IEEE binary32 operation per
// written operation, no FMA), so sample positions, counts and offsets match the
// CPU oracle exactly.  Integer/byte work here is HBM/latency bound - no MFMA.
#include "common.h"

namespace inr {

static thread_local char g_err[512] = "";
int g_field_lds_min = 0;
int g_march_lds_pad = 0;
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int kRayBlock = 256;
constexpr float kSqrt3x2 = 3.4641016151377544f;

// ------------------------------------------------------------------------------------------
// a2: ray / AABB slab test
__global__ void __launch_bounds__(kRayBlock) k_near_far(const float* __restrict__ rays_o,
                                                        const float* __restrict__ rays_d,
                                                        const float* __restrict__ aabb, int64_t N,
                                                        float min_near, float* __restrict__ nears,
                                                        float* __restrict__ fars,
                                                        const int64_t* __restrict__ labels, int64_t ignore) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float near = -INFINITY, far = INFINITY;
  // a ray whose label is the ignored one is reported as a miss: it is never marched (inr_near_far_from_aabb_skip)
  bool miss = labels != nullptr && labels[n] == ignore;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float o = rays_o[n * 3 + a];
    const float rd = 1.0f / rays_d[n * 3 + a];
    const float t0 = (aabb[a] - o) * rd;
    const float t1 = (aabb[a + 3] - o) * rd;
    const float lo = t0 > t1 ? t1 : t0;
    const float hi = t0 > t1 ? t0 : t1;
    if (a > 0) miss = miss || (near > hi) || (lo > far);
    near = lo > near ? lo : near;
    far = hi < far ? hi : far;
  }
  if (near < min_near) near = min_near;
  nears[n] = miss ? 3.402823466e+38f : near;
  fars[n] = miss ? 3.402823466e+38f : far;
}

// ------------------------------------------------------------------------------------------
// a1: ray generation (replaces the dozen elementwise ops of nerf/utils.py::get_rays): pixel centre at +0.5,
// dir = ((i-cx)/fx, (j-cy)/fy, 1) normalised, rotated by the pose; same operation order as oracle/rays.py
// (no FMA) so rays are bit-identical to the CPU oracle's.
__global__ void __launch_bounds__(kRayBlock) k_get_rays(const float* __restrict__ poses, int64_t B, float fx, float fy,
                                                        float cx, float cy, int W, const int64_t* __restrict__ inds,
                                                        int64_t n, float* __restrict__ rays_o,
                                                        float* __restrict__ rays_d) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * n) return;
  const int64_t b = idx / n, k = idx - b * n;
  const int64_t pix = inds ? inds[k] : k;
  const float i = (float)(pix % W) + 0.5f, j = (float)(pix / W) + 0.5f;
  const float xs = (i - cx) / fx, ys = (j - cy) / fy;
  const float nrm = sqrtf(xs * xs + ys * ys + 1.0f * 1.0f);
  const float dx = xs / nrm, dy = ys / nrm, dz = 1.0f / nrm;
  const float* P = poses + b * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rays_d[idx * 3 + r] = dx * P[r * 4 + 0] + dy * P[r * 4 + 1] + dz * P[r * 4 + 2];
    rays_o[idx * 3 + r] = P[r * 4 + 3];
  }
}

// ------------------------------------------------------------------------------------------
// The per-step work of a training loader whose images live on the device, in ONE launch (round 6; replaces the body of
// upstream's NeRFDataset.collate [U]: torch.randint pixel draw -> get_rays -> image gather -> mask gather, eight small
// launches per step in front of a 0.75-1.3 ms step): sample k of step `step` draws pixel
//   pix = (mix64(seed + GOLDEN * (step * 2^32 + k)) >> 32) * (H*W) >> 32          (SplitMix64 finaliser, Lemire range)
// of ONE H x W image, generates its ray with k_get_rays' arithmetic (bit-identical), and gathers the pixel's colour
// (C = 3 or 4 floats) and its matched-mask label (int32 -> int64; ids >= num_instances have no logit: -1).
// Restated bit for bit by oracle/rays.py::sample_training_batch.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void __launch_bounds__(kRayBlock) k_sample_batch(const float* __restrict__ pose, float fx, float fy, float cx,
                                                            float cy, int H, int W, const float* __restrict__ image,
                                                            int C, const int32_t* __restrict__ mask, int num_instances,
                                                            uint64_t seed, uint64_t step, int64_t n,
                                                            int64_t* __restrict__ inds, float* __restrict__ rays_o,
                                                            float* __restrict__ rays_d, float* __restrict__ rgb,
                                                            int64_t* __restrict__ labels) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * ((step << 32) + (uint64_t)k));
  const uint64_t pix = ((h >> 32) * (uint64_t)((int64_t)H * W)) >> 32;
  inds[k] = (int64_t)pix;
  const float i = (float)(pix % (uint64_t)W) + 0.5f, j = (float)(pix / (uint64_t)W) + 0.5f;
  const float xs = (i - cx) / fx, ys = (j - cy) / fy;
  const float nrm = sqrtf(xs * xs + ys * ys + 1.0f * 1.0f);
  const float dx = xs / nrm, dy = ys / nrm, dz = 1.0f / nrm;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    rays_d[k * 3 + r] = dx * pose[r * 4 + 0] + dy * pose[r * 4 + 1] + dz * pose[r * 4 + 2];
    rays_o[k * 3 + r] = pose[r * 4 + 3];
  }
  if (image) {
    for (int c = 0; c < C; ++c) rgb[k * C + c] = image[pix * C + c];
  }
  if (mask) {
    const int32_t l = mask[pix];
    labels[k] = l >= num_instances ? -1 : (int64_t)l;
  }
}

// ------------------------------------------------------------------------------------------
// a3: morton / packbits
__global__ void k_morton3D(const int32_t* __restrict__ coords, int64_t N, int32_t* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  out[n] = (int32_t)morton3((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void k_morton3D_invert(const int32_t* __restrict__ idx, int64_t N, int32_t* __restrict__ coords) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const uint32_t m = (uint32_t)idx[n];
  coords[n * 3 + 0] = (int32_t)compact_bits10(m);
  coords[n * 3 + 1] = (int32_t)compact_bits10(m >> 1);
  coords[n * 3 + 2] = (int32_t)compact_bits10(m >> 2);
}
__global__ void k_packbits(const float* __restrict__ grid, int64_t n_bytes, float thresh,
                           uint8_t* __restrict__ bits) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_bytes) return;
  const float4 a = reinterpret_cast<const float4*>(grid)[n * 2];
  const float4 b = reinterpret_cast<const float4*>(grid)[n * 2 + 1];
  uint32_t v = (a.x > thresh) | ((a.y > thresh) << 1) | ((a.z > thresh) << 2) | ((a.w > thresh) << 3) |
               ((b.x > thresh) << 4) | ((b.y > thresh) << 5) | ((b.z > thresh) << 6) | ((b.w > thresh) << 7);
  bits[n] = (uint8_t)v;
}

// ---- occupancy-grid update (upstream NeRFRenderer.update_extra_state) ---------------------------------
// Query position of cell `m` (a Morton index; nullable list = identity) of one cascade: the cell centre
// x = 2c/(H-1) - 1 scaled to the cascade's half extent minus half a cell, plus a jitter of +-half a cell.  Operation
// order is oracle/occupancy.py::cell_centers (this file is built with -ffp-contract=off).
__global__ void k_occ_positions(const int32_t* __restrict__ morton, const float* __restrict__ noise, int64_t n, float Hm1,
                                float ext, float half, float* __restrict__ xyz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t m = morton ? (uint32_t)morton[i] : (uint32_t)i;
  const uint32_t c[3] = {compact_bits10(m), compact_bits10(m >> 1), compact_bits10(m >> 2)};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = (2.0f * (float)c[a] / Hm1 - 1.0f) * ext;
    if (noise) v = v + (noise[i * 3 + a] * 2.0f - 1.0f) * half;
    xyz[i * 3 + a] = v;
  }
}
// mark_untrained_grid: a cell no training camera sees gets -1 and never becomes occupied.  One lane per
// (cascade, cell in Morton order); the cell centre goes into every camera frame, cam = R^T (x - t); seen if z > 0,
// |x| < cx/fx * z + 2 * half and |y| < cy/fy * z + 2 * half (half = half a cell of the cascade).
__global__ void __launch_bounds__(256) k_mark_untrained(const float* __restrict__ poses, int B, float kx, float ky, int H,
                                                        int C, float bound, float* __restrict__ grid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t H3 = (int64_t)H * H * H;
  if (i >= (int64_t)C * H3) return;
  const int cas = (int)(i / H3);
  const uint32_t m = (uint32_t)(i - (int64_t)cas * H3);
  const float bnd = fminf(ldexpf(1.0f, cas), bound), half = bnd / (float)H, ext = bnd - half;
  const float wx = (2.0f * (float)compact_bits10(m) / (float)(H - 1) - 1.0f) * ext;
  const float wy = (2.0f * (float)compact_bits10(m >> 1) / (float)(H - 1) - 1.0f) * ext;
  const float wz = (2.0f * (float)compact_bits10(m >> 2) / (float)(H - 1) - 1.0f) * ext;
  bool seen = false;
  for (int b = 0; b < B && !seen; ++b) {
    const float* P = poses + (int64_t)b * 16;                       // row-major 4x4 camera-to-world
    const float dx = wx - P[3], dy = wy - P[7], dz = wz - P[11];
    const float cx = dx * P[0] + dy * P[4] + dz * P[8];
    const float cy = dx * P[1] + dy * P[5] + dz * P[9];
    const float cz = dx * P[2] + dy * P[6] + dz * P[10];
    seen = cz > 0.0f && fabsf(cx) < kx * cz + half * 2.0f && fabsf(cy) < ky * cz + half * 2.0f;
  }
  if (!seen) grid[i] = -1.0f;
}

// tmp[morton[i]] = sigma[i] * density_scale (cells visited twice keep one of the values, as upstream's index_put)
__global__ void k_occ_scatter(const float* __restrict__ sigma, const int32_t* __restrict__ morton, int64_t m,
                              float density_scale, float* __restrict__ tmp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) tmp[morton[i]] = sigma[i] * density_scale;
}
// grid = max(grid * decay, tmp) where both are >= 0 (cells marked -1 stay out; tmp -1 = not visited);
// mean_sum += sum(max(grid, 0)) in double (one atomic per workgroup).  `scale` multiplies tmp (1 after k_occ_scatter).
__global__ void __launch_bounds__(256) k_occ_update(float* __restrict__ grid, const float* __restrict__ tmp, int64_t n,
                                                    float decay, float scale, double* __restrict__ mean_sum) {
  __shared__ double red[4];
  double acc = 0.0;
  auto cell = [&](float g, float t, bool& changed) {
    t *= scale;
    if (g >= 0.0f && t >= 0.0f) { g = fmaxf(g * decay, t); changed = true; }
    acc += (double)fmaxf(g, 0.0f);
    return g;
  };
  // 16 bytes per lane and step; the launch is kept to a few hundred workgroups because each ends in ONE double atomic on
  // the same address and those run one after the other (2048 workgroups: 29 us for 24 MB, all of it that chain)
  const int64_t n4 = (((uintptr_t)grid | (uintptr_t)tmp) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<float4*>(grid)[i];
    const float4 t = reinterpret_cast<const float4*>(tmp)[i];
    bool changed = false;
    g.x = cell(g.x, t.x, changed); g.y = cell(g.y, t.y, changed); g.z = cell(g.z, t.z, changed); g.w = cell(g.w, t.w, changed);
    if (changed) reinterpret_cast<float4*>(grid)[i] = g;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    bool changed = false;
    const float g = cell(grid[i], tmp[i], changed);
    if (changed) grid[i] = g;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(mean_sum, red[0] + red[1] + red[2] + red[3]);
}
// packbits with the threshold formed on the device: thresh = min(mean, density_thresh), mean = mean_sum / n_cells
__global__ void k_packbits_mean(const float* __restrict__ grid, int64_t n_bytes, const double* __restrict__ mean_sum,
                                double inv_cells, float density_thresh, uint8_t* __restrict__ bits,
                                float* __restrict__ mean_out, const int32_t* __restrict__ counters, int n_counters,
                                int counter_stride, double* __restrict__ stats_out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float mean = (float)(*mean_sum * inv_cells);
  if (n == 0 && mean_out) *mean_out = mean;
  if (n == 0 && stats_out) {             // what the host reads back after an update, in one 16-byte copy
    int64_t total = 0;
    for (int k = 0; k < n_counters; ++k) total += counters[(int64_t)k * counter_stride];
    stats_out[0] = (double)mean;
    stats_out[1] = (double)total;
  }
  if (n >= n_bytes) return;
  const float thresh = fminf(mean, density_thresh);
  const float4 a = reinterpret_cast<const float4*>(grid)[n * 2];
  const float4 b = reinterpret_cast<const float4*>(grid)[n * 2 + 1];
  uint32_t v = (a.x > thresh) | ((a.y > thresh) << 1) | ((a.z > thresh) << 2) | ((a.w > thresh) << 3) |
               ((b.x > thresh) << 4) | ((b.y > thresh) << 5) | ((b.z > thresh) << 6) | ((b.w > thresh) << 7);
  bits[n] = (uint8_t)v;
}

// ------------------------------------------------------------------------------------------
// The marcher.  Operation order is the contract written at the top of oracle/march.py.
struct MarchParams {
  const uint8_t* bits;
  float bound, dt_gamma, dt_min, dt_max, rH;
  int C, H, H3;
};

struct Ray {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, sx, sy, sz;
};

__device__ __forceinline__ Ray load_ray(const float* __restrict__ o, const float* __restrict__ d, int64_t n) {
  Ray r;
  r.ox = o[n * 3]; r.oy = o[n * 3 + 1]; r.oz = o[n * 3 + 2];
  r.dx = d[n * 3]; r.dy = d[n * 3 + 1]; r.dz = d[n * 3 + 2];
  r.rdx = 1.0f / r.dx; r.rdy = 1.0f / r.dy; r.rdz = 1.0f / r.dz;
  r.sx = copysignf(1.0f, r.dx); r.sy = copysignf(1.0f, r.dy); r.sz = copysignf(1.0f, r.dz);
  return r;
}

__device__ __forceinline__ float step_dt(const MarchParams& P, float t) {
  return clampf(t * P.dt_gamma, P.dt_min, P.dt_max);
}
// dt_gamma == 0 (indoor scenes, the reference's 3D-FRONT rooms): t * 0 is +-0 (or NaN for an infinite t), and
// clamp(+-0 or NaN, dt_min, dt_max) is the same value for every t - fmaxf drops the NaN - so the three instructions
// per step candidate are loop-invariant.  kConst: that value, formed once by the very same expression.
template <bool kConst>
struct StepRule {
  float c;
  __device__ __forceinline__ explicit StepRule(const MarchParams& P) : c(step_dt(P, 0.0f)) {}
  __device__ __forceinline__ float operator()(const MarchParams& P, float t) const {
    if constexpr (kConst) return c;
    else return step_dt(P, t);
  }
};

// Marches from t until t >= far or max_emit samples were emitted.  emit(px,py,pz,dt,delta,t); advance() is called
// every time t moves on by one step, i.e. once per step CANDIDATE (the sequence t_0 = t, t_{i+1} = t_i + dt(t_i)
// does not depend on the grid - only which candidates are visited and emitted does).
// kSingle: one cascade (bound <= 1) - the mip level is 0 for every sample, so the two frexp, the level clamp and
// the per-iteration 1/mb are loop invariants (same values, a quarter fewer instructions per iteration).
struct MortonCalc {      // bit interleave in registers (~26 VALU instructions)
  __device__ __forceinline__ uint32_t operator()(uint32_t x, uint32_t y, uint32_t z) const { return morton3(x, y, z); }
};
struct MortonLut {       // expand_bits10 of every coordinate < H tabulated in LDS: 3 reads + 2 shift-ors, same bits
  const uint32_t* lut;
  __device__ __forceinline__ uint32_t operator()(uint32_t x, uint32_t y, uint32_t z) const {
    return lut[x] | (lut[y] << 1) | (lut[z] << 2);
  }
};

template <bool kSingle = false, bool kConstDt = false, class Emit, class Advance, class Morton = MortonCalc>
__device__ __forceinline__ int march_ray(const MarchParams& P, const Ray& r, float t, float far,
                                         int max_emit, Emit&& emit, Advance&& advance, Morton morton = Morton()) {
  int n = 0;
  float last_t = t;
  const StepRule<kConstDt> step(P);
  const float mb1 = fminf(ldexpf(1.0f, 0), P.bound), rmb1 = 1.0f / mb1;
  while (t < far && n < max_emit) {
    const float px = clampf(r.ox + t * r.dx, -P.bound, P.bound);
    const float py = clampf(r.oy + t * r.dy, -P.bound, P.bound);
    const float pz = clampf(r.oz + t * r.dz, -P.bound, P.bound);
    const float dt = step(P, t);
    int level = 0;
    float mb = mb1, rmb = rmb1;
    if constexpr (!kSingle) {
      int e0, e1;
      (void)frexpf(fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz))), &e0);
      (void)frexpf(dt * (float)P.H * 0.5f, &e1);
      level = max(clampi(e0, 0, P.C - 1), clampi(e1, 0, P.C - 1));
      mb = fminf(ldexpf(1.0f, level), P.bound);
      rmb = 1.0f / mb;
    }
    const int nx = clampi((int)(((px * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const int ny = clampi((int)(((py * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const int nz = clampi((int)(((pz * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const uint32_t idx = (uint32_t)level * (uint32_t)P.H3 + morton((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const bool occ = (P.bits[idx >> 3] >> (idx & 7)) & 1;
    if (occ) {
      const float tn = t + dt;
      emit(px, py, pz, dt, tn - last_t, t);
      t = tn;
      advance();
      last_t = tn;
      ++n;
    } else {
      const float ax = ((float)nx + 0.5f) + 0.5f * r.sx;
      const float ay = ((float)ny + 0.5f) + 0.5f * r.sy;
      const float az = ((float)nz + 0.5f) + 0.5f * r.sz;
      const float cx = (((ax * P.rH) * 2.0f - 1.0f) * mb - px) * r.rdx;
      const float cy = (((ay * P.rH) * 2.0f - 1.0f) * mb - py) * r.rdy;
      const float cz = (((az * P.rH) * 2.0f - 1.0f) * mb - pz) * r.rdz;
      const float tt = t + fmaxf(0.0f, fminf(cx, fminf(cy, cz)));
      do {
        t = t + step(P, t);
        advance();
      } while (t < tt);
    }
  }
  return n;
}
template <class Emit>
__device__ __forceinline__ int march_ray(const MarchParams& P, const Ray& r, float t, float far,
                                         int max_emit, Emit&& emit) {
  return P.dt_gamma == 0.0f ? march_ray<false, true>(P, r, t, far, max_emit, emit, [] {})
                            : march_ray<false, false>(P, r, t, far, max_emit, emit, [] {});
}
// run-time dispatch on the cascade count (wave-uniform)
template <class Emit, class Advance, class Morton>
__device__ __forceinline__ int march_ray_any(const MarchParams& P, const Ray& r, float t, float far,
                                             int max_emit, Emit&& emit, Advance&& advance, Morton morton) {
  if (P.dt_gamma == 0.0f)
    return P.C == 1 ? march_ray<true, true>(P, r, t, far, max_emit, emit, advance, morton)
                    : march_ray<false, true>(P, r, t, far, max_emit, emit, advance, morton);
  return P.C == 1 ? march_ray<true, false>(P, r, t, far, max_emit, emit, advance, morton)
                  : march_ray<false, false>(P, r, t, far, max_emit, emit, advance, morton);
}

__device__ __forceinline__ float start_t(const MarchParams& P, float near, float noise) {
  return near + step_dt(P, near) * noise;
}

// pass 1: per-ray sample counts + per-block sums
__global__ void __launch_bounds__(kRayBlock) k_march_count(MarchParams P, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, int64_t N,
                                                           int max_steps, const float* __restrict__ nears,
                                                           const float* __restrict__ fars,
                                                           const float* __restrict__ noises,
                                                           int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ block_sums,
                                                           uint32_t* __restrict__ mask, int cap_words) {
  __shared__ int32_t wsum[kRayBlock / 64];
  // this kernel is VALU-bound (2.3e8 wave instructions per 800x800 frame, every SIMD issuing all the time), and a
  // quarter of its loop was the Morton bit interleave: tabulate expand_bits10 once per workgroup
  __shared__ uint32_t lut[1024];
  for (int i = threadIdx.x; i < P.H; i += kRayBlock) lut[i] = expand_bits10((uint32_t)i);
  __syncthreads();
  const MortonLut morton{lut};
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int cnt = 0;
  if (n < N) {
    const Ray r = load_ray(rays_o, rays_d, n);
    const float t0 = start_t(P, nears[n], noises ? noises[n] : 0.0f);
    if (cap_words > 0) {
      // Record WHICH step candidates were emitted, one bit per candidate, words laid out [word][ray]: the write
      // pass regenerates the candidate sequence with the same additions and emits at the set bits - no second
      // walk through the occupancy grid.  A ray that runs past the last word is flagged and marched again.
      uint32_t word = 0;
      int widx = 0, cand = 0;
      bool ok = true;
      cnt = march_ray_any(P, r, t0, fars[n], max_steps,
                      [&](float, float, float, float, float, float) {
                        const int w = cand >> 5;
                        if (w < cap_words) {
                          while (widx < w) {
                            mask[(int64_t)widx * N + n] = word;
                            word = 0;
                            ++widx;
                          }
                          word |= 1u << (cand & 31);
                        } else {
                          ok = false;
                        }
                      },
                      [&] { ++cand; }, morton);
      if (widx < cap_words) mask[(int64_t)widx * N + n] = word;
      mask[(int64_t)cap_words * N + n] = ok ? 1u : 0u;
    } else {
      cnt = march_ray_any(P, r, t0, fars[n], max_steps, [](float, float, float, float, float, float) {}, [] {}, morton);
    }
    counts[n] = cnt;
  }
  const int incl = wave_inclusive_scan(cnt);
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
#pragma unroll
    for (int w = 0; w < kRayBlock / 64; ++w) s += wsum[w];
    block_sums[blockIdx.x] = s;
  }
}

// pass 2: one workgroup turns block_sums into exclusive prefixes; counter = {total, N}
__global__ void __launch_bounds__(1024) k_scan_block_sums(int32_t* __restrict__ block_sums, int n_blocks,
                                                          int64_t N, int32_t* __restrict__ counter) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? block_sums[i] : 0;
    const int incl = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += wsum[w];
    const int carry = carry_s;
    if (i < n_blocks) block_sums[i] = carry + wave_off + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + wave_off + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counter[0] = carry_s;
    counter[1] = (int32_t)N;
  }
}

// pass 3: rays[n] = (n, offset, count)
__global__ void __launch_bounds__(kRayBlock) k_finalize_offsets(const int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ block_prefix,
                                                                int64_t N, int32_t* __restrict__ rays) {
  __shared__ int32_t wsum[kRayBlock / 64];
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cnt = n < N ? counts[n] : 0;
  const int incl = wave_inclusive_scan(cnt);
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int off = block_prefix[blockIdx.x] + incl - cnt;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
  if (n < N) {
    rays[n * 3 + 0] = (int32_t)n;
    rays[n * 3 + 1] = off;
    rays[n * 3 + 2] = cnt;
  }
}

// Replays a ray from its candidate bit mask: emit(px,py,pz,dt,delta,t) with exactly the values the march produced
// (same t accumulation, same clamp(o + t*d) and step_dt(t) operations).
template <bool kConstDt, class Emit>
__device__ __forceinline__ void replay_ray_t(const MarchParams& P, const Ray& r, const uint32_t* __restrict__ mask, int64_t N,
                                             int64_t n, float t, int cnt, Emit&& emit) {
  const StepRule<kConstDt> step(P);
  // Candidate by candidate, all lanes in lockstep.  (Looping over the SAMPLES instead - every lane running its own
  // tight loop of additions up to its next set bit, the wave executing the emit body once per sample index - was
  // measured in round 2: 528 vs 332 us per frame.  The gaps between set bits do not line up across the rays of a
  // wave, and a divergent inner loop costs more than the emit bodies it saves.)
  // Eight candidates per trip with constant bit positions (the loop control of a one-candidate trip was a dozen scalar
  // instructions around seven vector ones), and a stretch in which NO ray of the wave has a sample - the rays of a wave
  // are neighbours, their empty stretches mostly coincide - is eight bare additions.  A ray has exactly cnt set bits,
  // so the zero bits behind its last sample only move a t nobody reads.
  float last_t = t;
  int emitted = 0;
  for (int w = 0; emitted < cnt; ++w) {
    const uint32_t bits = mask[(int64_t)w * N + n];
    for (int b0 = 0; b0 < 32; b0 += 8) {
      const uint32_t sub = (bits >> b0) & 0xFFu;
      if (__ballot(sub != 0u) == 0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t = t + step(P, t);
        continue;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dt = step(P, t);
        const float tn = t + dt;
        if (sub & (1u << j)) {
          const float px = clampf(r.ox + t * r.dx, -P.bound, P.bound);
          const float py = clampf(r.oy + t * r.dy, -P.bound, P.bound);
          const float pz = clampf(r.oz + t * r.dz, -P.bound, P.bound);
          emit(px, py, pz, dt, tn - last_t, t);
          last_t = tn;
          ++emitted;
        }
        t = tn;
      }
    }
  }
}
template <class Emit>
__device__ __forceinline__ void replay_ray(const MarchParams& P, const Ray& r, const uint32_t* __restrict__ mask, int64_t N,
                                           int64_t n, float t, int cnt, Emit&& emit) {
  if (P.dt_gamma == 0.0f) replay_ray_t<true>(P, r, mask, N, n, t, cnt, emit);     // wave-uniform
  else replay_ray_t<false>(P, r, mask, N, n, t, cnt, emit);
}

// pass 4: write the samples into their slots (replay of the recorded samples, or a second march for rays
// longer than the capture row)
__global__ void __launch_bounds__(kRayBlock) k_march_write(MarchParams P, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, int64_t N,
                                                           int64_t M, const float* __restrict__ nears,
                                                           const float* __restrict__ fars,
                                                           const float* __restrict__ noises,
                                                           const int32_t* __restrict__ rays,
                                                           float* __restrict__ xyzs, float* __restrict__ dirs,
                                                           float* __restrict__ deltas,
                                                           const uint32_t* __restrict__ mask, int cap_words) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int off = rays[n * 3 + 1];
  const int cnt = rays[n * 3 + 2];
  if (cnt == 0 || (int64_t)off + cnt > M) return;
  const Ray r = load_ray(rays_o, rays_d, n);
  int64_t i = off;
  auto emit = [&](float px, float py, float pz, float dt, float delta, float) {
    xyzs[i * 3 + 0] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
    dirs[i * 3 + 0] = r.dx; dirs[i * 3 + 1] = r.dy; dirs[i * 3 + 2] = r.dz;
    deltas[i * 2 + 0] = dt; deltas[i * 2 + 1] = delta;
    ++i;
  };
  const float t0 = start_t(P, nears[n], noises ? noises[n] : 0.0f);
  if (cap_words > 0 && mask[(int64_t)cap_words * N + n]) {
    replay_ray(P, r, mask, N, n, t0, cnt, emit);
  } else {
    march_ray(P, r, t0, fars[n], cnt, emit);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-per-ray marcher for SMALL batches (training: 4096 rays).  With one thread per ray the march is a chain of
// ~600 dependent bit tests per ray and 4096 threads leave the machine empty: 0.19 ms per pass, all latency.
// The candidate sequence c_0 = t, c_{j+1} = c_j + dt(c_j) does not depend on the occupancy grid, so a wave
// evaluates 64 consecutive candidates at once (position, level, cell, bit test, exit distance of a miss - the
// very expressions of march_ray) and then resolves which of them the serial walk visits: a hit goes to the next
// candidate, a miss to the first candidate not below its exit distance.  Same samples, same bits as march_ray.
// emit(k, px, py, pz, dt, delta, t) runs in the lanes that own a sample; k is its index along the ray.
template <bool kSingle = false, class Emit>
__device__ __forceinline__ int march_ray_coop(const MarchParams& P, const Ray& r, float t, float far, int max_emit,
                                              Emit&& emit) {
  const float mb1 = fminf(ldexpf(1.0f, 0), P.bound), rmb1 = 1.0f / mb1;      // kSingle: see march_ray
  const int lane = threadIdx.x & 63;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int n = 0;
  float last_t = t;
  while (t < far && n < max_emit) {                       // wave-uniform
    // candidate of this lane: `lane` sequential additions, exactly the serial accumulation
    float c = t;
    if (P.dt_gamma == 0.0f) {
      // constant step (indoor scenes): clamp(c * 0) is dt_min for every c.  The chain runs along the wave: each
      // wave_shr:1 addition gives lane j (j >= 1) the value of lane j-1 plus dt0 and leaves lane 0 (no source lane:
      // disabled) at t, so after 63 of them lane j holds t after j sequential additions - one VALU instruction per
      // addition where compare + select + add took three (the kernel is issue bound: 4 waves per SIMD, and 73 % of
      // its vector instructions were this chain; profiles/r03_NOTES.txt 20).  All 64 lanes are active here (the
      // callers branch per wave); s_nop 1 covers the two wait states a DPP read needs after a VALU write.
      const float dt0 = step_dt(P, t);
      asm volatile(
          ".rept 63\n"
          "s_nop 1\n"
          "v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
          ".endr\n"
          : "+v"(c)
          : "v"(dt0));
    } else {
      for (int j = 0; j < 63; ++j) c = j < lane ? c + step_dt(P, c) : c;
    }
    const bool in = c < far;
    const float px = clampf(r.ox + c * r.dx, -P.bound, P.bound);
    const float py = clampf(r.oy + c * r.dy, -P.bound, P.bound);
    const float pz = clampf(r.oz + c * r.dz, -P.bound, P.bound);
    const float dt = step_dt(P, c);
    const float tn = c + dt;
    int level = 0;
    float mb = mb1, rmb = rmb1;
    if constexpr (!kSingle) {
      int e0, e1;
      (void)frexpf(fmaxf(fabsf(px), fmaxf(fabsf(py), fabsf(pz))), &e0);
      (void)frexpf(dt * (float)P.H * 0.5f, &e1);
      level = max(clampi(e0, 0, P.C - 1), clampi(e1, 0, P.C - 1));
      mb = fminf(ldexpf(1.0f, level), P.bound);
      rmb = 1.0f / mb;
    }
    const int nx = clampi((int)(((px * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const int ny = clampi((int)(((py * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const int nz = clampi((int)(((pz * rmb + 1.0f) * 0.5f) * (float)P.H), 0, P.H - 1);
    const uint32_t idx = (uint32_t)level * (uint32_t)P.H3 + morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const bool occ = (P.bits[idx >> 3] >> (idx & 7)) & 1;
    const float ax = ((float)nx + 0.5f) + 0.5f * r.sx;
    const float ay = ((float)ny + 0.5f) + 0.5f * r.sy;
    const float az = ((float)nz + 0.5f) + 0.5f * r.sz;
    const float cx = (((ax * P.rH) * 2.0f - 1.0f) * mb - px) * r.rdx;
    const float cy = (((ay * P.rH) * 2.0f - 1.0f) * mb - py) * r.rdy;
    const float cz = (((az * P.rH) * 2.0f - 1.0f) * mb - pz) * r.rdz;
    const float tt = c + fmaxf(0.0f, fminf(cx, fminf(cy, cz)));
    // a miss continues at the first later candidate that is not below tt (do { t += dt } while (t < tt));
    // candidates increase with the lane, so a binary search over the lanes finds it (64 = beyond this window)
    int lo = lane + 1, hi = 64;
    bool settled_all = false;
    if (P.dt_gamma == 0.0f) {
      // constant step: the candidates are c + k * dt up to rounding (far below one step over 64 additions), so
      // the landing lane is lane + floor((tt - c) / dt) or the one after it.  Three independent lane reads check
      // that guess against the very predicate of the search (c_i < tt); any lane it does not settle sends the
      // wave through the search below - the estimate only ever saves time, it never decides a sample.
      const int i0 = lane + (int)fminf((tt - c) * (1.0f / dt), 64.0f);          // lane .. lane + 64
      const float ca = __shfl(c, clampi(i0 - 1, 0, 63), 64);
      const float cb = __shfl(c, min(i0, 63), 64);
      const float cc = __shfl(c, min(i0 + 1, 63), 64);
      const bool lt_a = i0 - 1 <= lane || ca < tt;
      const bool ge_b = i0 >= 64 || !(cb < tt);
      const bool ge_c = i0 + 1 >= 64 || !(cc < tt);
      const bool take_b = i0 > lane && ge_b;
      const bool settled = take_b ? lt_a : ge_c;
      lo = take_b ? min(i0, 64) : i0 + 1;
      settled_all = __ballot(in && !occ && !settled) == 0;
    }
    if (!settled_all) {                                    // wave-uniform
      lo = lane + 1;
#pragma unroll
      for (int it = 0; it < 7; ++it) {
        const int mid = (lo + hi) >> 1;
        const float cm = __shfl(c, min(mid, 63), 64);
        const bool below = mid < 64 && cm < tt;
        if (lo < hi) {
          if (below) lo = mid + 1; else hi = mid;
        }
      }
    }
    const int next = lo;
    const uint64_t in_mask = __ballot(in);
    const uint64_t miss_mask = __ballot(in && !occ);
    const int n_in = __popcll(in_mask);                   // the lanes below `far` are a prefix
    // serial walk over the window: runs of hits are taken whole, every visited miss jumps
    uint64_t visited = 0;
    int v = 0, last_miss = -1;
    while (v < n_in) {
      const uint64_t rest = miss_mask >> v;
      const int m = rest ? v + __builtin_ctzll(rest) : 64;          // next miss at or after v
      const int stop = min(m, n_in);
      if (stop > v) visited |= (stop - v >= 64 ? ~0ull : ((1ull << (stop - v)) - 1ull)) << v;
      if (m >= n_in) { v = stop; last_miss = -1; break; }
      visited |= 1ull << m;
      last_miss = m;
      v = __builtin_amdgcn_readlane(next, m);
    }
    uint64_t emit_mask = visited & in_mask & ~miss_mask;
    int k_new = __popcll(emit_mask);
    bool full = false;
    if (n + k_new >= max_emit) {                          // keep the first max_emit - n samples, the ray ends there
      full = true;
      const int keep = max_emit - n;
      uint64_t m2 = emit_mask;
      for (int i = 0; i < keep; ++i) m2 &= m2 - 1;                  // clear the `keep` lowest set bits
      emit_mask &= ~m2;
      k_new = keep;
    }
    const uint64_t before = emit_mask & lt_mask;
    const int prev = before ? 63 - __builtin_clzll(before) : 0;
    const float prev_tn = __shfl(tn, prev, 64);            // executed by all lanes
    if ((emit_mask >> lane) & 1ull) {
      const float last = before ? prev_tn : last_t;
      emit(n + __popcll(before), px, py, pz, dt, tn - last, c);
    }
    if (emit_mask) last_t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tn),
                                                                                63 - __builtin_clzll(emit_mask)));
    n += k_new;
    if (full) break;
    if (v < 64) {                                         // reached a candidate at or beyond `far` inside the window
      t = far;
    } else {
      // the walk left the window: continue from c_64, and finish the pending skip of the last visited miss
      const float c63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tn), 63));
      t = c63;                                            // c_63 + dt(c_63)
      if (last_miss >= 0) {
        const float tt_m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), last_miss));
        while (t < tt_m) t = t + step_dt(P, t);
      }
    }
  }
  return n;
}

constexpr int kCoopRaysPerBlock = 4;

__global__ void __launch_bounds__(kCoopRaysPerBlock * 64) k_march_count_coop(MarchParams P, const float* __restrict__ rays_o,
                                                                             const float* __restrict__ rays_d, int64_t N,
                                                                             int max_steps, const float* __restrict__ nears,
                                                                             const float* __restrict__ fars,
                                                                             const float* __restrict__ noises,
                                                                             int32_t* __restrict__ counts,
                                                                             float* __restrict__ stage, int stage_pitch) {
  const int64_t n = (int64_t)blockIdx.x * kCoopRaysPerBlock + (threadIdx.x >> 6);
  if (n >= N) return;
  const Ray r = load_ray(rays_o, rays_d, n);
  const float t0 = start_t(P, nears[n], noises ? noises[n] : 0.0f);
  int cnt;
  if (stage) {
    // Staged: the samples of the ONE walk are parked in the workspace, ray n at row n * stage_pitch (pitch >=
    // max_steps: every ray fits; only the rows that exist are written, the rest of the reservation is never touched),
    // and the write pass is a copy to the scanned offsets instead of a second walk - a ray is a chain of dependent
    // windows, so walking it twice cost twice the latency (45 + 45 us for 4096 rays; the copy takes ~5).
    float* row = stage + (size_t)n * stage_pitch * 5;
    auto park = [&](int k, float px, float py, float pz, float dt, float delta, float) {
      float* q = row + (size_t)k * 5;
      q[0] = px; q[1] = py; q[2] = pz; q[3] = dt; q[4] = delta;
    };
    cnt = P.C == 1 ? march_ray_coop<true>(P, r, t0, fars[n], max_steps, park)
                   : march_ray_coop<false>(P, r, t0, fars[n], max_steps, park);
  } else {
    auto none = [](int, float, float, float, float, float, float) {};
    cnt = P.C == 1 ? march_ray_coop<true>(P, r, t0, fars[n], max_steps, none)
                   : march_ray_coop<false>(P, r, t0, fars[n], max_steps, none);
  }
  if ((threadIdx.x & 63) == 0) counts[n] = cnt;
}

// write pass of the staged marcher: one wave per ray copies its parked samples to their slots
__global__ void __launch_bounds__(kCoopRaysPerBlock * 64) k_march_write_staged(const float* __restrict__ rays_d, int64_t N,
                                                                               int64_t M, const int32_t* __restrict__ rays,
                                                                               const float* __restrict__ stage, int stage_pitch,
                                                                               float* __restrict__ xyzs, float* __restrict__ dirs,
                                                                               float* __restrict__ deltas) {
  const int64_t n = (int64_t)blockIdx.x * kCoopRaysPerBlock + (threadIdx.x >> 6);
  if (n >= N) return;
  const int off = rays[n * 3 + 1];
  const int cnt = rays[n * 3 + 2];
  const int lane = threadIdx.x & 63;
  // Rows nobody owns are zero-filled HERE (the caller hands over uninitialised buffers: one fill launch less per
  // step): the tail behind the last ray's samples is shared out over all waves, and the one dropped ray that starts
  // inside the buffer (offset < M < offset + count; every later ray starts beyond M) clears from its offset on.
  auto zero_rows = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo + lane; i < hi; i += 64) {
      xyzs[i * 3 + 0] = 0.f; xyzs[i * 3 + 1] = 0.f; xyzs[i * 3 + 2] = 0.f;
      dirs[i * 3 + 0] = 0.f; dirs[i * 3 + 1] = 0.f; dirs[i * 3 + 2] = 0.f;
      deltas[i * 2 + 0] = 0.f; deltas[i * 2 + 1] = 0.f;
    }
  };
  const int64_t total = (int64_t)rays[(N - 1) * 3 + 1] + rays[(N - 1) * 3 + 2];
  if (total < M) {
    const int64_t per = (M - total + N - 1) / N;
    zero_rows(total + n * per, min(M, total + (n + 1) * per));
  }
  if ((int64_t)off + cnt > M) {
    if (cnt > 0 && off < M) zero_rows(off, M);
    return;
  }
  if (cnt == 0) return;
  const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
  const float* row = stage + (size_t)n * stage_pitch * 5;
  for (int k = lane; k < cnt; k += 64) {
    const float* q = row + (size_t)k * 5;
    const int64_t i = (int64_t)off + k;
    xyzs[i * 3 + 0] = q[0]; xyzs[i * 3 + 1] = q[1]; xyzs[i * 3 + 2] = q[2];
    dirs[i * 3 + 0] = dx; dirs[i * 3 + 1] = dy; dirs[i * 3 + 2] = dz;
    deltas[i * 2 + 0] = q[3]; deltas[i * 2 + 1] = q[4];
  }
}

// ---- steady-state cell choice of the occupancy update ---------------------------------------------------
// upstream: N/4 cells with uniformly random coordinates + N/4 picks, with replacement, among the occupied cells
// (nonzero(grid > 0)[randint]).  With tensor ops that is ~30 small launches (randint, morton, compare, cumsum, scatter,
// gather, cat ...) and the picks arrive in random order, so the field kernel that evaluates them afterwards runs at
// half its rate (every 16-sample tile scattered over the volume: 0.91 ms for 1 M points against 0.93 ms for the 2.1 M
// of a Morton-ordered full sweep).  Here: three launches, and the picks come out grouped by kOccBuckets slices of the
// Morton range WITHOUT being sorted or moved - an i.i.d. uniform sample is the same as (a) multinomial counts per
// slice, (b) uniform positions inside each slice, independently: (a) is the histogram of one set of uniform draws,
// (b) uses a second set.  Output slot j then belongs to the slice whose scanned count range contains j.
//   uniform half:   cell = floor((b + u2) * n_cells / kOccBuckets)                 (a Morton index IS a uniform cell)
//   occupied half:  rank = floor((b + u2) * n_occ / kOccBuckets); cell = rank-th cell with grid > 0 in Morton order,
//                   found through per-1024-cell counts (binary search) and the 64-cell ballot masks of pass 1.
constexpr int kOccBuckets = 4096;
constexpr int kOccBlockCells = 1024;
constexpr int kOccMaxBlocks = 16384;                   // 256^3 cells
__global__ void __launch_bounds__(kOccBlockCells) k_occ_count(const float* __restrict__ grid, int64_t n_cells,
                                                              uint64_t* __restrict__ masks, int32_t* __restrict__ block_counts,
                                                              int32_t* __restrict__ hist) {
  __shared__ int32_t wcount[kOccBlockCells / 64];
  const int64_t i = (int64_t)blockIdx.x * kOccBlockCells + threadIdx.x;
  const uint64_t m = __ballot(i < n_cells && grid[i] > 0.0f);
  if ((threadIdx.x & 63) == 0) {
    masks[i >> 6] = m;
    wcount[threadIdx.x >> 6] = __popcll(m);
  }
  for (int64_t k = i; k < 2 * kOccBuckets; k += (int64_t)gridDim.x * kOccBlockCells) hist[k] = 0;   // for pass 2
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < kOccBlockCells / 64; ++w) c += wcount[w];
    block_counts[blockIdx.x] = c;
  }
}
// u [2n]: slice draws of the uniform half, then of the occupied half
__global__ void __launch_bounds__(1024) k_occ_hist(const float* __restrict__ u, int64_t n, int32_t* __restrict__ hist) {
  __shared__ int32_t h[2 * kOccBuckets];
  for (int k = threadIdx.x; k < 2 * kOccBuckets; k += 1024) h[k] = 0;
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x; j < 2 * n; j += (int64_t)gridDim.x * 1024) {
    const int b = min((int)(u[j] * (float)kOccBuckets), kOccBuckets - 1);
    atomicAdd(&h[(j >= n ? kOccBuckets : 0) + b], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * kOccBuckets; k += 1024)
    if (h[k]) atomicAdd(&hist[k], h[k]);
}
// exclusive scan of a[0..n) in LDS, in place; a[n] = total.  1024 threads.
__device__ void lds_exclusive_scan_1024(int32_t* a, int n, int32_t* wsum) {
  const int per = (n + 1023) / 1024;
  const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  const int incl = wave_inclusive_scan(s);
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int run = incl - s;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
  for (int i = lo; i < hi; ++i) { const int t = a[i]; a[i] = run; run += t; }
  if (threadIdx.x == 1023) a[n] = run;
  __syncthreads();
}
__device__ __forceinline__ int upper_slot(const int32_t* scanned, int n, int r) {   // scanned[b] <= r < scanned[b + 1]
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (scanned[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ void __launch_bounds__(1024) k_occ_pick(const float* __restrict__ u2, int64_t n, const int32_t* __restrict__ hist,
                                                   const uint64_t* __restrict__ masks,
                                                   const int32_t* __restrict__ block_counts, int n_blocks, int64_t n_cells,
                                                   int32_t* __restrict__ out) {
  extern __shared__ int32_t occ_lds[];
  int32_t* hs = occ_lds;                               // [2][kOccBuckets + 1]
  int32_t* bs = hs + 2 * (kOccBuckets + 1);            // [n_blocks + 1]
  __shared__ int32_t wsum[16];
  for (int k = threadIdx.x; k < kOccBuckets; k += 1024) { hs[k] = hist[k]; hs[kOccBuckets + 1 + k] = hist[kOccBuckets + k]; }
  for (int k = threadIdx.x; k < n_blocks; k += 1024) bs[k] = block_counts[k];
  __syncthreads();
  lds_exclusive_scan_1024(hs, kOccBuckets, wsum);
  lds_exclusive_scan_1024(hs + kOccBuckets + 1, kOccBuckets, wsum);
  lds_exclusive_scan_1024(bs, n_blocks, wsum);
  const int64_t n_occ = bs[n_blocks];
  for (int64_t j = (int64_t)blockIdx.x * 1024 + threadIdx.x; j < 2 * n; j += (int64_t)gridDim.x * 1024) {
    const bool occupied = j >= n;
    const int32_t* scanned = hs + (occupied ? kOccBuckets + 1 : 0);
    const int b = upper_slot(scanned, kOccBuckets, (int)(occupied ? j - n : j));
    const double v = ((double)b + (double)u2[j]) / (double)kOccBuckets;
    int64_t cell;
    if (!occupied) {
      cell = min((int64_t)(v * (double)n_cells), n_cells - 1);
    } else if (n_occ == 0) {
      cell = 0;                                        // nothing occupied: one more visit of cell 0
    } else {
      const int rank = (int)min((int64_t)(v * (double)n_occ), n_occ - 1);
      const int blk = upper_slot(bs, n_blocks, rank);
      int rem = rank - bs[blk];
      const uint64_t* mk = masks + (int64_t)blk * (kOccBlockCells / 64);
      int w = 0;
      uint64_t m = mk[0];
      for (int c = __popcll(m); rem >= c; c = __popcll(m)) { rem -= c; m = mk[++w]; }
      for (int k = 0; k < rem; ++k) m &= m - 1;        // drop the rem lowest set bits
      cell = (int64_t)blk * kOccBlockCells + w * 64 + __ffsll((unsigned long long)m) - 1;
    }
    out[j] = (int32_t)cell;
  }
}

// one workgroup: rays[n] = (n, exclusive scan of counts in ray order, count); counter = {total, N}
__global__ void __launch_bounds__(1024) k_scan_counts(const int32_t* __restrict__ counts, int64_t N,
                                                      int32_t* __restrict__ rays, int32_t* __restrict__ counter) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < N; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = i < N ? counts[i] : 0;
    const int incl = wave_inclusive_scan(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int off = carry_s + incl - v;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    if (i < N) {
      rays[i * 3 + 0] = (int32_t)i;
      rays[i * 3 + 1] = off;
      rays[i * 3 + 2] = v;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + v;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    counter[0] = carry_s;
    counter[1] = (int32_t)N;
  }
}

__global__ void __launch_bounds__(kCoopRaysPerBlock * 64) k_march_write_coop(MarchParams P, const float* __restrict__ rays_o,
                                                                             const float* __restrict__ rays_d, int64_t N,
                                                                             int64_t M, const float* __restrict__ nears,
                                                                             const float* __restrict__ fars,
                                                                             const float* __restrict__ noises,
                                                                             const int32_t* __restrict__ rays,
                                                                             float* __restrict__ xyzs, float* __restrict__ dirs,
                                                                             float* __restrict__ deltas) {
  const int64_t n = (int64_t)blockIdx.x * kCoopRaysPerBlock + (threadIdx.x >> 6);
  if (n >= N) return;
  const int off = rays[n * 3 + 1];
  const int cnt = rays[n * 3 + 2];
  if (cnt == 0 || (int64_t)off + cnt > M) return;
  const Ray r = load_ray(rays_o, rays_d, n);
  const float t0 = start_t(P, nears[n], noises ? noises[n] : 0.0f);
  auto put = [&](int k, float px, float py, float pz, float dt, float delta, float) {
    const int64_t i = (int64_t)off + k;
    xyzs[i * 3 + 0] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
    dirs[i * 3 + 0] = r.dx; dirs[i * 3 + 1] = r.dy; dirs[i * 3 + 2] = r.dz;
    deltas[i * 2 + 0] = dt; deltas[i * 2 + 1] = delta;
  };
  if (P.C == 1) march_ray_coop<true>(P, r, t0, fars[n], cnt, put);
  else march_ray_coop<false>(P, r, t0, fars[n], cnt, put);
}

// ---- mask-supervised loss of the instance stage: cross entropy of the rendered logits against the
// matched-mask ids, ignore_index rows skipped, mean over the kept rows - forward value AND d loss / d logits in two
// launches (torch needs log_softmax, nll_loss and their two backward kernels plus fills).  One wave per row
// (K <= 64 classes, lane = class), at most kCeBlocks workgroups; every workgroup leaves (sum of its row losses, its
// kept rows) in part[block] - NO atomics: a first version added every wave's sum to one address and took 96 us for
// 4096 rows, same-address float atomics from different CUs serialise at ~12 ns each.
constexpr int kCeBlocks = 64;
__global__ void __launch_bounds__(256) k_ce_rows(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                 int64_t N, int K, int64_t ignore_index, float* __restrict__ dlogits,
                                                 float2* __restrict__ part) {
  __shared__ float2 red[4];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  float loss_sum = 0.f, kept = 0.f, bad = 0.f;     // bad: labels that are neither ignore_index nor a class
  for (int64_t r = wave; r < N; r += n_waves) {
    const int64_t y = labels[r];
    const bool keep = y != ignore_index && y >= 0 && y < K;
    if (y != ignore_index && !(y >= 0 && y < K)) bad += 1.f;
    const float x = lane < K ? logits[r * K + lane] : -INFINITY;
    float m = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const float e = lane < K ? expf(x - m) : 0.f;
    float ssum = e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) ssum += __shfl_xor(ssum, d, 64);
    const float xy = __shfl(x, keep ? (int)y : 0, 64);
    if (keep) {
      loss_sum += (m + logf(ssum)) - xy;
      kept += 1.f;
    }
    if (lane < K) dlogits[r * K + lane] = keep ? e / ssum - (lane == (int)y ? 1.f : 0.f) : 0.f;
  }
  // a label that is neither ignore_index nor a class poisons the loss (NaN): torch's cross_entropy asserts on the device
  // for such a label; dropping the row silently would train on fewer rows without anyone noticing
  if (bad > 0.f) loss_sum = NAN;
  if (lane == 0) red[threadIdx.x >> 6] = make_float2(loss_sum, kept);
  __syncthreads();
  if (threadIdx.x == 0)
    part[blockIdx.x] = make_float2(red[0].x + red[1].x + red[2].x + red[3].x, red[0].y + red[1].y + red[2].y + red[3].y);
}
// dlogits *= 1 / kept; loss = sum / kept (NaN when nothing is kept, as torch's mean over an empty set)
__global__ void __launch_bounds__(256) k_ce_scale(float* __restrict__ dlogits, int64_t n, const float2* __restrict__ part,
                                                  int n_part, float* __restrict__ loss) {
  float sum = 0.f, kept = 0.f;
  for (int i = 0; i < n_part; ++i) {               // <= 64 pairs, the same order in every thread
    const float2 p = part[i];
    sum += p.x;
    kept += p.y;
  }
  const float inv = kept > 0.f ? 1.0f / kept : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dlogits[i] *= inv;
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss = sum / kept;
}

// tail of run_cuda (a14): image += (1 - weights_sum) * bg, depth = clamp(depth - near, 0) / (far - near); one launch
// instead of seven elementwise ones (no-grad use only).  t0 (nullable): start parameter of every ray.  Upstream's
// TRAINING compositing accumulates depth over t counted from the ray's first step (composite_rays_train: t = 0,
// t += delta), its INFERENCE compositing over the absolute ray parameter (composite_rays: t = rays_t, t += delta);
// the one-pass inference kernels here count like the training kernel, so inference passes t0 and the depth becomes
// sum w * (t0 + t_rel) = depth + t0 * weights_sum - upstream's inference value.
__global__ void __launch_bounds__(256) k_finish_rays(const float* image, const float* depth,
                                                     const float* __restrict__ weights_sum,
                                                     const float* __restrict__ nears, const float* __restrict__ fars,
                                                     const float* __restrict__ t0, float bg_r, float bg_g, float bg_b,
                                                     int64_t N, float* image_out, float* depth_out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float ws = weights_sum[n];
  const float a = 1.0f - ws;
  image_out[n * 3 + 0] = image[n * 3 + 0] + a * bg_r;
  image_out[n * 3 + 1] = image[n * 3 + 1] + a * bg_g;
  image_out[n * 3 + 2] = image[n * 3 + 2] + a * bg_b;
  float d = depth[n];
  if (t0) d = d + t0[n] * ws;
  depth_out[n] = fmaxf(d - nears[n], 0.0f) / (fars[n] - nears[n]);
}

// Training tail of the NeRF stage in ONE launch: background blend + depth normalisation (k_finish_rays) + the mean squared
// error against the batch's pixels + its gradients.  Under autograd the same tail is ~18 element-wise / reduce launches
// of 2-5 us each between the compositing forward and backward (profiles/r03_NOTES.txt section 11).  One workgroup: a
// training batch is a few thousand rays, and a single workgroup sums in a fixed order (deterministic loss).
//   image_out = image + (1 - ws) * bg          bg: one colour, or bg_rays [N,3] (upstream's random background for RGBA)
//   loss      = mean over the 3N channels of (image_out - target)^2
//   grad[0, 3N)  = d loss / d image  = 2 (image_out - target) / 3N
//   grad[3N, 4N) = d loss / d ws     = -sum_c grad_image_c * bg_c
constexpr int kFinishMseThreads = 1024;
__global__ void __launch_bounds__(kFinishMseThreads) k_finish_rays_mse(
    const float* __restrict__ image, const float* __restrict__ depth, const float* __restrict__ weights_sum,
    const float* __restrict__ nears, const float* __restrict__ fars, float bg_r, float bg_g, float bg_b,
    const float* __restrict__ bg_rays, const float* __restrict__ target, int64_t N, float* __restrict__ image_out,
    float* __restrict__ depth_out, float* __restrict__ grad, float* __restrict__ loss) {
  __shared__ float part[kFinishMseThreads / 64];
  const float inv = 1.0f / (3.0f * (float)N);
  float acc = 0.0f;
  for (int64_t n = threadIdx.x; n < N; n += kFinishMseThreads) {
    const float ws = weights_sum[n];
    const float a = 1.0f - ws;
    float bg[3] = {bg_r, bg_g, bg_b};
    if (bg_rays) { bg[0] = bg_rays[n * 3 + 0]; bg[1] = bg_rays[n * 3 + 1]; bg[2] = bg_rays[n * 3 + 2]; }
    float gws = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float px = image[n * 3 + c] + a * bg[c];
      const float e = px - target[n * 3 + c];
      const float g = 2.0f * e * inv;
      image_out[n * 3 + c] = px;
      grad[n * 3 + c] = g;
      gws -= g * bg[c];
      acc += e * e;
    }
    grad[3 * N + n] = gws;
    depth_out[n] = fmaxf(depth[n] - nears[n], 0.0f) / (fars[n] - nears[n]);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < kFinishMseThreads / 64; ++w) t += part[w];
    loss[0] = t * inv;
  }
}

// a5: inference march, up to n_step samples per live ray; buffers pre-zeroed by this kernel
__global__ void __launch_bounds__(kRayBlock) k_march_rays(MarchParams P, int64_t n_alive, int n_step,
                                                          const int32_t* __restrict__ rays_alive,
                                                          const float* __restrict__ rays_t,
                                                          const float* __restrict__ rays_o,
                                                          const float* __restrict__ rays_d,
                                                          const float* __restrict__ fars,
                                                          float* __restrict__ xyzs, float* __restrict__ dirs,
                                                          float* __restrict__ deltas) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_alive) return;
  const int64_t base = n * n_step;
  const int32_t ridx = rays_alive[n];
  int written = 0;
  if (ridx >= 0) {
    const Ray r = load_ray(rays_o, rays_d, ridx);
    int64_t i = base;
    written = march_ray(P, r, rays_t[ridx], fars[ridx], n_step,
                        [&](float px, float py, float pz, float dt, float delta, float) {
                          xyzs[i * 3 + 0] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
                          dirs[i * 3 + 0] = r.dx; dirs[i * 3 + 1] = r.dy; dirs[i * 3 + 2] = r.dz;
                          deltas[i * 2 + 0] = dt; deltas[i * 2 + 1] = delta;
                          ++i;
                        });
  }
  for (int64_t i = base + written; i < base + n_step; ++i) {
    xyzs[i * 3 + 0] = 0.f; xyzs[i * 3 + 1] = 0.f; xyzs[i * 3 + 2] = 0.f;
    dirs[i * 3 + 0] = 0.f; dirs[i * 3 + 1] = 0.f; dirs[i * 3 + 2] = 0.f;
    deltas[i * 2 + 0] = 0.f; deltas[i * 2 + 1] = 0.f;
  }
}

__global__ void __launch_bounds__(kRayBlock) k_composite_rays(int64_t n_alive, int n_step,
                                                              int32_t* __restrict__ rays_alive,
                                                              float* __restrict__ rays_t,
                                                              const float* __restrict__ sigmas,
                                                              const float* __restrict__ rgbs,
                                                              const float* __restrict__ deltas,
                                                              float* __restrict__ weights_sum,
                                                              float* __restrict__ depth, float* __restrict__ image,
                                                              float T_thresh, const float* __restrict__ extra,
                                                              float* __restrict__ extra_acc, int K) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_alive) return;
  const int32_t r = rays_alive[n];
  if (r < 0) return;
  float t = rays_t[r], ws = weights_sum[r], d = depth[r];
  float cr = image[r * 3], cg = image[r * 3 + 1], cb = image[r * 3 + 2];
  int step = 0;
  bool dead = false;
  while (step < n_step) {
    const int64_t i = n * n_step + step;
    const float d0 = deltas[i * 2];
    if (d0 == 0.0f) { dead = true; break; }
    const float alpha = 1.0f - expf(-sigmas[i] * d0);
    const float T = 1.0f - ws;
    const float w = alpha * T;
    ws = ws + w;
    t = t + deltas[i * 2 + 1];
    d = d + w * t;
    cr = cr + w * rgbs[i * 3]; cg = cg + w * rgbs[i * 3 + 1]; cb = cb + w * rgbs[i * 3 + 2];
    if (extra) {
      for (int k = 0; k < K; ++k) extra_acc[(int64_t)r * K + k] += w * extra[i * K + k];
    }
    ++step;
    if (T * (1.0f - alpha) < T_thresh) { dead = true; break; }
  }
  if (dead) rays_alive[n] = -1; else rays_t[r] = t;
  weights_sum[r] = ws; depth[r] = d;
  image[r * 3] = cr; image[r * 3 + 1] = cg; image[r * 3 + 2] = cb;
}

// order-preserving compaction of live rays: wave ballot + prefix, one workgroup per 1024 rays
// with a two-level scan (block sums scanned by k_scan_block_sums).
__global__ void __launch_bounds__(kRayBlock) k_alive_flags(const int32_t* __restrict__ rays_alive,
                                                           int64_t n, int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ block_sums) {
  __shared__ int32_t wsum[kRayBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int f = (i < n && rays_alive[i] >= 0) ? 1 : 0;
  if (i < n) counts[i] = f;
  const unsigned long long m = __ballot(f);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kRayBlock / 64; ++w) s += wsum[w];
    block_sums[blockIdx.x] = s;
  }
}
__global__ void __launch_bounds__(kRayBlock) k_alive_scatter(const int32_t* __restrict__ rays_alive,
                                                             int64_t n, const int32_t* __restrict__ block_prefix,
                                                             int32_t* __restrict__ out) {
  __shared__ int32_t wsum[kRayBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t v = i < n ? rays_alive[i] : -1;
  const int f = v >= 0 ? 1 : 0;
  const unsigned long long m = __ballot(f);
  const int lane = threadIdx.x & 63;
  const int rank = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  int off = block_prefix[blockIdx.x] + rank;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
  if (f) out[off] = v;
}

// ------------------------------------------------------------------------------------------
// a12: training compositing.  One WAVE per ray: a training batch has few rays (4096) with ~50-200
// samples each, so one-lane-per-ray loops are pure latency (100+ us for 0.2 M samples).  Here the 64
// lanes take 64 consecutive samples (coalesced loads), the transmittance is an inclusive product scan
// across the wave, sums are wave reductions, and the per-sample weights w = alpha*T (0 behind the
// termination point) are stored once so that the K-channel kernels and the backward need no scan of
// their own.  Rounding differs from a sequential loop by ~1 ulp per scan step (tree order).
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float u = __shfl_up(v, d, 64);
    if (lane >= d) v *= u;
  }
  return v;
}
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__global__ void __launch_bounds__(kRayBlock) k_composite_train_fwd(const float* __restrict__ sigmas,
                                                                   const float* __restrict__ rgbs,
                                                                   const float* __restrict__ deltas,
                                                                   const int32_t* __restrict__ rays, int64_t N,
                                                                   int64_t M, float T_thresh,
                                                                   float* __restrict__ weights_sum,
                                                                   float* __restrict__ depth,
                                                                   float* __restrict__ image,
                                                                   float* __restrict__ wbuf,
                                                                   int32_t* __restrict__ sample_ray) {
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int32_t rid = rays[n * 3], off = rays[n * 3 + 1];
  int32_t cnt = rays[n * 3 + 2];
  if ((int64_t)off + cnt > M) {            // the writer dropped this ray (sample buffer sized from mean_count)
    // the rows between its offset and the end of the buffer belong to nobody: zero weight, so that a consumer that
    // walks the samples (k_instance_head_bwd) needs no ownership test
    if (wbuf)
      for (int64_t k = (int64_t)off + lane; k < M; k += 64) {
        wbuf[k] = 0.0f;
        if (sample_ray) sample_ray[k] = rid;
      }
    cnt = 0;
  }
  float T_carry = 1.0f, t_carry = 0.0f, r = 0, g = 0, b = 0, ws = 0, dsum = 0;
  for (int base = 0; base < cnt; base += 64) {
    const bool act = base + lane < cnt;
    const int64_t i = (int64_t)off + base + lane;
    float alpha = 0.f, dy = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    if (act) {
      const float2 dl = reinterpret_cast<const float2*>(deltas)[i];
      alpha = 1.0f - expf(-sigmas[i] * dl.x);
      dy = dl.y;
      cr = rgbs[i * 3]; cg = rgbs[i * 3 + 1]; cb = rgbs[i * 3 + 2];
    }
    const float P = wave_incl_prod(1.0f - alpha, lane);
    float Pprev = __shfl_up(P, 1, 64);
    if (lane == 0) Pprev = 1.0f;
    const float T = T_carry * Pprev;                 // transmittance before this sample
    const bool used = act && (T >= T_thresh || (base + lane) == 0);
    const float w = used ? alpha * T : 0.0f;
    const float t = t_carry + wave_incl_sum(dy, lane);
    r += w * cr; g += w * cg; b += w * cb; ws += w; dsum += w * t;
    if (wbuf && act) wbuf[i] = w;
    if (sample_ray && act) sample_ray[i] = rid;      // row of the ray's outputs (k_instance_head_bwd looks dL/dpix up by it)
    T_carry *= __shfl(P, 63, 64);
    t_carry = __shfl(t, 63, 64);
    if (T_carry < T_thresh) {                        // wave-uniform: the rest of the ray is unused
      if (wbuf)
        for (int64_t k = (int64_t)base + 64 + lane; k < cnt; k += 64) {
          wbuf[(int64_t)off + k] = 0.0f;
          if (sample_ray) sample_ray[(int64_t)off + k] = rid;
        }
      break;
    }
  }
  r = wave_sum(r); g = wave_sum(g); b = wave_sum(b); ws = wave_sum(ws); dsum = wave_sum(dsum);
  if (lane == 0) {
    weights_sum[rid] = ws; depth[rid] = dsum;
    image[rid * 3] = r; image[rid * 3 + 1] = g; image[rid * 3 + 2] = b;
  }
}

// a13: K extra channels (instance logits): out[ray][ch] = sum_i w_i * extra[i][ch]; one wave per ray,
// lane = channel (coalesced 4K-byte rows), no scan (weights come from k_composite_train_fwd)
struct CeArgs {                  // optional cross-entropy epilogue of the K-channel compositing (labels == nullptr: none)
  const int64_t* labels;         // [N] by output row (rays[n][0])
  int n_classes;                 // logits 0 .. n_classes-1 take part (K may be padded beyond)
  int64_t ignore_index;
  float* dpix;                   // [N,K]: softmax - onehot for kept rows, 0 otherwise (NOT yet divided by the kept count)
  float4* part;                  // [N]: (row loss, kept 0/1, label out of range 0/1, 0)
};
__global__ void __launch_bounds__(kRayBlock) k_composite_train_extra_fwd(const float* __restrict__ wbuf,
                                                                         const float* __restrict__ extra,
                                                                         const int32_t* __restrict__ rays, int64_t N,
                                                                         int64_t M, int K,
                                                                         float* __restrict__ extra_out, CeArgs ce) {
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int32_t rid = rays[n * 3], off = rays[n * 3 + 1];
  int32_t cnt = rays[n * 3 + 2];
  if ((int64_t)off + cnt > M) cnt = 0;
  float acc = 0.0f;
  if (lane < K) {
    int s = 0;
    for (; s + 8 <= cnt; s += 8) {                   // 8 independent row loads in flight (a ray is a chain of round trips)
      const int64_t i = (int64_t)off + s;
      float e[8], w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { e[u] = extra[(i + u) * K + lane]; w[u] = wbuf[i + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u] * e[u];       // same order of additions as the scalar tail
    }
    for (; s + 4 <= cnt; s += 4) {
      const int64_t i = (int64_t)off + s;
      const float e0 = extra[i * K + lane], e1 = extra[(i + 1) * K + lane], e2 = extra[(i + 2) * K + lane],
                  e3 = extra[(i + 3) * K + lane];
      acc += wbuf[i] * e0;
      acc += wbuf[i + 1] * e1;
      acc += wbuf[i + 2] * e2;
      acc += wbuf[i + 3] * e3;
    }
    for (; s < cnt; ++s) acc += wbuf[(int64_t)off + s] * extra[((int64_t)off + s) * K + lane];
    extra_out[(int64_t)rid * K + lane] = acc;
  }
  if (ce.labels) {
    // the rendered logits of this ray are in the lanes: its cross-entropy row right here (same arithmetic as k_ce_rows)
    const int64_t y = ce.labels[rid];
    const bool in_range = y >= 0 && y < ce.n_classes;
    const bool keep = y != ce.ignore_index && in_range;
    const bool live = lane < ce.n_classes;
    const float x = live ? acc : -INFINITY;
    float m = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const float e = live ? expf(x - m) : 0.f;
    float ssum = e;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) ssum += __shfl_xor(ssum, d, 64);
    const float xy = __shfl(x, keep ? (int)y : 0, 64);
    if (lane < K) ce.dpix[(int64_t)rid * K + lane] = (keep && live) ? e / ssum - (lane == (int)y ? 1.f : 0.f) : 0.f;
    if (lane == 0)
      ce.part[rid] = make_float4(keep ? (m + logf(ssum)) - xy : 0.f, keep ? 1.f : 0.f,
                                 (y != ce.ignore_index && !in_range) ? 1.f : 0.f, 0.f);
  }
}

// loss_out[0] = mean row loss over the kept rows (NaN when nothing is kept, as torch's mean over an empty set - and NaN
// when a label is neither ignore_index nor a class: torch's cross_entropy asserts on the device for such a label, a
// silent drop would train on fewer rows without anyone noticing), [1] = 1 / kept (0 if none), [2] = kept, [3] = bad.
// One workgroup; sums in a fixed order (deterministic).
__global__ void __launch_bounds__(1024) k_ce_finalize(const float4* __restrict__ part, int64_t N, float* __restrict__ loss_out) {
  __shared__ float red[3][16];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int64_t i = threadIdx.x; i < N; i += 1024) {
    const float4 p = part[i];
    a += p.x; b += p.y; c += p.z;
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; red[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int w = 0; w < 16; ++w) { sa += red[0][w]; sb += red[1][w]; sc += red[2][w]; }
    loss_out[0] = sc > 0.f ? NAN : sa / sb;
    loss_out[1] = sb > 0.f ? 1.0f / sb : 0.f;
    loss_out[2] = sb;
    loss_out[3] = sc;
  }
}

__global__ void __launch_bounds__(kRayBlock) k_composite_train_bwd(
    const float* __restrict__ g_ws, const float* __restrict__ g_img, const float* __restrict__ sigmas,
    const float* __restrict__ rgbs, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
    const float* __restrict__ weights_sum, const float* __restrict__ image, int64_t N, int64_t M, float T_thresh,
    float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs, const int32_t* __restrict__ total_dev) {
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int32_t rid = rays[n * 3], off = rays[n * 3 + 1];
  int32_t cnt = rays[n * 3 + 2];
  // total_dev (the marcher's sample total): the rays' rows tile [0, total) without gaps, so every row of [0, M) has
  // exactly one writer here and the caller's buffers need no zero fill (two fill launches less per training step) -
  // owned rows below; the rows from the one dropped ray that starts inside the buffer to its end; and the padding
  // [total, M), shared out over the waves.  Without it unowned rows are left untouched (the caller zero-initialises).
  auto zero_rows = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo + lane; i < hi; i += 64) {
      grad_sigmas[i] = 0.f; grad_rgbs[i * 3] = 0.f; grad_rgbs[i * 3 + 1] = 0.f; grad_rgbs[i * 3 + 2] = 0.f;
    }
  };
  if (total_dev) {
    const int64_t total = total_dev[0];
    if (total < M) {
      const int64_t per = (M - total + N - 1) / N;
      zero_rows(total + n * per, min(M, total + (n + 1) * per));
    }
  }
  if ((int64_t)off + cnt > M) {            // dropped ray
    if (total_dev && cnt > 0 && off < M) zero_rows(off, M);
    cnt = 0;
  }
  const float gr = g_img[rid * 3], gg = g_img[rid * 3 + 1], gb = g_img[rid * 3 + 2];
  const float gw = g_ws ? g_ws[rid] : 0.0f;
  const float rf = image[rid * 3], gf = image[rid * 3 + 1], bf = image[rid * 3 + 2];
  const float wsf = weights_sum[rid];
  float T_carry = 1.0f, cr_carry = 0.f, cg_carry = 0.f, cb_carry = 0.f;
  bool dead = false;                                  // wave-uniform: termination point passed
  for (int base = 0; base < cnt; base += 64) {
    const bool act = base + lane < cnt;
    const int64_t i = (int64_t)off + base + lane;
    if (dead) {
      if (act) { grad_sigmas[i] = 0.f; grad_rgbs[i * 3] = 0.f; grad_rgbs[i * 3 + 1] = 0.f; grad_rgbs[i * 3 + 2] = 0.f; }
      continue;
    }
    float alpha = 0.f, d0 = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    if (act) {
      d0 = deltas[i * 2];
      alpha = 1.0f - expf(-sigmas[i] * d0);
      cr = rgbs[i * 3]; cg = rgbs[i * 3 + 1]; cb = rgbs[i * 3 + 2];
    }
    const float P = wave_incl_prod(1.0f - alpha, lane);
    float Pprev = __shfl_up(P, 1, 64);
    if (lane == 0) Pprev = 1.0f;
    const float T = T_carry * Pprev, Tn = T_carry * P;     // before / after this sample
    const bool used = act && (T >= T_thresh || (base + lane) == 0);
    const float w = used ? alpha * T : 0.0f;
    const float ar = cr_carry + wave_incl_sum(w * cr, lane);   // colour accumulated up to and including i
    const float ag = cg_carry + wave_incl_sum(w * cg, lane);
    const float ab = cb_carry + wave_incl_sum(w * cb, lane);
    if (act) {
      grad_rgbs[i * 3] = gr * w; grad_rgbs[i * 3 + 1] = gg * w; grad_rgbs[i * 3 + 2] = gb * w;
      grad_sigmas[i] = used ? d0 * (gr * (Tn * cr - (rf - ar)) + gg * (Tn * cg - (gf - ag)) + gb * (Tn * cb - (bf - ab)) +
                                    gw * (1.0f - wsf))
                            : 0.0f;
    }
    T_carry *= __shfl(P, 63, 64);
    cr_carry = __shfl(ar, 63, 64); cg_carry = __shfl(ag, 63, 64); cb_carry = __shfl(ab, 63, 64);
    if (T_carry < T_thresh) dead = true;
  }
}

__global__ void __launch_bounds__(kRayBlock) k_composite_train_extra_bwd(
    const float* __restrict__ g_extra_out, const float* __restrict__ wbuf, const int32_t* __restrict__ rays, int64_t N,
    int64_t M, int K, float* __restrict__ grad_extra) {
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int32_t rid = rays[n * 3], off = rays[n * 3 + 1];
  int32_t cnt = rays[n * 3 + 2];
  if ((int64_t)off + cnt > M) cnt = 0;
  if (lane >= K) return;
  const float g = g_extra_out[(int64_t)rid * K + lane];
  for (int s = 0; s < cnt; ++s) grad_extra[((int64_t)off + s) * K + lane] = wbuf[(int64_t)off + s] * g;
}

// ------------------------------------------------------------------------------------------
// Patch-interleaved sample layout (fused full-frame inference path).
//
// Rays are taken in groups of 16 consecutive rays (the caller orders rays so that a group is a
// 4x4 pixel patch; any order is CORRECT, a compact patch is FAST).  Inside a group the samples
// are stored step-major: all k-th samples of the group's rays are adjacent,
//     slot(r, k) = base_g + sum_i min(c_i, k) + #{ i < r : c_i > k }
// (c_i = sample count of ray i of the group, base_g = exclusive scan of the counts at the group's
// first ray - exactly what inr_march_rays_train_count leaves in rays[:,1]).  A 16-sample tile of
// the field kernel is then a compact 4x4xfew-steps block of space instead of 16 steps along one
// ray: neighbouring lanes hit the same hash-grid cells and 128-byte lines (measured: gather phase
// 10.2 -> 6.8 ms per 37 M samples), and the compositing reads of a group are coalesced.
constexpr int kGroup = 16;

struct GroupCursor {
  int c[kGroup];
  int S;      // base + sum_i min(c_i, k) for the current k
  int r;
  int k;
  __device__ __forceinline__ void init(int my_count, int my_offset) {
    const int lane = threadIdx.x & 63, g0 = lane & ~(kGroup - 1);
    r = lane & (kGroup - 1);
#pragma unroll
    for (int i = 0; i < kGroup; ++i) c[i] = __shfl(my_count, g0 + i, 64);
    S = __shfl(my_offset, g0, 64);
    k = 0;
  }
  __device__ __forceinline__ int total() const {
    int t = 0;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) t += c[i];
    return t;
  }
  // slot of this ray's k-th sample; must be called for k = 0, 1, 2, ... in order.
  // nact = #{i : c_i > k} and rank = #{i < r : c_i > k} only change when k passes one of the 16 counts: they are
  // recomputed (16 compares) at those break points - at most 16 times per ray - and a sample otherwise costs one add
  // (round 1 recomputed both for every sample: ~50 VALU instructions per emitted sample).
  int nact_ = 0, rank_ = 0, k_change_ = 0;       // valid for k in [.., k_change_)
  __device__ __forceinline__ int next() {
    if (k >= k_change_) {
      int nact = 0, rank = 0, nxt = 0x7FFFFFFF;
#pragma unroll
      for (int i = 0; i < kGroup; ++i) {
        const bool gt = c[i] > k;
        nact += gt ? 1 : 0;
        rank += (gt && i < r) ? 1 : 0;
        nxt = gt ? min(nxt, c[i]) : nxt;        // the next k at which a ray of the group runs out
      }
      nact_ = nact; rank_ = rank; k_change_ = nxt;
    }
    const int slot = S + rank_;
    S += nact_;
    ++k;
    return slot;
  }
};

__global__ void __launch_bounds__(kRayBlock) k_march_write_patch(MarchParams P, const float* __restrict__ rays_o,
                                                                 const float* __restrict__ rays_d, int64_t N, int64_t M,
                                                                 const float* __restrict__ nears,
                                                                 const float* __restrict__ fars,
                                                                 const float* __restrict__ noises,
                                                                 const int32_t* __restrict__ rays,
                                                                 float* __restrict__ xyzs, float* __restrict__ dirs,
                                                                 float* __restrict__ deltas,
                                                                 const uint32_t* __restrict__ mask, int cap_words,
                                                                 int32_t* __restrict__ ray_ids, int normalise) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int off = n < N ? rays[n * 3 + 1] : 0;
  const int cnt = n < N ? rays[n * 3 + 2] : 0;
  GroupCursor cur;
  cur.init(cnt, off);                                   // all 64 lanes take part in the shuffles
  if (cnt == 0 || (int64_t)cur.S + cur.total() > M) return;   // a group that does not fit is dropped whole
  const Ray r = load_ray(rays_o, rays_d, n);
  const float rb = 2.0f * P.bound;
  auto emit = [&](float px, float py, float pz, float dt, float delta, float) {
    const int64_t i = cur.next();
    if (normalise) {      // x01 = (p + bound) / (2 bound), the very operation the field kernel would do per lane
      px = (px + P.bound) / rb; py = (py + P.bound) / rb; pz = (pz + P.bound) / rb;
    }
    xyzs[i * 3 + 0] = px; xyzs[i * 3 + 1] = py; xyzs[i * 3 + 2] = pz;
    if (dirs) { dirs[i * 3 + 0] = r.dx; dirs[i * 3 + 1] = r.dy; dirs[i * 3 + 2] = r.dz; }
    if (ray_ids) ray_ids[i] = (int32_t)n;
    deltas[i * 2 + 0] = dt; deltas[i * 2 + 1] = delta;
  };
  const float t0 = start_t(P, nears[n], noises ? noises[n] : 0.0f);
  if (cap_words > 0 && mask[(int64_t)cap_words * N + n]) {
    replay_ray(P, r, mask, N, n, t0, cnt, emit);
  } else {
    march_ray(P, r, t0, fars[n], cnt, emit);
  }
}

// one lane per ray; the 16 lanes of a group walk k in lockstep, so slot ranks come from one ballot
__global__ void __launch_bounds__(kRayBlock) k_composite_patch_fwd(const float* __restrict__ sigmas,
                                                                   const float* __restrict__ rgbs,
                                                                   const float* __restrict__ deltas,
                                                                   const int32_t* __restrict__ rays, int64_t N, int64_t M,
                                                                   float T_thresh, float* __restrict__ wbuf,
                                                                   float* __restrict__ weights_sum,
                                                                   float* __restrict__ depth, float* __restrict__ image,
                                                                   unsigned long long* __restrict__ skippable) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, g0 = lane & ~(kGroup - 1), rr = lane & (kGroup - 1);
  const int32_t rid = n < N ? rays[n * 3] : 0;
  const int off = n < N ? rays[n * 3 + 1] : 0;
  int cnt = n < N ? rays[n * 3 + 2] : 0;
  int S = __shfl(off, g0, 64);
  int gtot = cnt;                                          // group total and wave-wide max count
#pragma unroll
  for (int d = 1; d < kGroup; d <<= 1) gtot += __shfl_xor(gtot, d, 64);
  if ((int64_t)S + gtot > M) cnt = 0;                      // group was dropped by the writer
  int maxc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) maxc = max(maxc, __shfl_xor(maxc, d, 64));
  float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, dsum = 0;
  bool done = false;
  unsigned n_skip = 0;            // samples of steps at which every ray of the group is already below T_thresh
  // Eight steps per trip: the slots of steps k..k+7 depend on the counts only, so their 8 x 3 loads are requested
  // together and the eight compositing updates then run in order - the same operations in the same order as a
  // one-step loop (which waited for its loads once per step: 0.25 ms per frame, measured; this: see r02_NOTES 25).
  constexpr int kAhead = 8;
  for (int k = 0; k < maxc; k += kAhead) {
    unsigned field[kAhead];
    int64_t slot[kAhead];
    bool active[kAhead];
    float2 dl[kAhead];
    float sg[kAhead], cr[kAhead], cg[kAhead], cb[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      active[j] = k + j < cnt;
      field[j] = (unsigned)(__ballot(active[j]) >> g0) & 0xFFFFu;
      slot[j] = (int64_t)S + __popc(field[j] & ((1u << rr) - 1u));
      S += __popc(field[j]);
      const int64_t i = (active[j] && !done) ? slot[j] : 0;      // a ray that is already opaque requests nothing new
      dl[j] = reinterpret_cast<const float2*>(deltas)[i];
      sg[j] = sigmas[i];
      cr[j] = rgbs[i * 3]; cg[j] = rgbs[i * 3 + 1]; cb[j] = rgbs[i * 3 + 2];
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      if (skippable) {            // (wave-uniform) what the early-terminating kernel would not have evaluated
        const unsigned live = (unsigned)(__ballot(active[j] && !done) >> g0) & 0xFFFFu;
        if (rr == 0 && field[j] != 0 && live == 0) n_skip += __popc(field[j]);
      }
      if (active[j] && !done) {
        const float alpha = 1.0f - expf(-sg[j] * dl[j].x);
        const float w = alpha * T;
        r += w * cr[j]; g += w * cg[j]; b += w * cb[j];
        t += dl[j].y;
        dsum += w * t;
        ws += w;
        if (wbuf) wbuf[slot[j]] = w;
        T *= 1.0f - alpha;
        if (T < T_thresh) done = true;
      } else if (active[j] && wbuf) {
        wbuf[slot[j]] = 0.0f;                                    // behind the termination point
      }
    }
  }
  if (n < N) {
    weights_sum[rid] = ws; depth[rid] = dsum;
    image[rid * 3] = r; image[rid * 3 + 1] = g; image[rid * 3 + 2] = b;
  }
  if (skippable) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) n_skip += __shfl_xor(n_skip, d, 64);
    if (lane == 0 && n_skip) atomicAdd(skippable, (unsigned long long)n_skip);
  }
}

// K extra channels of the patch-interleaved layout: out[ray][ch] = sum_k w[slot(r,k)] * extra[slot(r,k)][ch].
// One wave per ray, lane = channel (coalesced 4K-byte rows); slot(r,k) = base + sum_i min(c_i,k) + #{i<r: c_i>k}
// is advanced incrementally from the 16 counts of the ray's group (wave-uniform scalar work).
__global__ void __launch_bounds__(kRayBlock) k_composite_patch_extra(const float* __restrict__ wbuf,
                                                                     const float* __restrict__ extra,
                                                                     const int32_t* __restrict__ rays, int64_t N,
                                                                     int64_t M, int K, float* __restrict__ extra_out) {
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int64_t g0 = n & ~(int64_t)(kGroup - 1);
  const int r = (int)(n - g0);
  int c[kGroup];
  int total = 0;
#pragma unroll
  for (int i = 0; i < kGroup; ++i) {
    c[i] = (g0 + i < N) ? rays[(g0 + i) * 3 + 2] : 0;
    total += c[i];
  }
  const int32_t rid = rays[n * 3];
  int64_t S = rays[g0 * 3 + 1];
  const int cnt = (S + total <= M) ? c[r] : 0;          // dropped group
  float acc = 0.0f;
  for (int k = 0; k < cnt; ++k) {
    int nact = 0, rank = 0;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
      const int gt = c[i] > k ? 1 : 0;
      nact += gt;
      rank += (i < r) ? gt : 0;
    }
    const int64_t slot = S + rank;
    S += nact;
    const float w = wbuf[slot];
    if (lane < K && w != 0.0f) acc += w * extra[slot * K + lane];
  }
  if (lane < K) extra_out[(int64_t)rid * K + lane] = acc;
}

// NeRF-weighted projection of up to 32 voxel masks along the rays of a patch-interleaved frame (the
// counterpart of the reference pipeline's project_3d_masks step): out[ray][i] = sum_k w[slot(ray,k)] * bit_i(cell(x)).
// The masks arrive as ONE 32-bit word per voxel (bit i = mask i contains the voxel), so a sample costs one 4-byte load
// whatever the number of masks - the tensor-op version gathered a [k, M] float matrix (3.8 GB for a 800x800 frame and
// 30 masks) and composited it as K extra channels.  One lane per ray, the 16 lanes of a group walk k in lockstep (the
// slot ranks of k_composite_patch_fwd); the cell is formed with the operations of the tensor-op version, in its order:
// floor(((x - lo) / (hi - lo)) * res), outside -> no mask.  Same sums in the same order: bit-identical results.
struct MaskVolume {
  float lo[3], hi[3], res[3];
  int W, L, H;
};
__global__ void __launch_bounds__(kRayBlock) k_project_masks_patch(const float* __restrict__ xyzs,
                                                                   const float* __restrict__ wbuf,
                                                                   const int32_t* __restrict__ rays, int64_t N, int64_t M,
                                                                   const uint32_t* __restrict__ words, MaskVolume V,
                                                                   int k_total, int k_base, int k_count,
                                                                   float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, g0 = lane & ~(kGroup - 1), rr = lane & (kGroup - 1);
  const int32_t rid = n < N ? rays[n * 3] : 0;
  const int off = n < N ? rays[n * 3 + 1] : 0;
  int cnt = n < N ? rays[n * 3 + 2] : 0;
  int S = __shfl(off, g0, 64);
  int gtot = cnt;
#pragma unroll
  for (int d = 1; d < kGroup; d <<= 1) gtot += __shfl_xor(gtot, d, 64);
  if ((int64_t)S + gtot > M) cnt = 0;                      // group was dropped by the writer
  int maxc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) maxc = max(maxc, __shfl_xor(maxc, d, 64));
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.0f;
  for (int k = 0; k < maxc; ++k) {
    const bool active = k < cnt;
    const unsigned field = (unsigned)(__ballot(active) >> g0) & 0xFFFFu;
    const int64_t slot = (int64_t)S + __popc(field & ((1u << rr) - 1u));
    S += __popc(field);
    const float w = active ? wbuf[slot] : 0.0f;
    if (w != 0.0f) {                                       // (behind a ray's termination point the weights are zero)
      const float x = xyzs[slot * 3], y = xyzs[slot * 3 + 1], z = xyzs[slot * 3 + 2];
      const float fx = floorf(((x - V.lo[0]) / (V.hi[0] - V.lo[0])) * V.res[0]);
      const float fy = floorf(((y - V.lo[1]) / (V.hi[1] - V.lo[1])) * V.res[1]);
      const float fz = floorf(((z - V.lo[2]) / (V.hi[2] - V.lo[2])) * V.res[2]);
      const bool inside = fx >= 0.0f && fx < V.res[0] && fy >= 0.0f && fy < V.res[1] && fz >= 0.0f && fz < V.res[2];
      if (inside) {
        const uint32_t word = words[((int64_t)(int)fx * V.L + (int)fy) * V.H + (int)fz];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] += ((word >> i) & 1u) ? w : 0.0f;
      }
    }
  }
  if (n < N) {
    float* dst = out + (int64_t)rid * k_total + k_base;
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (i < k_count) dst[i] = acc[i];
  }
}

static MarchParams make_params(const uint8_t* bits, float bound, float dt_gamma, int max_steps, int C, int H) {
  MarchParams P;
  P.bits = bits;
  P.bound = bound;
  P.dt_gamma = dt_gamma;
  P.dt_min = kSqrt3x2 / (float)max_steps;
  P.dt_max = kSqrt3x2 * (float)(1 << (C - 1)) / (float)H;
  P.rH = (float)(1.0 / H);
  P.C = C;
  P.H = H;
  P.H3 = H * H * H;
  return P;
}

}  // namespace inr

using namespace inr;

extern "C" {

int inr_abi_version(void) { return INR_ABI_VERSION; }
const char* inr_last_error(void) { return inr::g_err; }

int inr_set_overlap_placement(int32_t on) {
  // LDS is the one per-CU resource a launch can ask for without changing the kernel: 160 KB per CU.  The field
  // workgroup asks for > 80 KB (so two never share a CU, also not while the next view's march holds registers on some
  // CUs and the dispatcher looks for room elsewhere), a march workgroup for 48 KB on top of its own, which bounds the
  // marchers that can sit beside a field workgroup.  Measured in profiles/r02_NOTES.txt section 18.
  // on >= 1024: the field workgroup's LDS request in BYTES instead of the 84 KB default (a tuning knob: 54..80 KB lets two
  // field workgroups share a CU again).  The value is NOT clamped: a request beyond what a CU has makes the next field
  // launch fail, which comes back from that call as INR_ELAUNCH with the runtime's message - never an abort
  // (tests/test_gpu_parity.py::test_launch_failure_is_a_return_code).
  INR_REQUIRE(on >= 0, "on must be 0 (off), 1 (on) or an LDS byte count >= 1024");
  INR_REQUIRE(on <= 1 || on >= 1024, "on must be 0 (off), 1 (on) or an LDS byte count >= 1024");
  inr::g_field_lds_min = on >= 1024 ? on : on ? 84 * 1024 : 0;
  inr::g_march_lds_pad = on ? 48 * 1024 : 0;
  return INR_OK;
}

int inr_device_info(int32_t device, int64_t* props) {
  INR_REQUIRE(props, "props is null");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) {
    set_error("inr_device_info: no device %d", device);
    return INR_ENODEV;
  }
  props[0] = p.multiProcessorCount;
  props[1] = p.warpSize;
  props[2] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
  props[3] = p.gcnArchName[0] ? atoi(p.gcnArchName + 3) : 0;
  return INR_OK;
}

int inr_get_rays(const float* poses, int64_t B, float fx, float fy, float cx, float cy, int32_t W, const int64_t* inds,
                 int64_t n, float* rays_o, float* rays_d, inr_stream_t s) {
  INR_REQUIRE(B >= 0 && n >= 0 && W > 0, "bad sizes");
  if (B * n == 0) return INR_OK;
  INR_REQUIRE(poses && rays_o && rays_d, "null pointer");
  k_get_rays<<<blocks_for(B * n, kRayBlock), kRayBlock, 0, as_stream(s)>>>(poses, B, fx, fy, cx, cy, W, inds, n, rays_o,
                                                                           rays_d);
  return check_launch("get_rays");
}

int inr_sample_training_batch(const float* pose, float fx, float fy, float cx, float cy, int32_t H, int32_t W,
                              const float* image, int32_t channels, const int32_t* mask, int32_t num_instances,
                              int64_t seed, int64_t step, int64_t n, int64_t* inds, float* rays_o, float* rays_d,
                              float* rgb, int64_t* labels, inr_stream_t s) {
  INR_REQUIRE(n >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1ll << 31), "bad sizes");
  INR_REQUIRE(step >= 0 && step < (1ll << 31), "step out of range (0 .. 2^31 - 1)");
  INR_REQUIRE(!image || channels == 3 || channels == 4, "channels must be 3 or 4");
  INR_REQUIRE(!mask || num_instances > 0, "num_instances must be positive with a mask");
  if (n == 0) return INR_OK;
  INR_REQUIRE(pose && inds && rays_o && rays_d, "null pointer");
  INR_REQUIRE((!image || rgb) && (!mask || labels), "null output for a given input");
  k_sample_batch<<<blocks_for(n, kRayBlock), kRayBlock, 0, as_stream(s)>>>(pose, fx, fy, cx, cy, H, W, image, channels, mask,
                                                                           num_instances, (uint64_t)seed, (uint64_t)step, n,
                                                                           inds, rays_o, rays_d, rgb, labels);
  return check_launch("sample_training_batch");
}

int inr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, int64_t N,
                           float min_near, float* nears, float* fars, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays_o && rays_d && aabb && nears && fars, "null pointer");
  k_near_far<<<blocks_for(N, kRayBlock), kRayBlock, 0, as_stream(s)>>>(rays_o, rays_d, aabb, N, min_near, nears, fars,
                                                                       nullptr, 0);
  return check_launch("near_far_from_aabb");
}

int inr_near_far_from_aabb_skip(const float* rays_o, const float* rays_d, const float* aabb, int64_t N,
                                float min_near, const int64_t* labels, int64_t ignore_index, float* nears,
                                float* fars, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays_o && rays_d && aabb && nears && fars && labels, "null pointer");
  k_near_far<<<blocks_for(N, kRayBlock), kRayBlock, 0, as_stream(s)>>>(rays_o, rays_d, aabb, N, min_near, nears, fars,
                                                                       labels, ignore_index);
  return check_launch("near_far_from_aabb_skip");
}

int inr_morton3D(const int32_t* coords, int64_t N, int32_t* indices, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(coords && indices, "null pointer");
  k_morton3D<<<blocks_for(N, 256), 256, 0, as_stream(s)>>>(coords, N, indices);
  return check_launch("morton3D");
}
int inr_morton3D_invert(const int32_t* indices, int64_t N, int32_t* coords, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(coords && indices, "null pointer");
  k_morton3D_invert<<<blocks_for(N, 256), 256, 0, as_stream(s)>>>(indices, N, coords);
  return check_launch("morton3D_invert");
}
int inr_packbits(const float* grid, int64_t n_bytes, float thresh, uint8_t* bitfield, inr_stream_t s) {
  INR_REQUIRE(n_bytes >= 0, "negative size");
  if (n_bytes == 0) return INR_OK;
  INR_REQUIRE(grid && bitfield, "null pointer");
  INR_REQUIRE(((uintptr_t)grid & 15) == 0, "grid must be 16-byte aligned");
  k_packbits<<<blocks_for(n_bytes, 256), 256, 0, as_stream(s)>>>(grid, n_bytes, thresh, bitfield);
  return check_launch("packbits");
}

int inr_mark_untrained_grid(const float* poses, int32_t B, float fx, float fy, float cx, float cy, int32_t H, int32_t cascade,
                            float bound, float* grid, inr_stream_t s) {
  INR_REQUIRE(B >= 0 && H >= 2 && H <= 1024 && cascade >= 1 && grid, "bad argument");
  INR_REQUIRE(B == 0 || poses, "null poses");
  INR_REQUIRE(fx != 0.0f && fy != 0.0f, "zero focal length");
  const int64_t n = (int64_t)cascade * H * H * H;
  k_mark_untrained<<<blocks_for(n, 256), 256, 0, as_stream(s)>>>(poses, B, cx / fx, cy / fy, H, cascade, bound, grid);
  return check_launch("mark_untrained_grid");
}

int inr_occ_cell_positions(const int32_t* morton_idx, const float* noise, int64_t n, int32_t H, float cascade_bound,
                           float* xyz, inr_stream_t s) {
  INR_REQUIRE(n >= 0 && H >= 2 && H <= 1024, "bad argument");
  if (n == 0) return INR_OK;
  INR_REQUIRE(xyz, "null pointer");
  const float half = cascade_bound / (float)H;
  k_occ_positions<<<blocks_for(n, 256), 256, 0, as_stream(s)>>>(morton_idx, noise, n, (float)(H - 1), cascade_bound - half,
                                                                half, xyz);
  return check_launch("occ_cell_positions");
}

int inr_occ_update(float* grid, const float* sigma, const int32_t* morton_idx, int64_t n_cells, int64_t m, float decay,
                   float density_scale, float* tmp, double* mean_sum, inr_stream_t s) {
  INR_REQUIRE(n_cells > 0 && m >= 0 && grid && mean_sum, "bad argument");
  INR_REQUIRE(m == 0 || sigma, "null sigma");
  hipStream_t st = as_stream(s);
  const unsigned nb = (unsigned)std::min<int64_t>((n_cells + 1023) / 1024, (int64_t)cu_count() * 2);
  if (morton_idx) {
    INR_REQUIRE(tmp, "a listed update needs the scratch grid");
    if (hipMemsetAsync(tmp, 0xBF, (size_t)n_cells * sizeof(float), st) != hipSuccess) {   // 0xBFBFBFBF = -1.498: not visited
      set_error("occ_update: memset failed");
      return INR_ELAUNCH;
    }
    if (m) k_occ_scatter<<<blocks_for(m, 256), 256, 0, st>>>(sigma, morton_idx, m, density_scale, tmp);
    k_occ_update<<<nb, 256, 0, st>>>(grid, tmp, n_cells, decay, 1.0f, mean_sum);
  } else {
    INR_REQUIRE(m == n_cells, "a full sweep needs one sigma per cell (Morton order)");
    k_occ_update<<<nb, 256, 0, st>>>(grid, sigma, n_cells, decay, density_scale, mean_sum);
  }
  return check_launch("occ_update");
}

int inr_packbits_mean(const float* grid, int64_t n_cells, const double* mean_sum, float density_thresh, uint8_t* bitfield,
                      float* mean_out, const int32_t* counters, int32_t n_counters, int32_t counter_stride,
                      double* stats_out, inr_stream_t s) {
  INR_REQUIRE(n_cells > 0 && n_cells % 8 == 0 && grid && mean_sum && bitfield, "bad argument");
  INR_REQUIRE(((uintptr_t)grid & 15) == 0, "grid must be 16-byte aligned");
  INR_REQUIRE(n_counters >= 0 && (n_counters == 0 || (counters && stats_out && counter_stride > 0)), "bad counters");
  k_packbits_mean<<<blocks_for(n_cells / 8, 256), 256, 0, as_stream(s)>>>(grid, n_cells / 8, mean_sum, 1.0 / (double)n_cells,
                                                                           density_thresh, bitfield, mean_out, counters,
                                                                           n_counters, counter_stride, stats_out);
  return check_launch("packbits_mean");
}

int64_t inr_occ_sample_workspace_bytes(int64_t n_cells) {
  if (n_cells <= 0) return -1;
  const int64_t n_blocks = (n_cells + kOccBlockCells - 1) / kOccBlockCells;
  return n_blocks * (kOccBlockCells / 64) * (int64_t)sizeof(uint64_t) + (n_blocks + 2 * kOccBuckets) * (int64_t)sizeof(int32_t);
}

int inr_occ_sample_cells(const float* grid, int64_t n_cells, const float* u, int64_t n, int32_t* morton_idx,
                         void* workspace, inr_stream_t s) {
  INR_REQUIRE(grid && u && morton_idx && workspace && n > 0 && n_cells > 0, "bad argument");
  INR_REQUIRE(2 * n < (int64_t)1 << 31 && n_cells < (int64_t)1 << 31, "sizes must fit int32");
  INR_REQUIRE(((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
  const int64_t n_blocks = (n_cells + kOccBlockCells - 1) / kOccBlockCells;
  INR_REQUIRE(n_blocks <= kOccMaxBlocks, "grid too large for the block table (H <= 256)");
  hipStream_t st = as_stream(s);
  uint64_t* masks = reinterpret_cast<uint64_t*>(workspace);
  int32_t* block_counts = reinterpret_cast<int32_t*>(masks + n_blocks * (kOccBlockCells / 64));
  int32_t* hist = block_counts + n_blocks;
  k_occ_count<<<(int)n_blocks, kOccBlockCells, 0, st>>>(grid, n_cells, masks, block_counts, hist);
  k_occ_hist<<<(int)std::min<int64_t>(64, blocks_for(2 * n, 1024)), 1024, 0, st>>>(u, n, hist);
  const size_t lds = (size_t)(2 * (kOccBuckets + 1) + n_blocks + 1) * sizeof(int32_t);
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {
    if (hipFuncSetAttribute((const void*)k_occ_pick, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      set_error("occ_sample_cells: %zu bytes of LDS refused", lds);
      return INR_ELAUNCH;
    }
    lds_allowed = lds;
  }
  k_occ_pick<<<(int)std::min<int64_t>(256, blocks_for(2 * n, 1024)), 1024, lds, st>>>(u + 2 * n, n, hist, masks, block_counts,
                                                                                       (int)n_blocks, n_cells, morton_idx);
  return check_launch("occ_sample_cells");
}

// Small batches (training) take the wave-per-ray marcher; both marchers produce the same bits.
// inr_set_march_mode() forces the choice (the parity tests run both); the library reads no environment variable.
static int g_march_mode = -1;            // -1: by batch size, 0: lane per ray, 1: wave per ray
static bool use_coop(int64_t N, int32_t sample_cap) {
  if (sample_cap > 0) return false;
  if (g_march_mode >= 0) return g_march_mode != 0;
  return N <= 32768;
}
int inr_set_march_mode(int32_t mode) {
  INR_REQUIRE(mode >= -1 && mode <= 1, "mode must be -1 (automatic), 0 (lane per ray) or 1 (wave per ray)");
  g_march_mode = mode;
  return INR_OK;
}

static int64_t ws_ints(int64_t N) {
  const int64_t nb = (N + kRayBlock - 1) / kRayBlock;
  return ((N + nb + 64 + 1) / 2) * 2;     // even: the capture rows that follow are float2
}
// sample_cap: number of step candidates per ray the count pass records as a bit mask (rounded up to 32; 0 = none)
static int cap_words_of(int32_t sample_cap) { return sample_cap > 0 ? (sample_cap + 31) / 32 : 0; }
// Staged wave-per-ray marcher (training batches): rows of 5 floats, kStagePitch rows reserved per ray.  Address space,
// not traffic: 4096 rays reserve 84 MB of the 288 GB and write ~4 MB of it.
constexpr int kStagePitch = 1024;
constexpr int64_t kStageMaxRays = 8192;
static bool use_stage(int64_t N, int32_t sample_cap, int max_steps) {
  return use_coop(N, sample_cap) && N <= kStageMaxRays && max_steps <= kStagePitch;
}
static int64_t stage_offset_bytes(int64_t N, int32_t sample_cap) {          // 16-byte aligned start of the stage rows
  const int w = cap_words_of(sample_cap);
  const int64_t b = ws_ints(N) * (int64_t)sizeof(int32_t) + (w ? N * (int64_t)(w + 1) * (int64_t)sizeof(uint32_t) : 0);
  return (b + 15) / 16 * 16;
}
int inr_march_write_fills_unowned_rows(int64_t N, int32_t sample_cap, int32_t max_steps) {
  return use_stage(N, sample_cap, max_steps) ? 1 : 0;
}
int64_t inr_march_workspace_bytes(int64_t N, int32_t sample_cap) {
  const int64_t base = stage_offset_bytes(N, sample_cap);
  return base + (N <= kStageMaxRays && sample_cap == 0 ? N * (int64_t)kStagePitch * 5 * (int64_t)sizeof(float) : 0);
}

// extra dynamic LDS per march workgroup while overlap placement is on (inr_set_overlap_placement)
static size_t march_lds_pad() { return (size_t)inr::g_march_lds_pad; }

int inr_march_rays_train_count(const float* rays_o, const float* rays_d, const uint8_t* bitfield, float bound,
                               float dt_gamma, int32_t max_steps, int64_t N, int32_t cascade, int32_t H,
                               const float* nears, const float* fars, const float* noises, int32_t* rays,
                               int32_t* counter, void* workspace, int32_t sample_cap, inr_stream_t s) {
  INR_REQUIRE(rays_o && rays_d && bitfield && nears && fars && rays && counter && workspace, "null pointer");
  INR_REQUIRE(N > 0 && N < (1ll << 31), "N out of range");
  INR_REQUIRE(max_steps > 0 && cascade >= 1 && H >= 8 && H <= 1024, "bad grid parameters");
  const MarchParams P = make_params(bitfield, bound, dt_gamma, max_steps, cascade, H);
  const unsigned nb = blocks_for(N, kRayBlock);
  INR_REQUIRE(sample_cap >= 0 && ((uintptr_t)workspace & 7) == 0, "workspace must be 8-byte aligned");
  int32_t* counts = reinterpret_cast<int32_t*>(workspace);
  int32_t* block_sums = counts + N;
  uint32_t* mask = reinterpret_cast<uint32_t*>(counts + ws_ints(N));
  hipStream_t st = as_stream(s);
  if (use_coop(N, sample_cap)) {
    float* stage = use_stage(N, sample_cap, max_steps)
                       ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + stage_offset_bytes(N, sample_cap)) : nullptr;
    k_march_count_coop<<<blocks_for(N, kCoopRaysPerBlock), kCoopRaysPerBlock * 64, 0, st>>>(P, rays_o, rays_d, N, max_steps,
                                                                                          nears, fars, noises, counts, stage,
                                                                                          kStagePitch);
    k_scan_counts<<<1, 1024, 0, st>>>(counts, N, rays, counter);
    return check_launch("march_rays_train_count");
  }
  k_march_count<<<nb, kRayBlock, march_lds_pad(), st>>>(P, rays_o, rays_d, N, max_steps, nears, fars, noises, counts, block_sums,
                                          mask, cap_words_of(sample_cap));
  k_scan_block_sums<<<1, 1024, 0, st>>>(block_sums, (int)nb, N, counter);
  k_finalize_offsets<<<nb, kRayBlock, 0, st>>>(counts, block_sums, N, rays);
  return check_launch("march_rays_train_count");
}

int inr_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* bitfield, float bound,
                               float dt_gamma, int32_t max_steps, int64_t N, int32_t cascade, int32_t H, int64_t M,
                               const float* nears, const float* fars, const float* noises, const int32_t* rays,
                               float* xyzs, float* dirs, float* deltas, const void* workspace, int32_t sample_cap,
                               inr_stream_t s) {
  INR_REQUIRE(rays_o && rays_d && bitfield && nears && fars && rays, "null pointer");
  INR_REQUIRE(N > 0 && M >= 0, "bad sizes");
  if (M == 0) return INR_OK;
  INR_REQUIRE(M == 0 || (xyzs && dirs && deltas), "null output");
  const MarchParams P = make_params(bitfield, bound, dt_gamma, max_steps, cascade, H);
  INR_REQUIRE(sample_cap == 0 || workspace, "sample_cap > 0 needs the workspace of the count pass");
  const uint32_t* mask = sample_cap > 0 ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const int32_t*>(workspace) + ws_ints(N)) : nullptr;
  if (use_stage(N, sample_cap, max_steps)) {     // the count pass parked the samples in the workspace
    INR_REQUIRE(workspace, "the staged marcher needs the workspace of the count pass");
    const float* stage = reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + stage_offset_bytes(N, sample_cap));
    k_march_write_staged<<<blocks_for(N, kCoopRaysPerBlock), kCoopRaysPerBlock * 64, 0, as_stream(s)>>>(
        rays_d, N, M, rays, stage, kStagePitch, xyzs, dirs, deltas);
    return check_launch("march_rays_train_write");
  }
  if (use_coop(N, sample_cap)) {
    k_march_write_coop<<<blocks_for(N, kCoopRaysPerBlock), kCoopRaysPerBlock * 64, 0, as_stream(s)>>>(
        P, rays_o, rays_d, N, M, nears, fars, noises, rays, xyzs, dirs, deltas);
    return check_launch("march_rays_train_write");
  }
  k_march_write<<<blocks_for(N, kRayBlock), kRayBlock, 0, as_stream(s)>>>(P, rays_o, rays_d, N, M, nears, fars, noises,
                                                                          rays, xyzs, dirs, deltas, mask, cap_words_of(sample_cap));
  return check_launch("march_rays_train_write");
}

int inr_march_rays_patch_write(const float* rays_o, const float* rays_d, const uint8_t* bitfield, float bound,
                               float dt_gamma, int32_t max_steps, int64_t N, int32_t cascade, int32_t H, int64_t M,
                               const float* nears, const float* fars, const float* noises, const int32_t* rays,
                               float* xyzs, float* dirs, float* deltas, const void* workspace, int32_t sample_cap,
                               int32_t* ray_ids, int32_t normalise, inr_stream_t s) {
  INR_REQUIRE(rays_o && rays_d && bitfield && nears && fars && rays, "null pointer");
  INR_REQUIRE(N > 0 && M >= 0, "bad sizes");
  if (M == 0) return INR_OK;
  INR_REQUIRE(M == 0 || (xyzs && deltas && (dirs || ray_ids)), "null output");
  const MarchParams P = make_params(bitfield, bound, dt_gamma, max_steps, cascade, H);
  INR_REQUIRE(sample_cap == 0 || workspace, "sample_cap > 0 needs the workspace of the count pass");
  const uint32_t* mask = sample_cap > 0 ? reinterpret_cast<const uint32_t*>(reinterpret_cast<const int32_t*>(workspace) + ws_ints(N)) : nullptr;
  k_march_write_patch<<<blocks_for(N, kRayBlock), kRayBlock, march_lds_pad(), as_stream(s)>>>(P, rays_o, rays_d, N, M, nears, fars,
                                                                                noises, rays, xyzs, dirs, deltas, mask,
                                                                                cap_words_of(sample_cap), ray_ids, normalise);
  return check_launch("march_rays_patch_write");
}

int inr_composite_rays_patch_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                     int64_t N, int64_t M, float T_thresh, const float* extra, int32_t K,
                                     float* weights_sum, float* depth, float* image, float* extra_out,
                                     float* weights, uint64_t* skippable, inr_stream_t s) {
  INR_REQUIRE(N >= 0 && M >= 0, "bad sizes");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays && weights_sum && depth && image, "null pointer");
  INR_REQUIRE(M == 0 || (sigmas && rgbs && deltas), "null sample arrays");
  INR_REQUIRE(!extra || (extra_out && weights && K > 0 && K <= 64), "extra needs extra_out, weights and 0 < K <= 64");
  INR_REQUIRE(((uintptr_t)deltas & 7) == 0, "deltas must be 8-byte aligned");
  hipStream_t st = as_stream(s);
  k_composite_patch_fwd<<<blocks_for(N, kRayBlock), kRayBlock, 0, st>>>(sigmas, rgbs, deltas, rays, N, M, T_thresh,
                                                                        weights, weights_sum, depth, image,
                                                                        reinterpret_cast<unsigned long long*>(skippable));
  if (extra_out && K > 0)     // also with no sample at all (M == 0, extra null): the rows of extra_out must be zeroed
    k_composite_patch_extra<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(weights, extra, rays, N, M, K,
                                                                                 extra_out);
  return check_launch("composite_rays_patch_forward");
}

int inr_project_masks_patch(const float* xyzs, const float* weights, const int32_t* rays, int64_t N, int64_t M,
                            const uint32_t* mask_words, int32_t W, int32_t L, int32_t H, const float* bbox /*host: lo[3], hi[3]*/,
                            int32_t k_total, int32_t k_base, int32_t k_count, float* out, inr_stream_t s) {
  INR_REQUIRE(N >= 0 && M >= 0 && W > 0 && L > 0 && H > 0, "bad sizes");
  INR_REQUIRE(k_total > 0 && k_base >= 0 && k_count > 0 && k_count <= 32 && k_base + k_count <= k_total, "bad mask range");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays && mask_words && bbox && out, "null pointer");
  INR_REQUIRE(M == 0 || (xyzs && weights), "null sample arrays");
  MaskVolume V;
  for (int a = 0; a < 3; ++a) { V.lo[a] = bbox[a]; V.hi[a] = bbox[3 + a]; }
  V.res[0] = (float)W; V.res[1] = (float)L; V.res[2] = (float)H;
  V.W = W; V.L = L; V.H = H;
  k_project_masks_patch<<<blocks_for(N, kRayBlock), kRayBlock, 0, as_stream(s)>>>(xyzs, weights, rays, N, M, mask_words, V,
                                                                                 k_total, k_base, k_count, out);
  return check_launch("project_masks_patch");
}

int inr_march_rays(int64_t n_alive, int32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma, int32_t max_steps,
                   int32_t cascade, int32_t H, const uint8_t* bitfield, const float* nears, const float* fars,
                   float* xyzs, float* dirs, float* deltas, inr_stream_t s) {
  (void)nears;
  INR_REQUIRE(rays_alive && rays_t && rays_o && rays_d && bitfield && fars && xyzs && dirs && deltas, "null pointer");
  INR_REQUIRE(n_alive >= 0 && n_step > 0, "bad sizes");
  if (n_alive == 0) return INR_OK;
  const MarchParams P = make_params(bitfield, bound, dt_gamma, max_steps, cascade, H);
  k_march_rays<<<blocks_for(n_alive, kRayBlock), kRayBlock, 0, as_stream(s)>>>(P, n_alive, n_step, rays_alive, rays_t,
                                                                               rays_o, rays_d, fars, xyzs, dirs, deltas);
  return check_launch("march_rays");
}

int inr_composite_rays(int64_t n_alive, int32_t n_step, int32_t* rays_alive, float* rays_t, const float* sigmas,
                       const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                       float T_thresh, const float* extra, float* extra_acc, int32_t K, inr_stream_t s) {
  INR_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image, "null pointer");
  INR_REQUIRE(!extra || (extra_acc && K > 0), "extra given without extra_acc/K");
  if (n_alive == 0) return INR_OK;
  k_composite_rays<<<blocks_for(n_alive, kRayBlock), kRayBlock, 0, as_stream(s)>>>(
      n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh, extra, extra_acc, K);
  return check_launch("composite_rays");
}

int inr_compact_alive(const int32_t* rays_alive, int64_t n_alive, int32_t* out, int32_t* n_out, inr_stream_t s) {
  // uses out[n_alive .. ] ? no: needs scratch; the caller passes `out` with room for
  // n_alive + inr_march_workspace_bytes(n_alive)/4 int32 (compacted list first, scratch after).
  INR_REQUIRE(rays_alive && out && n_out && n_alive >= 0, "bad argument");
  if (n_alive == 0) {
    (void)hipMemsetAsync(n_out, 0, 2 * sizeof(int32_t), as_stream(s));
    return check_launch("compact_alive");
  }
  const unsigned nb = blocks_for(n_alive, kRayBlock);
  int32_t* counts = out + n_alive;
  int32_t* block_sums = counts + n_alive;
  hipStream_t st = as_stream(s);
  k_alive_flags<<<nb, kRayBlock, 0, st>>>(rays_alive, n_alive, counts, block_sums);
  k_scan_block_sums<<<1, 1024, 0, st>>>(block_sums, (int)nb, n_alive, n_out);
  k_alive_scatter<<<nb, kRayBlock, 0, st>>>(rays_alive, n_alive, block_sums, out);
  return check_launch("compact_alive");
}

int inr_cross_entropy(const float* logits, const int64_t* labels, int64_t N, int32_t K, int64_t ignore_index,
                      float* grad_logits, float* acc, float* loss, inr_stream_t s) {
  INR_REQUIRE(N >= 0 && K > 0 && K <= 64, "K must be in 1..64");
  INR_REQUIRE(acc && loss && ((uintptr_t)acc & 7) == 0, "null or misaligned pointer");
  hipStream_t st = as_stream(s);
  int nb = 0;
  if (N > 0) {
    INR_REQUIRE(logits && labels && grad_logits, "null pointer");
    nb = (int)std::min<int64_t>((N + 3) / 4, kCeBlocks);
    k_ce_rows<<<nb, 256, 0, st>>>(logits, labels, N, K, ignore_index, grad_logits, reinterpret_cast<float2*>(acc));
  }
  const unsigned nb2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>((N * K + 255) / 256, (int64_t)cu_count() * 4));
  k_ce_scale<<<nb2, 256, 0, st>>>(grad_logits, N * K, reinterpret_cast<const float2*>(acc), nb, loss);
  return check_launch("cross_entropy");
}

int inr_finish_rays(const float* image, const float* depth, const float* weights_sum, const float* nears,
                    const float* fars, const float* t0, float bg_r, float bg_g, float bg_b, int64_t N, float* image_out,
                    float* depth_out, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(image && depth && weights_sum && nears && fars && image_out && depth_out, "null pointer");
  k_finish_rays<<<blocks_for(N, 256), 256, 0, as_stream(s)>>>(image, depth, weights_sum, nears, fars, t0, bg_r, bg_g, bg_b,
                                                              N, image_out, depth_out);
  return check_launch("finish_rays");
}

int inr_finish_rays_mse(const float* image, const float* depth, const float* weights_sum, const float* nears,
                        const float* fars, float bg_r, float bg_g, float bg_b, const float* bg_rays, const float* target,
                        int64_t N, float* image_out, float* depth_out, float* grad, float* loss, inr_stream_t s) {
  INR_REQUIRE(N > 0 && N <= INR_FINISH_MSE_MAX_RAYS, "N out of range (one workgroup: 0 < N <= INR_FINISH_MSE_MAX_RAYS)");
  INR_REQUIRE(image && depth && weights_sum && nears && fars && target && image_out && depth_out && grad && loss,
              "null pointer");
  k_finish_rays_mse<<<1, kFinishMseThreads, 0, as_stream(s)>>>(image, depth, weights_sum, nears, fars, bg_r, bg_g, bg_b,
                                                               bg_rays, target, N, image_out, depth_out, grad, loss);
  return check_launch("finish_rays_mse");
}

int inr_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                     int64_t N, int64_t M, float T_thresh, const float* extra, int32_t K,
                                     float* weights_sum, float* depth, float* image, float* extra_out, float* weights,
                                     int32_t* sample_ray, inr_stream_t s) {
  INR_REQUIRE(rays && weights_sum && depth && image && N >= 0 && M >= 0, "bad argument");
  INR_REQUIRE(!sample_ray || weights, "sample_ray needs weights");
  INR_REQUIRE(!extra || (extra_out && weights && K > 0 && K <= 64), "extra needs extra_out, weights and 0 < K <= 64");
  INR_REQUIRE(!extra_out || M == 0 || (extra && weights), "extra_out needs extra and weights");
  if (N == 0) return INR_OK;
  INR_REQUIRE(M == 0 || (sigmas && rgbs && deltas), "null sample arrays");   // M == 0: every ray is dropped
  INR_REQUIRE(((uintptr_t)deltas & 7) == 0, "deltas must be 8-byte aligned");
  hipStream_t st = as_stream(s);
  k_composite_train_fwd<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(sigmas, rgbs, deltas, rays, N, M, T_thresh,
                                                                             weights_sum, depth, image, weights, sample_ray);
  if (extra_out && K > 0)     // also with no sample at all (M == 0, extra null): the rows of extra_out must be zeroed
    k_composite_train_extra_fwd<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(weights, extra, rays, N, M, K,
                                                                                     extra_out, CeArgs{});
  return check_launch("composite_rays_train_forward");
}

int inr_composite_rays_extra_forward(const float* weights, const float* extra, const int32_t* rays, int64_t N, int64_t M,
                                     int32_t K, float* extra_out, const int64_t* labels, int32_t n_classes,
                                     int64_t ignore_index, float* grad_pix, void* workspace, float* loss_out,
                                     inr_stream_t s) {
  INR_REQUIRE(rays && extra_out && N >= 0 && M >= 0 && K > 0 && K <= 64, "bad argument");
  INR_REQUIRE(!labels || (grad_pix && workspace && loss_out && n_classes > 0 && n_classes <= K),
              "labels need grad_pix, workspace, loss_out and 0 < n_classes <= K");
  INR_REQUIRE(((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
  if (N == 0) return INR_OK;
  INR_REQUIRE(M == 0 || (weights && extra), "null sample arrays");
  hipStream_t st = as_stream(s);
  CeArgs ce{};
  if (labels) ce = CeArgs{labels, n_classes, ignore_index, grad_pix, reinterpret_cast<float4*>(workspace)};
  k_composite_train_extra_fwd<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(weights, extra, rays, N, M, K, extra_out, ce);
  if (labels) k_ce_finalize<<<1, 1024, 0, st>>>(reinterpret_cast<const float4*>(workspace), N, loss_out);
  return check_launch("composite_rays_extra_forward");
}

int inr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                      const float* grad_extra_out, const float* sigmas, const float* rgbs,
                                      const float* extra, const float* deltas, const int32_t* rays,
                                      const float* weights_sum, const float* image, const float* weights, int64_t N,
                                      int64_t M, float T_thresh, int32_t K, float* grad_sigmas, float* grad_rgbs,
                                      float* grad_extra, const int32_t* total_dev, inr_stream_t s) {
  (void)extra;
  INR_REQUIRE(rays && N >= 0 && M >= 0, "bad argument");
  INR_REQUIRE((grad_sigmas != nullptr) == (grad_rgbs != nullptr), "grad_sigmas and grad_rgbs go together");
  INR_REQUIRE(!grad_sigmas || (grad_image && weights_sum && image && (M == 0 || (sigmas && rgbs && deltas))), "null pointer");
  INR_REQUIRE(!grad_extra_out || M == 0 || (grad_extra && weights && K > 0 && K <= 64),
              "grad_extra_out needs grad_extra, weights and 0 < K <= 64");
  if (N == 0 || M == 0) return INR_OK;       // no sample: nothing to write (every ray is dropped)
  hipStream_t st = as_stream(s);
  if (grad_sigmas)   // null: the density/colour field is frozen (instance stage) - only the K channels flow back
    k_composite_train_bwd<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(grad_weights_sum, grad_image, sigmas,
                                                                               rgbs, deltas, rays, weights_sum, image, N, M,
                                                                               T_thresh, grad_sigmas, grad_rgbs, total_dev);
  if (grad_extra_out)
    k_composite_train_extra_bwd<<<blocks_for(N * 64, kRayBlock), kRayBlock, 0, st>>>(grad_extra_out, weights, rays, N,
                                                                                     M, K, grad_extra);
  return check_launch("composite_rays_train_backward");
}

}  // extern "C"
