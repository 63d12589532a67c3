// Stand-alone encoders and the optimiser sweep for gfx950:
//   hash-grid forward / backward, SH-4 forward / backward (a10), fused Adam (a15).
// Compiled with -ffp-contract=off so that cell selection (floor of x01*scale+0.5) is the same
// decision the CPU oracle takes; value blending uses explicit fmaf where it pays.
//
// These are the API-parity kernels (gridencoder / shencoder modules).  The render hot path
// uses the fused kernel in field_fused.hip, which shares grid_common.h with this file.
#include "common.h"
#include "grid_common.h"

namespace inr {

// One lane per (sample, level): 16 consecutive lanes write one sample's 128-byte feature row,
// so the [M, L*F] output is written fully coalesced; gathers are 8 B (float2) per corner.
__global__ void __launch_bounds__(256) k_grid_fwd(const float* __restrict__ x, const float2* __restrict__ emb,
                                                  GridDesc G, int64_t M, float bound, float2* __restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int L = G.num_levels;
  const int64_t m = tid / L;
  const int l = (int)(tid - m * L);
  if (m >= M) return;
  const float rb = 2.0f * bound;
  const float x0 = (x[m * 3 + 0] + bound) / rb;
  const float x1 = (x[m * 3 + 1] + bound) / rb;
  const float x2 = (x[m * 3 + 2] + bound) / rb;
  if (oob01(x0, x1, x2)) {             // outside the grid: zero features (upstream's flag_oob)
    out[m * L + l] = make_float2(0.f, 0.f);
    return;
  }
  Cell c;
  locate(G, l, x0, x1, x2, c);
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float w = corner_weight(c, k);
    const float2 v = emb[G.offsets[l] + corner_index(G, l, c, k)];
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
  }
  out[m * L + l] = acc;
}

// Scatter-add of w * dL/dy into the table gradient (hardware fp32 atomics: -munsafe-fp-atomics ->
// global_atomic_add_f32).  Measured on MI355X (tools/micro/atomic_bench.hip): scattered atomics run at
// 21 G/s, address-adjacent ones at 150 G/s.  So: ONE LEVEL PER WAVE (blockIdx.y), lanes = consecutive
// samples.  Training samples are ray-major, hence neighbouring lanes sit in the same or the next cell:
// their rows are equal or adjacent (x is the fastest index for dense AND hashed levels), atomics of one
// instruction fall into few 128-B lines, and runs of EQUAL rows are first summed inside the wave by a
// segmented scan so that only the last lane of a run issues the atomic (level 0: ~37 samples per cell).
// Lane layout: 4 lanes per sample = (x side xb) x (feature f); a wave holds 16 consecutive samples of
// ONE level.  For a (y,z) corner the four lanes of a sample address 2 rows x 2 features; x-neighbour rows
// are adjacent 15 times out of 16 (x is the fastest index in dense and hashed levels), so the four
// atomics of one instruction fall into one 16-byte piece of one line instead of four separate
// instructions each touching its own line.  Equal addresses of consecutive samples (same cell) are
// first summed inside the wave (segmented scan over lanes 4 apart); only the last lane of a run issues
// the atomic.
constexpr int kFxBlocks = 256;      // workgroups per level of the finishing pass
constexpr int kFxFlags = 64;        // state layout: [64, 80) non-finite flags, [80, 96) peak use of the int32 range,
constexpr int kFxPeak = 80;         // [96 + 256 l + b] block maxima
constexpr int kFxBase = 96;

// Fixed-point form (round 6, `scale` > 0): the run's sum is rounded to a multiple of 1 / scale and added as an INT32
// (global_atomic_add instead of global_atomic_add_f32).  Every atomic type is forwarded to the memory-side unit, but
// the unit takes integer adds at 26.9 G requests/s against 21.0 for fp32 (16-byte requests into a 49 MB table:
// tools/micro/atomic_type_bench.hip, profiles/r06_atomic_type_bench.txt) - and integer addition is associative, so the
// table gradient no longer depends on the order in which the waves' requests arrive: training steps become
// bit-reproducible.  scale == 0: the fp32 atomics of rounds 1-5.
// A non-finite run sum cannot be represented (float -> int saturates, NaN becomes 0): it raises the level's flag instead
// and the finishing pass turns the WHOLE level's gradient into NaN - as loud as the NaN rows fp32 atomics would leave.
// Rounding is to nearest.  (A stochastic rounding with a deterministic draw - unbiased: contributions below the quantum
// arrive in expectation - was built and measured in round 6: it injects +-1-quantum spikes into rows whose contributions
// cancel, which Adam (eps 1e-15) turns into full steps: trained-scene PSNR 27.8 -> 26.6 dB.  Removed.)
// 64-bit form (`acc64` non-null): the same sums in int64 with a scale 2^32 larger - a quantum of ~2e-16 of the level's recent
// maximum, below anything Adam (eps 1e-15) can see: the faithful, order-independent form.  The unit takes 8-byte integer
// adds at 23.6 G requests/s (fp32: 21.0, int32: 26.9).
__device__ __forceinline__ void run_reduce_atomic4(float* __restrict__ gemb, uint32_t addr, bool valid, float v, float scale,
                                                   float* __restrict__ bad_flag, long long* __restrict__ acc64) {
  const int lane = threadIdx.x & 63;
  const uint32_t key = valid ? addr : 0xFFFFFFFFu - (uint32_t)lane;     // invalid lanes never join a run
  const uint32_t prev = __shfl_up(key, 4, 64);
  const uint32_t next = __shfl_down(key, 4, 64);
  bool head = (lane < 4) || (prev != key);          // first sample of a run of equal addresses
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) {                // segmented inclusive scan along lanes 4 apart
    const float u = __shfl_up(v, d, 64);
    const bool uh = __shfl_up((int)head, d, 64) != 0;
    if (lane >= d && !head) {
      v += u;
      head = uh;
    }
  }
  const bool tail = (lane >= 60) || (next != key);
  if (scale > 0.0f) {
    if (acc64) {
      const long long q = __float2ll_rn(v * scale);
      if (valid && tail && q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(acc64) + addr, (unsigned long long)q);
    } else {
      const int q = __float2int_rn(v * scale);
      if (valid && tail && q != 0) atomicAdd(reinterpret_cast<int*>(gemb) + addr, q);
    }
    if (valid && tail && !(fabsf(v) <= 3.402823466e+38f)) *bad_flag = 1.0f;      // (idempotent plain store)
  } else if (valid && tail && v != 0.0f) {
    atomicAdd(gemb + addr, v);
  }
}

// `order` (nullable): a permutation of the samples; sample slot m processes sample order[m].
__global__ void __launch_bounds__(256) k_grid_bwd(const float* __restrict__ x, const float* __restrict__ gout,
                                                  const int32_t* __restrict__ order, GridDesc G, int64_t M, float bound,
                                                  float* __restrict__ gemb, int level0, float* __restrict__ fx,
                                                  long long* __restrict__ acc64) {
  // blockIdx.x = level (the FAST axis of the dispatch order), blockIdx.y = block of 64 samples: at any moment all
  // levels of a window of samples are in flight.  With the level on the slow axis (rounds 1-2) the coarse levels ran
  // alone at the start of the launch, and they are not throughput- but LATENCY-bound: all samples hit the same few
  // hundred 64-byte chunks (one memory-side atomic request each, ~0.4 us, same-chunk requests one after the other -
  // level 0 alone takes 36 us for 1 % of the requests, profiles/r03_NOTES.txt), while the hashed fine levels are bound
  // by the request RATE (~21 G 64-byte requests/s).  Interleaved, the chains of the coarse levels hide under the
  // stream of fine-level requests.
  const int64_t m = ((int64_t)blockIdx.y * blockDim.x + threadIdx.x) >> 2;
  const int sub = threadIdx.x & 3, xb = sub >> 1, f = sub & 1;
  const int l = level0 + blockIdx.x;
  const int L = G.num_levels;
  bool valid = m < M;
  const int64_t ms = valid ? m : M - 1;
  const int64_t mc = order ? (int64_t)order[ms] : ms;
  const float rb = 2.0f * bound;
  float x0 = (x[mc * 3 + 0] + bound) / rb;
  float x1 = (x[mc * 3 + 1] + bound) / rb;
  float x2 = (x[mc * 3 + 2] + bound) / rb;
  if (oob01(x0, x1, x2)) {             // outside the grid: no gradient (upstream's flag_oob); all lanes stay in
    valid = false;                     // the wave for the segmented scan
    x0 = x1 = x2 = 0.0f;
  }
  Cell c;
  locate(G, l, x0, x1, x2, c);
  const float g = valid ? gout[(mc * L + l) * 2 + f] : 0.f;
  const uint32_t base = G.offsets[l];
  const float fx_scale = fx ? fx[l] : 0.0f;         // uniform per workgroup: this level's fixed-point scale (0 = fp32 atomics)
#pragma unroll
  for (int yz = 0; yz < 4; ++yz) {
    const int k = xb | (yz << 1);                   // corner: bit 0 = x side, bits 1,2 = y, z sides
    const float w = corner_weight(c, k);
    const uint32_t row = base + corner_index(G, l, c, k);
    run_reduce_atomic4(gemb, 2u * row + (uint32_t)f, valid, w * g, fx_scale, fx + kFxFlags + l, acc64);
  }
}

// ---- fixed-point table gradient: state, finishing pass, scale update ----------------------------------------------
// State (device floats, INR_GRID_FX_STATE_FLOATS): [0,16) scale of each level for THIS step (a power of two; 0 = the level
// is scattered with fp32 atomics), [16,32) reference magnitude (a slowly decaying maximum of the level's largest |row
// gradient|), [32,48) this step's maximum, [48] steps with at least one fixed-point level, [49] near misses,
// [64,80) "a non-finite contribution was seen" per level, [80,96) the largest fraction of the int32 range a row sum of
// the level has used so far (1 = wrapped), [96 + 256 l + b] maximum seen by workgroup b of the finishing pass of level l.

// In place over the rows of levels [level0, level0 + gridDim.y): int32 sums -> fp32 gradients (levels with a scale), and
// the level's largest |gradient| into the workgroup's slot (all levels: the fp32 levels need it to get a scale).
__global__ void __launch_bounds__(256) k_grid_grad_finish(float* __restrict__ gemb, GridDesc G, int level0, float* __restrict__ fx) {
  __shared__ float red[4];
  const int l = level0 + blockIdx.y;
  const float scale = fx[l];
  const bool poisoned = fx[kFxFlags + l] != 0.0f;                 // a non-finite contribution in a fixed-point level
  const float inv = scale > 0.0f ? 1.0f / scale : 0.0f;          // scale is a power of two: exact
  const size_t lo = (size_t)G.offsets[l] * 2, hi = (size_t)G.offsets[l + 1] * 2;       // floats; multiples of 16
  float4* p = reinterpret_cast<float4*>(gemb + lo);
  const size_t n4 = (hi - lo) >> 2;
  float mx = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i];
    if (scale > 0.0f) {
      v.x = (float)__float_as_int(v.x) * inv; v.y = (float)__float_as_int(v.y) * inv;
      v.z = (float)__float_as_int(v.z) * inv; v.w = (float)__float_as_int(v.w) * inv;
      if (poisoned) v.x = v.y = v.z = v.w = __int_as_float(0x7FC00000);
      p[i] = v;
    }
    // NaN / Inf gradients must reach the update as "not finite": fmaxf would drop a NaN
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    const bool bad = !(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w);
    mx = bad ? __int_as_float(0x7F800000) : fmaxf(mx, a);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) fx[kFxBase + kFxBlocks * l + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// The 64-bit form's finishing pass: acc64 rows (two int64 sums) -> fp32 gradient rows of gemb, acc64 zeroed for the next
// step, the level's maximum into the workgroup's slot.  A level without a scale was scattered with fp32 atomics straight
// into gemb: only its maximum is taken.
__global__ void __launch_bounds__(256) k_grid_grad_finish64(long long* __restrict__ acc64, float* __restrict__ gemb, GridDesc G,
                                                            int level0, float* __restrict__ fx) {
  __shared__ float red[4];
  const int l = level0 + blockIdx.y;
  const float scale = fx[l];
  const bool poisoned = fx[kFxFlags + l] != 0.0f;
  const double inv = scale > 0.0f ? 1.0 / (double)scale : 0.0;
  const size_t lo = G.offsets[l], hi = G.offsets[l + 1];                    // rows
  float mx = 0.0f;
  for (size_t r = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += (size_t)gridDim.x * blockDim.x) {
    float2 v;
    if (scale > 0.0f) {
      longlong2* p = reinterpret_cast<longlong2*>(acc64) + r;
      const longlong2 q = *p;
      *p = make_longlong2(0, 0);
      v.x = (float)((double)q.x * inv);
      v.y = (float)((double)q.y * inv);
      if (poisoned) v.x = v.y = __int_as_float(0x7FC00000);
      reinterpret_cast<float2*>(gemb)[r] = v;
    } else {
      v = reinterpret_cast<const float2*>(gemb)[r];
    }
    const bool bad = !(v.x == v.x) || !(v.y == v.y);
    mx = bad ? __int_as_float(0x7F800000) : fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y)));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) fx[kFxBase + kFxBlocks * l + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One workgroup, one wave per level: this step's maximum -> the level's scale for the NEXT step.
//   ref   = max(this step's max, 0.97 ref)                      (a slowly decaying maximum: half-life 23 steps)
//   scale = 2^floor(log2(2^30 / (headroom * ref)))              (so that headroom x the reference still fits in 31 bits)
// Why so slow and why 128 (the product's headroom): the per-level maxima are heavy-tailed from step to step
// (tools/fx_dynamics_probe.py, 1500 steps per stage: 99.9th percentile of the growth 26x / 152x, largest 90x / 766x in
// the NeRF / instance stage - one batch with a cluster of misclassified rays).  Against a reference that remembers the
// last ~50 steps the largest row sum of those runs used 0.06 of the int32 range; with the first cut (decay 0.75,
// headroom 64) it used 0.81.  The quantum is then ~5e-7 of a typical step's maximum.
// A level runs on fp32 atomics only while it has no reference (before the first step - the host primes it - and after an
// all-zero or non-finite gradient).  A NEAR MISS - a step whose maximum used more than 1/8 of the range - is counted and
// the peak use per level is kept ([80,96) of the state): the bench record reports both.  Only a row's FINAL sum must
// fit: int32 addition is modular, intermediate overflow cancels.  Should a row ever wrap, its garbage maximum makes the
// next scale coarser, never finer.
__global__ void __launch_bounds__(1024) k_grid_fx_update(float* __restrict__ fx, int num_levels, float headroom, float range) {
  // range = 2^31 (int32 sums) or 2^63 (int64 sums): `headroom x reference x scale <= range / 2`
  const int l = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (l >= num_levels) return;
  float m = 0.0f;
  for (int b = lane; b < kFxBlocks; b += 64) m = fmaxf(m, fx[kFxBase + kFxBlocks * l + b]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if (lane != 0) return;
  const float old_scale = fx[l], old_ref = fx[16 + l];
  const bool finite = m < __int_as_float(0x7F800000);
  const bool near_miss = old_scale > 0.0f && finite && m * old_scale > 0.125f * range;
  float ref = finite ? fmaxf(m, 0.97f * old_ref) : 0.0f;
  float scale = 0.0f;
  if (ref > 0.0f && finite) {
    float e = floorf(log2f(0.5f * range / (headroom * ref)));
    e = fminf(fmaxf(e, -100.0f), 100.0f);
    scale = exp2f(e);
  }
  fx[l] = scale;
  fx[16 + l] = ref;
  fx[32 + l] = m;
  fx[kFxFlags + l] = 0.0f;
  if (old_scale > 0.0f && finite) fx[kFxPeak + l] = fmaxf(fx[kFxPeak + l], m * old_scale / range);
  if (near_miss) atomicAdd(fx + 49, 1.0f);
  if (l == 0 && old_scale > 0.0f) fx[48] += 1.0f;
}

// Gradient with respect to the INPUT coordinates (upstream's dy_dx path, used when the positions require grad):
// feature_l,f = sum_k w_k(fx,fy,fz) v_k  =>  d/dx = scale_l / (2 bound) * sum_k (dw_k/dfx) v_k, dw/dfx = +-(wy*wz).
// One lane per (sample, level); the 16 levels of a sample sit in 16 adjacent lanes and meet in a butterfly.
__global__ void __launch_bounds__(256) k_grid_bwd_input(const float* __restrict__ x, const float2* __restrict__ gout,
                                                        const float2* __restrict__ emb, GridDesc G, int64_t M, float bound,
                                                        float* __restrict__ gx) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t m = tid >> 4;
  const int l = (int)(tid & 15);
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (m < M && l < G.num_levels) {
    const float rb = 2.0f * bound;
    const float x0 = (x[m * 3 + 0] + bound) / rb, x1 = (x[m * 3 + 1] + bound) / rb, x2 = (x[m * 3 + 2] + bound) / rb;
    if (!oob01(x0, x1, x2)) {
      Cell c;
      locate(G, l, x0, x1, x2, c);
      const float2 go = gout[m * G.num_levels + l];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float2 v = emb[G.offsets[l] + corner_index(G, l, c, k)];
        const float d = go.x * v.x + go.y * v.y;
        const float wx = (k & 1) ? c.fx : 1.0f - c.fx, wy = (k & 2) ? c.fy : 1.0f - c.fy, wz = (k & 4) ? c.fz : 1.0f - c.fz;
        g0 += ((k & 1) ? 1.0f : -1.0f) * wy * wz * d;
        g1 += ((k & 2) ? 1.0f : -1.0f) * wx * wz * d;
        g2 += ((k & 4) ? 1.0f : -1.0f) * wx * wy * d;
      }
      const float sc = G.scales[l] / rb;
      g0 *= sc; g1 *= sc; g2 *= sc;
    }
  }
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    g0 += __shfl_xor(g0, d, 64); g1 += __shfl_xor(g1, d, 64); g2 += __shfl_xor(g2, d, 64);
  }
  if (l == 0 && m < M) { gx[m * 3 + 0] = g0; gx[m * 3 + 1] = g1; gx[m * 3 + 2] = g2; }
}

// ---- SH ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sh_fwd(const float* __restrict__ d, int64_t M, int degree,
                                                float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float v[16];
  sh4(d[m * 3], d[m * 3 + 1], d[m * 3 + 2], v);
  const int C = degree * degree;
  for (int i = 0; i < C; ++i) out[m * C + i] = v[i];
}

// per-ray SH table in the lane order of the fused field kernel: out[n][q][ks] = sh[4*ks + q]
__global__ void __launch_bounds__(256) k_sh_table_q(const float* __restrict__ d, int64_t N, float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float v[16];
  sh4(d[n * 3], d[n * 3 + 1], d[n * 3 + 2], v);
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) out[n * 16 + q * 4 + ks] = v[4 * ks + q];
}

__global__ void __launch_bounds__(256) k_sh_bwd(const float* __restrict__ go, const float* __restrict__ d,
                                                int64_t M, int degree, float* __restrict__ gd) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float x = d[m * 3], y = d[m * 3 + 1], z = d[m * 3 + 2];
  const int C = degree * degree;
  const float* g = go + m * C;
  float gx = 0, gy = 0, gz = 0;
  if (degree > 1) {
    gy += -0.48860251190291987f * g[1];
    gz += 0.48860251190291987f * g[2];
    gx += -0.48860251190291987f * g[3];
  }
  if (degree > 2) {
    gx += 1.0925484305920792f * y * g[4];  gy += 1.0925484305920792f * x * g[4];
    gy += -1.0925484305920792f * z * g[5]; gz += -1.0925484305920792f * y * g[5];
    gz += 2.0f * 0.94617469575755997f * z * g[6];
    gx += -1.0925484305920792f * z * g[7]; gz += -1.0925484305920792f * x * g[7];
    gx += 2.0f * 0.54627421529603959f * x * g[8]; gy += -2.0f * 0.54627421529603959f * y * g[8];
  }
  if (degree > 3) {
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    // 9: a*y*(-3x^2+y^2)
    gx += 0.59004358992664352f * (-6.0f * x * y) * g[9];
    gy += 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2) * g[9];
    // 10: b*x*y*z
    gx += 2.8906114426405538f * y * z * g[10]; gy += 2.8906114426405538f * x * z * g[10];
    gz += 2.8906114426405538f * x * y * g[10];
    // 11: c*y*(1-5z^2)
    gy += 0.45704579946446572f * (1.0f - 5.0f * z2) * g[11]; gz += 0.45704579946446572f * (-10.0f * y * z) * g[11];
    // 12: e*z*(5z^2-3)
    gz += 0.3731763325901154f * (15.0f * z2 - 3.0f) * g[12];
    // 13: c*x*(1-5z^2)
    gx += 0.45704579946446572f * (1.0f - 5.0f * z2) * g[13]; gz += 0.45704579946446572f * (-10.0f * x * z) * g[13];
    // 14: f*z*(x^2-y^2)
    gx += 1.4453057213202769f * 2.0f * x * z * g[14]; gy += -1.4453057213202769f * 2.0f * y * z * g[14];
    gz += 1.4453057213202769f * (x2 - y2) * g[14];
    // 15: a*x*(-x^2+3y^2)
    gx += 0.59004358992664352f * (-3.0f * x2 + 3.0f * y2) * g[15];
    gy += 0.59004358992664352f * (6.0f * x * y) * g[15];
  }
  gd[m * 3] = gx; gd[m * 3 + 1] = gy; gd[m * 3 + 2] = gz;
}

// ---- Adam: one read of p,g,m,v and one write of p,m,v per element, 16 B per lane -------------
__global__ void __launch_bounds__(256) k_adam(float4* __restrict__ p, const float4* __restrict__ g,
                                              float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                              float lr_t, float b1, float b2, float eps_t, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
#define INR_ADAM1(c)                                   \
  {                                                    \
    const float gr = gg.c * gscale;                    \
    mm.c = b1 * mm.c + (1.0f - b1) * gr;               \
    vv.c = b2 * vv.c + (1.0f - b2) * gr * gr;          \
    pp.c = pp.c - lr_t * mm.c / (sqrtf(vv.c) + eps_t); \
  }
    INR_ADAM1(x) INR_ADAM1(y) INR_ADAM1(z) INR_ADAM1(w)
#undef INR_ADAM1
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}
__global__ void k_adam_tail(float* p, const float* g, float* m, float* v, int64_t start, int64_t n, float lr_t,
                            float b1, float b2, float eps_t, float gscale) {
  const int64_t i = start + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i] * gscale;
  const float mm = b1 * m[i] + (1.0f - b1) * gr;
  const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
  m[i] = mm; v[i] = vv;
  p[i] = p[i] - lr_t * mm / (sqrtf(vv) + eps_t);
}

// all tensors of an optimiser step in ONE launch (blockIdx.y = tensor): a training step has 4-9 parameter
// tensors, most of them a few KB, and a launch per tensor costs more than their arithmetic
constexpr int kAdamMaxTensors = 16;
struct AdamJobs {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  int64_t n[kAdamMaxTensors];
  float lr_t[kAdamMaxTensors];
  float* s[kAdamMaxTensors];      // parameter EMA (nullable per tensor): s += ema_w * (p_new - s) while p is in registers
};
struct AdamHyper {
  float v[2 + kAdamMaxTensors];     // [eps_t, lr_t[0..15], ema weight]
};
__global__ void k_adam_set_hyper(AdamHyper h, float* __restrict__ hyper, int n) {
  if ((int)threadIdx.x < n) hyper[threadIdx.x] = h.v[threadIdx.x];
}

// Up to 8 small device-to-device copies in ONE launch (the static inputs of a captured training step: rays, labels and
// the sample counter are handed to the graph through fixed buffers - five copy launches per step serialised in front
// of every replay, ~25 us of a 0.85 ms step).  Byte counts and addresses must be multiples of 4.
constexpr int kCopyMaxJobs = 8;
struct CopyJobs {
  uint32_t* dst[kCopyMaxJobs];
  const uint32_t* src[kCopyMaxJobs];
  int64_t words[kCopyMaxJobs];
};
__global__ void __launch_bounds__(256) k_copy_multi(CopyJobs J) {
  const int t = blockIdx.y;
  const int64_t n = J.words[t], stride = (int64_t)gridDim.x * blockDim.x;
  uint32_t* __restrict__ d = J.dst[t];
  const uint32_t* __restrict__ s = J.src[t];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}

// hyper (nullable, device): [eps_t, lr_t[0..15]] - the step-dependent scalars live in memory so that a captured
// hipGraph of the training step can be replayed with a new learning rate / bias correction
__global__ void __launch_bounds__(256) k_adam_multi(AdamJobs J, float b1, float b2, float eps_t, float gscale,
                                                    const float* __restrict__ hyper, float ema_w) {
  const int t = blockIdx.y;
  if (hyper) {
    eps_t = hyper[0];
    J.lr_t[t] = hyper[1 + t];
    if (J.s[t]) ema_w = hyper[1 + kAdamMaxTensors];
  }
  float* __restrict__ p = J.p[t];
  const float* __restrict__ g = J.g[t];
  float* __restrict__ m = J.m[t];
  float* __restrict__ v = J.v[t];
  float* __restrict__ sh = J.s[t];
  const int64_t n = J.n[t];
  const float lr_t = J.lr_t[t];
  const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)sh) & 15) == 0;
  const int64_t n4 = aligned ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = tid; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i],
           mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define INR_ADAM1(c)                                   \
  {                                                    \
    const float gr = gg.c * gscale;                    \
    mm.c = b1 * mm.c + (1.0f - b1) * gr;               \
    vv.c = b2 * vv.c + (1.0f - b2) * gr * gr;          \
    pp.c = pp.c - lr_t * mm.c / (sqrtf(vv.c) + eps_t); \
  }
    INR_ADAM1(x) INR_ADAM1(y) INR_ADAM1(z) INR_ADAM1(w)
#undef INR_ADAM1
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    if (sh) {
      float4 ss = reinterpret_cast<float4*>(sh)[i];
      ss.x += ema_w * (pp.x - ss.x); ss.y += ema_w * (pp.y - ss.y);
      ss.z += ema_w * (pp.z - ss.z); ss.w += ema_w * (pp.w - ss.w);
      reinterpret_cast<float4*>(sh)[i] = ss;
    }
  }
  for (int64_t i = n4 * 4 + tid; i < n; i += stride) {
    const float gr = g[i] * gscale;
    const float mm = b1 * m[i] + (1.0f - b1) * gr;
    const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mm; v[i] = vv;
    const float pn = p[i] - lr_t * mm / (sqrtf(vv) + eps_t);
    p[i] = pn;
    if (sh) sh[i] += ema_w * (pn - sh[i]);
  }
}

// ---- weight gradient of a bias-free linear layer: dW[o][i] += sum_m gy[m][o] * x[m][i] -------------------
// The reduction runs over SAMPLES (m ~ 2e5) while o, i <= 64: a GEMM with a tiny output and a huge K,
// which the BLAS library runs on 2-4 workgroups (measured 0.43-0.48 ms per layer).  Here every wave
// takes 16-sample units, keeps the whole [<=64 x <=64] result in 16 MFMA accumulators
// (v_mfma_f32_16x16x4_f32: exact fp32, k = 4 samples per instruction); the waves of a workgroup sum their
// results in LDS and the workgroup adds one set to dW with address-adjacent atomics (split-K over the CUs).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Feature f of the 16*N-wide operand sits in MFMA block f % N, row/column f / N: lane j then owns the N
// CONTIGUOUS features N*j .. N*j+N-1 of a sample row - one 16-byte load when N = 4 - and any such relabelling
// is legal as long as the write-out uses it too.
template <int N>
__device__ __forceinline__ void wgrad_load(const float* __restrict__ row, int n, int j, bool valid, float* v) {
  if (n == 16 * N) {                   // wave-uniform: full width, vector load
    if constexpr (N == 4) {
      const float4 t = valid ? *reinterpret_cast<const float4*>(row + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      return;
    } else if constexpr (N == 2) {
      const float2 t = valid ? *reinterpret_cast<const float2*>(row + 2 * j) : make_float2(0.f, 0.f);
      v[0] = t.x; v[1] = t.y;
      return;
    }
  }
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = (valid && N * j + e < n) ? row[N * j + e] : 0.f;
}

// Pass 1: units of 16 samples are dealt round-robin to all waves of the grid (8 waves per workgroup, two
// workgroups per CU).  The 8 partial results of a workgroup are summed by a 3-round tree through LDS (plain
// 16-byte stores and loads - ds_add_f32 and same-address global atomics were both measured far slower) and the
// workgroup's [N_OT x N_IT x 256] fragment-ordered partial goes to `partial[blockIdx.x]` with plain stores.
constexpr int kWgradWaves = 8;       // tree buffer = 4 waves x 16 KB = 64 KB; two workgroups per CU
constexpr int kWgradGroupsPerCU = 2;
template <int N_OT, int N_IT>
__global__ void __launch_bounds__(kWgradWaves * 64) k_linear_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                                  int64_t M, int n_in, int n_out,
                                                                  float4* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float4 tree[];     // [kWgradWaves/2][N_OT*N_IT][64 lanes]
  constexpr int kBlk = N_OT * N_IT;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15, w = threadIdx.x >> 6;
  const int64_t n_waves = (int64_t)gridDim.x * kWgradWaves;
  const int64_t wave = (int64_t)w * gridDim.x + blockIdx.x;          // neighbouring units -> different CUs
  const int64_t n_units = (M + 15) >> 4;
  f32x4 acc[N_OT][N_IT];
#pragma unroll
  for (int a = 0; a < N_OT; ++a)
#pragma unroll
    for (int b = 0; b < N_IT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 16 samples per iteration: the 8 row loads (4 k-steps x 2 operands) are issued before the MFMAs
  for (int64_t unit = wave; unit < n_units; unit += n_waves) {
    const int64_t mg = unit * 16;
    float av[4][N_OT], bv[4][N_IT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t row = mg + 4 * u + q;            // k-slot q of k-step u
      const bool rv = row < M;
      wgrad_load<N_OT>(gy + row * n_out, n_out, j, rv, av[u]);      // A[i = out][k = sample]
      wgrad_load<N_IT>(x + row * n_in, n_in, j, rv, bv[u]);         // B[k = sample][j = in]
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int ot = 0; ot < N_OT; ++ot)
#pragma unroll
        for (int it = 0; it < N_IT; ++it)
          acc[ot][it] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][ot], bv[u][it], acc[ot][it], 0, 0, 0);
  }
#pragma unroll
  for (int half = kWgradWaves / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) {
#pragma unroll
      for (int b = 0; b < kBlk; ++b) {
        const f32x4 v = acc[b / N_IT][b % N_IT];
        tree[((w - half) * kBlk + b) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    __syncthreads();
    if (w < half) {
#pragma unroll
      for (int b = 0; b < kBlk; ++b) {
        const float4 v = tree[(w * kBlk + b) * 64 + lane];
        acc[b / N_IT][b % N_IT] += f32x4{v.x, v.y, v.z, v.w};
      }
    }
    __syncthreads();
  }
  if (w == 0) {
#pragma unroll
    for (int b = 0; b < kBlk; ++b) {
      const f32x4 v = acc[b / N_IT][b % N_IT];
      partial[((size_t)blockIdx.x * kBlk + b) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// Pass 2: gw[o][i] += sum over workgroups; fragment (ot, it) lane (q, j) register r holds output feature
// N_OT*(4q+r)+ot, input feature N_IT*j+it (see wgrad_load).
__global__ void __launch_bounds__(1024) k_wgrad_reduce(const float* __restrict__ partial, int n_groups, int n_ot, int n_it,
                                                       int n_in, int n_out, float* __restrict__ gw) {
  // 64 consecutive elements per block, the groups split over 16 waves with 8 loads in flight each
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  const int per = n_ot * n_it * 256;
  float s = 0.f;
  int g = slice;
  for (; g + 7 * 16 < n_groups; g += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(g + 16 * u) * per + idx];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; g < n_groups; g += 16) s += partial[(size_t)g * per + idx];
  red[slice][e] = s;
  __syncthreads();
  if (slice == 0) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][e];
    const int b = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
    const int o = n_ot * (4 * (lane >> 4) + r) + b / n_it, i = n_it * (lane & 15) + b % n_it;
    if (o < n_out && i < n_in) gw[(size_t)o * n_in + i] += s;
  }
}

}  // namespace inr

using namespace inr;

static int grid_bwd_launch(const float* x, const float* grad_out, const int32_t* order, const inr_grid_desc* desc, int64_t M,
                           float bound, float* grad_embeddings, int32_t level_lo, int32_t level_hi, float* fx_state,
                           long long* acc64, inr_stream_t s);

template <int N_OT>
static void wgrad_launch(unsigned nb, hipStream_t st, const float* x, const float* gy, int64_t M, int n_in, int n_out,
                         float4* partial) {
  const int n_it = (n_in + 15) / 16;
  const size_t lds = (size_t)(kWgradWaves / 2) * N_OT * n_it * 64 * sizeof(float4);
  switch (n_it) {
    case 1: k_linear_wgrad<N_OT, 1><<<nb, kWgradWaves * 64, lds, st>>>(x, gy, M, n_in, n_out, partial); break;
    case 2: k_linear_wgrad<N_OT, 2><<<nb, kWgradWaves * 64, lds, st>>>(x, gy, M, n_in, n_out, partial); break;
    case 3: k_linear_wgrad<N_OT, 3><<<nb, kWgradWaves * 64, lds, st>>>(x, gy, M, n_in, n_out, partial); break;
    default: k_linear_wgrad<N_OT, 4><<<nb, kWgradWaves * 64, lds, st>>>(x, gy, M, n_in, n_out, partial); break;
  }
}

extern "C" {

int inr_grid_encode_forward(const float* x, const float* embeddings, const inr_grid_desc* desc, int64_t M,
                            float bound, float* out, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && out, "null pointer");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)out & 7) == 0, "embeddings/out must be 8-byte aligned");
  if (M == 0) return INR_OK;
  const int64_t total = M * G.num_levels;
  k_grid_fwd<<<blocks_for(total, 256), 256, 0, as_stream(s)>>>(x, reinterpret_cast<const float2*>(embeddings), G, M,
                                                               bound, reinterpret_cast<float2*>(out));
  return check_launch("grid_encode_forward");
}

int inr_grid_encode_backward(const float* x, const float* grad_out, const inr_grid_desc* desc, int64_t M, float bound,
                             float* grad_embeddings, inr_stream_t s) {
  return inr_grid_encode_backward_ordered(x, grad_out, nullptr, desc, M, bound, grad_embeddings, s);
}

int inr_grid_encode_backward_ordered(const float* x, const float* grad_out, const int32_t* order,
                                     const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                     inr_stream_t s) {
  return inr_grid_encode_backward_levels(x, grad_out, order, desc, M, bound, grad_embeddings, 0, desc ? desc->num_levels : 0, s);
}

int inr_grid_encode_backward_levels(const float* x, const float* grad_out, const int32_t* order,
                                    const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                    int32_t level_lo, int32_t level_hi, inr_stream_t s) {
  return inr_grid_encode_backward_levels_fx(x, grad_out, order, desc, M, bound, grad_embeddings, level_lo, level_hi, nullptr, s);
}

int inr_grid_grad_finish_fx(float* grad_embeddings, const inr_grid_desc* desc, int32_t level_lo, int32_t level_hi,
                            float* fx_state, inr_stream_t s) {
  INR_REQUIRE(desc && grad_embeddings && fx_state, "null pointer");
  INR_REQUIRE(level_lo >= 0 && level_lo < level_hi && level_hi <= desc->num_levels, "bad level range");
  INR_REQUIRE(((uintptr_t)grad_embeddings & 15) == 0, "grad_embeddings must be 16-byte aligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  for (int l = level_lo; l <= level_hi; ++l)
    INR_REQUIRE((desc->offsets[l] & 7u) == 0, "level offsets must be multiples of 8 rows");
  const dim3 grid(kFxBlocks, (unsigned)(level_hi - level_lo));
  k_grid_grad_finish<<<grid, 256, 0, as_stream(s)>>>(grad_embeddings, G, level_lo, fx_state);
  return check_launch("grid_grad_finish_fx");
}

int inr_grid_fx_update(float* fx_state, int32_t num_levels, float headroom, int32_t sum_bits, inr_stream_t s) {
  INR_REQUIRE(fx_state, "null pointer");
  INR_REQUIRE(num_levels >= 1 && num_levels <= INR_MAX_LEVELS, "num_levels out of range");
  INR_REQUIRE(headroom >= 2.0f && headroom <= 1048576.0f, "headroom must be in [2, 2^20]");
  INR_REQUIRE(sum_bits == 32 || sum_bits == 64, "sum_bits must be 32 or 64");
  k_grid_fx_update<<<1, 1024, 0, as_stream(s)>>>(fx_state, num_levels, headroom, sum_bits == 64 ? 9223372036854775808.0f : 2147483648.0f);
  return check_launch("grid_fx_update");
}

int inr_grid_grad_finish_fx64(int64_t* acc64, float* grad_embeddings, const inr_grid_desc* desc, int32_t level_lo, int32_t level_hi,
                              float* fx_state, inr_stream_t s) {
  INR_REQUIRE(desc && acc64 && grad_embeddings && fx_state, "null pointer");
  INR_REQUIRE(level_lo >= 0 && level_lo < level_hi && level_hi <= desc->num_levels, "bad level range");
  INR_REQUIRE(((uintptr_t)acc64 & 15) == 0 && ((uintptr_t)grad_embeddings & 7) == 0, "acc64 must be 16-byte, grad_embeddings 8-byte aligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const dim3 grid(kFxBlocks, (unsigned)(level_hi - level_lo));
  k_grid_grad_finish64<<<grid, 256, 0, as_stream(s)>>>(reinterpret_cast<long long*>(acc64), grad_embeddings, G, level_lo, fx_state);
  return check_launch("grid_grad_finish_fx64");
}

int inr_grid_encode_backward_levels_fx64(const float* x, const float* grad_out, const int32_t* order, const inr_grid_desc* desc,
                                         int64_t M, float bound, float* grad_embeddings, int64_t* acc64, int32_t level_lo,
                                         int32_t level_hi, float* fx_state, inr_stream_t s) {
  INR_REQUIRE(acc64 && fx_state, "null pointer (acc64 / fx_state)");
  INR_REQUIRE(((uintptr_t)acc64 & 15) == 0, "acc64 must be 16-byte aligned");
  return grid_bwd_launch(x, grad_out, order, desc, M, bound, grad_embeddings, level_lo, level_hi, fx_state,
                         reinterpret_cast<long long*>(acc64), s);
}

int inr_grid_encode_backward_levels_fx(const float* x, const float* grad_out, const int32_t* order,
                                       const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                       int32_t level_lo, int32_t level_hi, float* fx_state, inr_stream_t s) {
  return grid_bwd_launch(x, grad_out, order, desc, M, bound, grad_embeddings, level_lo, level_hi, fx_state, nullptr, s);
}

}  // extern "C"

static int grid_bwd_launch(const float* x, const float* grad_out, const int32_t* order, const inr_grid_desc* desc, int64_t M,
                           float bound, float* grad_embeddings, int32_t level_lo, int32_t level_hi, float* fx_state,
                           long long* acc64, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  INR_REQUIRE(level_lo >= 0 && level_lo < level_hi && level_hi <= desc->num_levels, "bad level range");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && grad_out && grad_embeddings, "null pointer");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  INR_REQUIRE(((uintptr_t)grad_out & 7) == 0, "grad_out must be 8-byte aligned");
  if (M == 0) return INR_OK;
  INR_REQUIRE((uint64_t)desc->offsets[desc->num_levels] * 2ull < (1ull << 32), "table too large for 32-bit element offsets");
  // grid.y is limited to 65535 blocks of 64 samples: longer sample arrays go in several launches
  const int64_t chunk = (int64_t)65535 * 64;
  for (int64_t m0 = 0; m0 < M; m0 += chunk) {
    const int64_t mc = std::min(chunk, M - m0);
    const dim3 grid((unsigned)(level_hi - level_lo), blocks_for(mc * 4, 256));
    k_grid_bwd<<<grid, 256, 0, as_stream(s)>>>(order ? x : x + m0 * 3, order ? grad_out : grad_out + m0 * G.num_levels * 2,
                                               order ? order + m0 : nullptr, G, mc, bound, grad_embeddings, level_lo, fx_state, acc64);
  }
  return check_launch("grid_encode_backward");
}

extern "C" {

int inr_grid_encode_backward_input(const float* x, const float* grad_out, const float* embeddings,
                                   const inr_grid_desc* desc, int64_t M, float bound, float* grad_x, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && grad_out && embeddings && grad_x, "null pointer");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)grad_out & 7) == 0, "embeddings/grad_out must be 8-byte aligned");
  k_grid_bwd_input<<<blocks_for(M * 16, 256), 256, 0, as_stream(s)>>>(x, reinterpret_cast<const float2*>(grad_out),
                                                                      reinterpret_cast<const float2*>(embeddings), G, M,
                                                                      bound, grad_x);
  return check_launch("grid_encode_backward_input");
}

int inr_sh_encode_forward(const float* d, int64_t M, int32_t degree, float* out, inr_stream_t s) {
  INR_REQUIRE(M >= 0, "negative M");
  INR_REQUIRE(degree >= 1 && degree <= 4, "degree must be 1..4");
  if (M == 0) return INR_OK;
  INR_REQUIRE(d && out, "null pointer");
  k_sh_fwd<<<blocks_for(M, 256), 256, 0, as_stream(s)>>>(d, M, degree, out);
  return check_launch("sh_encode_forward");
}

int inr_sh_table_q(const float* d, int64_t N, float* out, inr_stream_t s) {
  INR_REQUIRE(N >= 0, "negative N");
  if (N == 0) return INR_OK;
  INR_REQUIRE(d && out && ((uintptr_t)out & 15) == 0, "null or misaligned pointer");
  k_sh_table_q<<<blocks_for(N, 256), 256, 0, as_stream(s)>>>(d, N, out);
  return check_launch("sh_table_q");
}

int inr_sh_encode_backward(const float* grad_out, const float* d, int64_t M, int32_t degree, float* grad_d,
                           inr_stream_t s) {
  INR_REQUIRE(M >= 0, "negative M");
  INR_REQUIRE(degree >= 1 && degree <= 4, "degree must be 1..4");
  if (M == 0) return INR_OK;
  INR_REQUIRE(grad_out && d && grad_d, "null pointer");
  k_sh_bwd<<<blocks_for(M, 256), 256, 0, as_stream(s)>>>(grad_out, d, M, degree, grad_d);
  return check_launch("sh_encode_backward");
}

int64_t inr_linear_wgrad_workspace_bytes(void) {
  return (int64_t)cu_count() * kWgradGroupsPerCU * 64 * 64 * (int64_t)sizeof(float);
}

int inr_linear_wgrad(const float* x, const float* grad_y, int64_t M, int32_t n_in, int32_t n_out, float* grad_w,
                     void* workspace, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && n_in > 0 && n_in <= 64 && n_out > 0 && n_out <= 64, "n_in and n_out must be in 1..64");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && grad_y && grad_w && workspace, "null pointer");
  // full-width rows are read with 16/8-byte loads
  INR_REQUIRE((n_in % 16 != 0 || ((uintptr_t)x & 15) == 0) && (n_out % 16 != 0 || ((uintptr_t)grad_y & 15) == 0) &&
                  ((uintptr_t)workspace & 15) == 0, "x / grad_y / workspace must be 16-byte aligned");
  const int64_t units = (M + 15) / 16;
  const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>((units + kWgradWaves - 1) / kWgradWaves, (int64_t)cu_count() * kWgradGroupsPerCU));
  hipStream_t st = as_stream(s);
  float4* partial = reinterpret_cast<float4*>(workspace);
  const int n_ot = (n_out + 15) / 16, n_it = (n_in + 15) / 16;
  switch (n_ot) {
    case 1: wgrad_launch<1>(nb, st, x, grad_y, M, n_in, n_out, partial); break;
    case 2: wgrad_launch<2>(nb, st, x, grad_y, M, n_in, n_out, partial); break;
    case 3: wgrad_launch<3>(nb, st, x, grad_y, M, n_in, n_out, partial); break;
    default: wgrad_launch<4>(nb, st, x, grad_y, M, n_in, n_out, partial); break;
  }
  k_wgrad_reduce<<<n_ot * n_it * 4, 1024, 0, st>>>(reinterpret_cast<const float*>(workspace), (int)nb, n_ot, n_it, n_in, n_out,
                                              grad_w);
  return check_launch("linear_wgrad");
}

static int adam_multi_launch(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                             float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                             float eps, int32_t step, float grad_scale, float* const* ema, float ema_weight,
                             inr_stream_t s, const char* what) {
  INR_REQUIRE(n_tensors >= 0 && n_tensors <= kAdamMaxTensors && step >= 1, "bad argument (at most 16 tensors per call)");
  if (n_tensors == 0) return INR_OK;
  INR_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && numels && lrs, "null pointer");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamJobs J;
  int64_t n_max = 0;
  for (int t = 0; t < n_tensors; ++t) {
    INR_REQUIRE(numels[t] >= 0 && (numels[t] == 0 || (params[t] && grads[t] && exp_avgs[t] && exp_avg_sqs[t])),
                "null tensor pointer");
    J.p[t] = params[t]; J.g[t] = grads[t]; J.m[t] = exp_avgs[t]; J.v[t] = exp_avg_sqs[t]; J.n[t] = numels[t];
    J.s[t] = ema ? ema[t] : nullptr;
    J.lr_t[t] = (float)(lrs[t] * sqrt(bc2) / bc1);      // same folding as inr_adam_step
    n_max = std::max(n_max, numels[t]);
  }
  if (n_max == 0) return INR_OK;
  const unsigned nb = (unsigned)std::min<int64_t>((n_max / 4 + 255) / 256 + 1, 256 * 16);
  k_adam_multi<<<dim3(nb, n_tensors), 256, 0, as_stream(s)>>>(J, beta1, beta2, (float)(eps * sqrt(bc2)), grad_scale,
                                                              nullptr, ema_weight);
  return check_launch(what);
}

int inr_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                        float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                        float eps, int32_t step, float grad_scale, inr_stream_t s) {
  return adam_multi_launch(n_tensors, params, grads, exp_avgs, exp_avg_sqs, numels, lrs, beta1, beta2, eps, step,
                           grad_scale, nullptr, 0.f, s, "adam_step_multi");
}

int inr_adam_ema_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                            float eps, int32_t step, float grad_scale, float* const* ema_shadows, float ema_weight,
                            inr_stream_t s) {
  INR_REQUIRE(n_tensors == 0 || ema_shadows, "null pointer");
  INR_REQUIRE(ema_weight >= 0.f && ema_weight <= 1.f, "ema_weight outside [0, 1]");
  return adam_multi_launch(n_tensors, params, grads, exp_avgs, exp_avg_sqs, numels, lrs, beta1, beta2, eps, step,
                           grad_scale, ema_shadows, ema_weight, s, "adam_ema_step_multi");
}

int inr_copy_multi(int32_t n, void* const* dsts, const void* const* srcs, const int64_t* nbytes, inr_stream_t s) {
  INR_REQUIRE(n >= 0 && n <= kCopyMaxJobs, "bad argument (at most 8 copies per call)");
  if (n == 0) return INR_OK;
  INR_REQUIRE(dsts && srcs && nbytes, "null pointer");
  CopyJobs J;
  int64_t w_max = 0;
  for (int t = 0; t < n; ++t) {
    INR_REQUIRE(nbytes[t] >= 0 && nbytes[t] % 4 == 0 && (nbytes[t] == 0 || (dsts[t] && srcs[t])), "bad copy job");
    INR_REQUIRE((((uintptr_t)dsts[t] | (uintptr_t)srcs[t]) & 3) == 0, "copy job not 4-byte aligned");
    J.dst[t] = (uint32_t*)dsts[t]; J.src[t] = (const uint32_t*)srcs[t]; J.words[t] = nbytes[t] / 4;
    w_max = std::max(w_max, J.words[t]);
  }
  if (w_max == 0) return INR_OK;
  const unsigned nb = (unsigned)std::min<int64_t>((w_max + 255) / 256, 1024);
  k_copy_multi<<<dim3(nb, n), 256, 0, as_stream(s)>>>(J);
  return check_launch("copy_multi");
}

int inr_adam_set_hyper(const float* lrs, int32_t n_tensors, float beta1, float beta2, float eps, int32_t step,
                       float* hyper_dev, inr_stream_t s) {
  INR_REQUIRE(lrs && hyper_dev && n_tensors >= 0 && n_tensors <= kAdamMaxTensors && step >= 1, "bad argument");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamHyper h;
  h.v[0] = (float)(eps * sqrt(bc2));
  for (int t = 0; t < kAdamMaxTensors; ++t) h.v[1 + t] = t < n_tensors ? (float)(lrs[t] * sqrt(bc2) / bc1) : 0.f;
  // the values travel as kernel arguments (copied at launch), so the caller's buffers may change right away
  h.v[1 + kAdamMaxTensors] = 0.f;
  k_adam_set_hyper<<<1, 32, 0, as_stream(s)>>>(h, hyper_dev, 1 + kAdamMaxTensors);
  return check_launch("adam_set_hyper");
}

int inr_adam_set_hyper_ema(const float* lrs, int32_t n_tensors, float beta1, float beta2, float eps, int32_t step,
                           float ema_weight, float* hyper_dev, inr_stream_t s) {
  INR_REQUIRE(lrs && hyper_dev && n_tensors >= 0 && n_tensors <= kAdamMaxTensors && step >= 1, "bad argument");
  INR_REQUIRE(ema_weight >= 0.f && ema_weight <= 1.f, "ema_weight outside [0, 1]");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamHyper h;
  h.v[0] = (float)(eps * sqrt(bc2));
  for (int t = 0; t < kAdamMaxTensors; ++t) h.v[1 + t] = t < n_tensors ? (float)(lrs[t] * sqrt(bc2) / bc1) : 0.f;
  h.v[1 + kAdamMaxTensors] = ema_weight;
  k_adam_set_hyper<<<1, 32, 0, as_stream(s)>>>(h, hyper_dev, 2 + kAdamMaxTensors);
  return check_launch("adam_set_hyper_ema");
}

static int adam_multi_dev_launch(int32_t n_tensors, float* const* params, const float* const* grads,
                                 float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* numels,
                                 const float* hyper_dev, float beta1, float beta2, float grad_scale,
                                 float* const* ema_shadows, inr_stream_t s);

int inr_adam_step_multi_dev(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* numels, const float* hyper_dev, float beta1,
                            float beta2, float grad_scale, inr_stream_t s) {
  return adam_multi_dev_launch(n_tensors, params, grads, exp_avgs, exp_avg_sqs, numels, hyper_dev, beta1, beta2,
                               grad_scale, nullptr, s);
}

int inr_adam_ema_step_multi_dev(int32_t n_tensors, float* const* params, const float* const* grads,
                                float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* numels,
                                const float* hyper_dev, float beta1, float beta2, float grad_scale,
                                float* const* ema_shadows, inr_stream_t s) {
  INR_REQUIRE(n_tensors == 0 || ema_shadows, "null pointer");
  return adam_multi_dev_launch(n_tensors, params, grads, exp_avgs, exp_avg_sqs, numels, hyper_dev, beta1, beta2,
                               grad_scale, ema_shadows, s);
}

static int adam_multi_dev_launch(int32_t n_tensors, float* const* params, const float* const* grads,
                                 float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* numels,
                                 const float* hyper_dev, float beta1, float beta2, float grad_scale,
                                 float* const* ema_shadows, inr_stream_t s) {
  INR_REQUIRE(n_tensors >= 0 && n_tensors <= kAdamMaxTensors, "bad argument (at most 16 tensors per call)");
  if (n_tensors == 0) return INR_OK;
  INR_REQUIRE(params && grads && exp_avgs && exp_avg_sqs && numels && hyper_dev, "null pointer");
  AdamJobs J;
  int64_t n_max = 0;
  for (int t = 0; t < n_tensors; ++t) {
    INR_REQUIRE(numels[t] >= 0 && (numels[t] == 0 || (params[t] && grads[t] && exp_avgs[t] && exp_avg_sqs[t])),
                "null tensor pointer");
    J.p[t] = params[t]; J.g[t] = grads[t]; J.m[t] = exp_avgs[t]; J.v[t] = exp_avg_sqs[t]; J.n[t] = numels[t];
    J.lr_t[t] = 0.f;
    J.s[t] = ema_shadows ? ema_shadows[t] : nullptr;
    n_max = std::max(n_max, numels[t]);
  }
  if (n_max == 0) return INR_OK;
  const unsigned nb = (unsigned)std::min<int64_t>((n_max / 4 + 255) / 256 + 1, 256 * 16);
  k_adam_multi<<<dim3(nb, n_tensors), 256, 0, as_stream(s)>>>(J, beta1, beta2, 0.f, grad_scale, hyper_dev, 0.f);
  return check_launch("adam_step_multi_dev");
}

int inr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, int32_t step, float grad_scale, inr_stream_t s) {
  INR_REQUIRE(n >= 0 && step >= 1, "bad argument");
  if (n == 0) return INR_OK;
  INR_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null pointer");
  // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  //                     = (lr*sqrt(bc2)/bc1) * m / (sqrt(v) + eps*sqrt(bc2))
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float lr_t = (float)(lr * sqrt(bc2) / bc1);
  const float eps_t = (float)(eps * sqrt(bc2));
  const bool aligned = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  const int64_t n4 = aligned ? n / 4 : 0;
  hipStream_t st = as_stream(s);
  if (n4) {
    const unsigned nb = (unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
    k_adam<<<nb, 256, 0, st>>>(reinterpret_cast<float4*>(param), reinterpret_cast<const float4*>(grad),
                               reinterpret_cast<float4*>(exp_avg), reinterpret_cast<float4*>(exp_avg_sq), n4, lr_t,
                               beta1, beta2, eps_t, grad_scale);
  }
  for (int64_t start = n4 * 4; start < n; start += 1024)
    k_adam_tail<<<1, 1024, 0, st>>>(param, grad, exp_avg, exp_avg_sq, start, std::min<int64_t>(n, start + 1024), lr_t,
                                    beta1, beta2, eps_t, grad_scale);
  return check_launch("adam_step");
}

}  // extern "C"
