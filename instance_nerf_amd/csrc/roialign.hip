// 3-D RoIAlign for gfx950: the op behind the reference's one in-tree FFI call,
// roi_align.roi_align.roi_align_3d (/root/reference/nerf_rcnn/model/utils.py:604-609; callers
// model/poolers.py:144,174 and model/nerf_rcnn.py:831).  Semantics = torchvision roi_align
// (aligned=False, adaptive ceil(roi/out) sampling grid, average pooling) on three axes, as the
// reference's wrapper documents (utils.py:556-592); the extension itself is an un-vendored submodule.
//
// Two implementations, same sample geometry (float32, operation by operation as oracle/roialign.py):
//  * SEPARABLE (round 4, the default): the op is linear and its sample weights factor per axis - a sample's
//    trilinear weight is wx*wy*wz and its inside test is a conjunction of per-axis tests - so
//        out[pw,pl,ph] = 1/count * sum_{x,y,z} Tx[pw][x] * Ty[pl][y] * Tz[ph][z] * in[x,y,z]
//    with T_a[p][cell] = the summed lo/hi weights of output index p's grid samples on axis a.  A workgroup owns
//    one RoI and a run of channels: it builds the three small tables ONCE in LDS (the torchvision kernel shape
//    recomputes the geometry and 8 weights per sample PER CHANNEL), then per group of 4 channels contracts
//    z (global loads, h is the contiguous axis of [N,C,W,L,H]) -> y (LDS) -> x (LDS -> registers) over x slabs
//    and writes [K,C,ow*ol*oh] coalesced.  8 g^3 loads per output become ~(g+2) per axis pass: 24 000 -> ~4 000
//    global loads per (RoI, channel) on BASELINE configs[4].  Workgroups are dealt so that the ones resident on
//    an XCD at a time share ONE run of channels (2 MB of the volume: stays in that XCD's 4 MB L2).
//  * one lane per output element (rounds 1-3, the torchvision kernel shape): kept for shapes whose tables do not
//    fit the LDS budget and as the A/B the tests compare with (inr_roi_align_3d_set_mode).
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

// ---------------------------------------------------------------------------------------------------------------
// Separable implementation.
// LDS image of one RoI (floats / ints, in this order):
//   T[0] [ow][W], T[1] [ol][L], T[2] [oh][H]   dense per-axis weight rows over ABSOLUTE cell indices
//   first/last cell of every row (ints; first > last: no sample of that output index lies inside the volume)
//   (backward only) first/last output index of every cell
//   tmp1, tmp2: the slab intermediates, SEP_CH channels interleaved (f32x4)
constexpr int SEP_THREADS = 256;
constexpr int SEP_CH = 4;            // channels a thread carries through every pass (weights and indices are shared)
constexpr int SEP_TMP_FLOATS = 10240; // tmp1 + tmp2 budget per workgroup (40 KB): ~45 KB in all -> 3 workgroups per CU,
                                      // what the forward kernel's registers allow anyway; fewer x slabs beat a 4th
                                      // workgroup (tools/micro/roialign_bench.hip: 0.152 vs 0.160 ms at 8192)

struct SepArgs {
  int C, W, L, H, ow, ol, oh;
  float scale;
  int cpb;       // channels per workgroup (a multiple of SEP_CH)
  int ngroups;   // ceil(C / cpb)
  int K;
  int tmp_floats;
};

struct SepRoi {       // in LDS (lSepRoi)
  float start[3], bin[3];
  int grid[3];
  float inv_count;
  int lo[3], hi[3];   // region of cells any output index touches (lo > hi: empty)
};

// LDS pointers carry their address space in the type: the tables are reached through small arrays of pointers, and
// for a pointer the compiler cannot trace back to the __shared__ array it emits FLAT instructions (found in the ISA:
// flat_store_dwordx4 for the slab writes) - the slow path through the vector memory unit.
typedef __attribute__((address_space(3))) float lfloat;
typedef __attribute__((address_space(3))) int lint;
typedef float f32x4 __attribute__((ext_vector_type(4)));          // a native vector: HIP's f32x4 class has no LDS overloads
typedef __attribute__((address_space(3))) f32x4 lfloat4;

struct __attribute__((packed, aligned(4))) F4U { float v[4]; };   // four consecutive floats at dword alignment

typedef __attribute__((address_space(3))) SepRoi lSepRoi;

__device__ __forceinline__ int fdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

// Builds the per-axis tables of RoI k (all threads of the workgroup; two barriers, the second one last).
// n[a] = volume extent, o[a] = output extent.  A row is written over [first, last] only - nothing reads outside.
__device__ __forceinline__ void sep_build_tables(const float* __restrict__ roi, const SepArgs& A, lSepRoi* R,
                                                 lfloat* const T[3], lint* const first[3], lint* const last[3]) {
  const int n[3] = {A.W, A.L, A.H};
  const int o[3] = {A.ow, A.ol, A.oh};
  const int t = threadIdx.x;
  const int nrow = o[0] + o[1] + o[2];
  for (int r = t; r < nrow; r += SEP_THREADS) {
    const int a = r < o[0] ? 0 : (r < o[0] + o[1] ? 1 : 2);
    const int p = r - (a == 0 ? 0 : (a == 1 ? o[0] : o[0] + o[1]));
    const int na = n[a];
    // the same float32 operations, in the same order, as roi_geom() / oracle/roialign.py
    const float start = roi[a] * A.scale, e = roi[a + 3] * A.scale;
    const float size = fmaxf(e - start, 1.0f);
    const float bin = size / (float)o[a];
    const int g = (int)ceilf(size / (float)o[a]);
    lfloat* row = T[a] + p * na;
    int f = na, l = -1;
    for (int i = 0; i < g; ++i) {                       // which cells do this output index's samples touch
      const float v = start + p * bin + ((float)i + 0.5f) * bin / (float)g;
      if (v < -1.0f || v > (float)na) continue;
      const int lo = min((int)fmaxf(v, 0.0f), na - 1);
      f = min(f, lo);
      l = max(l, min(lo + 1, na - 1));
    }
    for (int c = f; c <= l; ++c) row[c] = 0.0f;
    for (int i = 0; i < g; ++i) {
      const float v = start + p * bin + ((float)i + 0.5f) * bin / (float)g;
      if (v < -1.0f || v > (float)na) continue;
      float c = fmaxf(v, 0.0f);
      int lo = (int)c, hi;
      if (lo >= na - 1) { lo = hi = na - 1; c = (float)lo; } else { hi = lo + 1; }
      const float fr = c - (float)lo;
      row[lo] += 1.0f - fr;                             // one lane owns the row: its LDS operations are ordered
      row[hi] += fr;
    }
    first[a][p] = f;
    last[a][p] = l;
    if (p == 0) {
      R->grid[a] = g;
      R->start[a] = start;
      R->bin[a] = bin;
    }
  }
  __syncthreads();
  if (t < 3) {
    int lo = n[t], hi = -1;
    for (int p = 0; p < o[t]; ++p) { lo = min(lo, first[t][p]); hi = max(hi, last[t][p]); }
    R->lo[t] = lo;
    R->hi[t] = hi;
  }
  if (t == 3) {
    const int count = R->grid[0] * R->grid[1] * R->grid[2];
    R->inv_count = 1.0f / (float)max(count, 1);
  }
  __syncthreads();
}

// Which (RoI, channel run) a workgroup owns: workgroups are dispatched round robin over the 8 XCDs, so
// blockIdx % 8 is the XCD; within an XCD the RoI is the fast index and the channel run the slow one - the ~128
// workgroups an XCD holds at a time read the same cpb channels of the volume (cpb * W*L*H*4 bytes: 2 MB for 8
// channels of 40^3) out of its L2 instead of pulling every channel through the fabric once per RoI.
__device__ __forceinline__ bool sep_assign(const SepArgs& A, int* k, int* c0) {
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int gi = i / A.K;
  const int grp = xcd + 8 * gi;
  if (grp >= A.ngroups) return false;
  *k = i - gi * A.K;
  *c0 = grp * A.cpb;
  return true;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef SEP_NT_STORE
#define SEP_NT_STORE 1
#endif
#ifndef SEP_RUN_BARRIER
#define SEP_RUN_BARRIER 0
#endif
#if SEP_NT_STORE
#define SEP_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define SEP_STORE(p, v) (*(p) = (v))
#endif

__device__ __forceinline__ void fma4(f32x4& a, float w, const f32x4 v) {
  a.x = __builtin_fmaf(w, v.x, a.x);
  a.y = __builtin_fmaf(w, v.y, a.y);
  a.z = __builtin_fmaf(w, v.z, a.z);
  a.w = __builtin_fmaf(w, v.w, a.w);
}

// A thread's ROLE in a pass is fixed for the whole RoI - output index ph in the z pass, (pl, ph) in the y pass, its
// NOUT output elements in the x pass - so the taps of that role (first cell + four weights: every row of a RoI whose
// sampling grid is <= 2 per bin, i.e. every box up to 2 x out cells on a side) are loaded into registers once and
// the passes are loads + fused multiply-adds; longer rows take the remaining cells from the LDS table (`more*`).
// The kernel is bound by vector-instruction issue (PMC, profiles/r04_roialign_pmc.txt: a wave64 instruction holds a
// SIMD for four cycles), so instructions per element are what is trimmed: buffer loads (one 32-bit offset per row
// for all four channels, the channel in the scalar offset), explicit fma, no index decoding in the inner loops.
template <int NOUT, bool kMulti>   // NOUT output elements per thread; kMulti: outputs beyond NOUT * SEP_THREADS, in chunks
__global__ void __launch_bounds__(SEP_THREADS) k_roi_align3d_sep_fwd(const float* __restrict__ in,
                                                                     const float* __restrict__ rois,
                                                                     const int32_t* __restrict__ roi_inds, SepArgs A,
                                                                     float* __restrict__ out) {
  extern __shared__ f32x4 lds4[];
  int k, c0;
  if (!sep_assign(A, &k, &c0)) return;
  lfloat* lds = (lfloat*)lds4;
  const int t = threadIdx.x;
  const int W = A.W, L = A.L, H = A.H, ow = A.ow, ol = A.ol, oh = A.oh;
  lfloat* T[3];
  lint* first[3];
  lint* last[3];
  T[0] = lds; T[1] = T[0] + ow * W; T[2] = T[1] + ol * L;
  lint* ip = (lint*)(T[2] + oh * H);
  first[0] = ip; last[0] = first[0] + ow; first[1] = last[0] + ow; last[1] = first[1] + ol;
  first[2] = last[1] + ol; last[2] = first[2] + oh;
  lSepRoi* R = (lSepRoi*)(last[2] + oh);
  // the slab intermediates start at the next 16-byte boundary (LDS offsets: the dynamic window starts at 0)
  const int fixed_words = ow * W + ol * L + oh * H + 2 * (ow + ol + oh) + (int)(sizeof(SepRoi) / 4);
  lfloat* tmpbase = lds + ((fixed_words + 3) & ~3);

  sep_build_tables(rois + (int64_t)k * 6, A, R, T, first, last);

  const int olh = ol * oh, nout = ow * olh;
  const int WLH = W * L * H;
  const int cend = min(c0 + A.cpb, A.C);
  const int x0 = R->lo[0], y0 = R->lo[1], z0 = R->lo[2];
  const int sx = R->hi[0] - x0 + 1, sy = R->hi[1] - y0 + 1, sz = R->hi[2] - z0 + 1;
  float* obase = out + ((int64_t)k * A.C) * nout;
  if (sx <= 0 || sy <= 0 || sz <= 0) {       // no sample inside the volume: zeros
    for (int c = c0; c < cend; ++c)
      for (int o = t; o < nout; o += SEP_THREADS) obase[(int64_t)c * nout + o] = 0.0f;
    return;
  }
  const float inv_count = R->inv_count;
  // x slab: XB planes at a time so that tmp1 [XB][sy][oh] + tmp2 [XB][ol][oh] (x SEP_CH) fit the budget
  int XB = (A.tmp_floats - 4 * ow) / (SEP_CH * (sy * oh + olh));
  XB = max(1, min(XB, sx));
  lfloat4* xt = (lfloat4*)tmpbase;           // x taps per output index pw: four weights from its first cell on
  lfloat4* tmp1 = xt + ow;
  lfloat4* tmp2 = tmp1 + XB * sy * oh;
  const float inv_sy = 1.0f / (float)sy;

  // ---- z role: output index ph, row slot
  const int NS1 = SEP_THREADS / oh;
  const int slot1 = fdiv(t, 1.0f / (float)oh), ph1 = t - slot1 * oh;
  const bool act1 = slot1 < NS1;
  const int fz = first[2][ph1], lz = last[2][ph1];
  // the taps of an output index are consecutive cells: ONE 16-byte load per channel covers four of them (the vector
  // memory path charges a gather per 128-byte line an INSTRUCTION touches: four one-tap loads touched the same lines
  // four times); the window is pulled back from the end of the row, cells outside [fz, lz] get weight zero
  // (round 5: pulled back from the end of the RoI's sampled REGION where that has four cells - the window then never
  //  holds a cell no sample of this RoI touches, see the mask below - and from the end of the volume otherwise)
  const int z1r = R->hi[2];
  const bool tiny_z = sz < 4;                                    // RoI-uniform
  const int jz = tiny_z ? min(min(fz, H - 1), H - 4) : min(fz <= lz ? fz : z0, z1r - 3);
  float wz[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wz[j] = (jz + j >= fz && jz + j <= lz) ? T[2][ph1 * H + jz + j] : 0.0f;
  const bool morez = lz >= jz + 4;
  // Round-4 advisor: a window cell that NO sample of this RoI touches (before z0 when the window is pulled back from the
  // end of the row, behind z1 for the last output index) was multiplied by a zero weight - 0 x Inf = NaN where the
  // lane-per-output kernel, the oracle and torchvision never read the voxel.  Such cells are masked to +0 before the
  // multiply (four ANDs per row and channel, only in RoIs whose region is thinner than the window along z); cells INSIDE
  // the sampled region still meet zero weights of the output indices that do not reach them: a non-finite voxel inside
  // a RoI's region poisons that RoI's outputs along the row, as it poisons the ones that sample it anyway.
  uint32_t zm[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) zm[j] = (jz + j >= z0 && jz + j <= z1r) ? 0xFFFFFFFFu : 0u;
  const bool maskz = tiny_z;      // only regions thinner than the window can leave it: one uniform branch per row otherwise

  // ---- y role: (pl, ph), plane slot
  const int NS2 = SEP_THREADS / olh;
  const int slot2 = fdiv(t, 1.0f / (float)olh), r2 = t - slot2 * olh;
  const bool act2 = slot2 < NS2;
  const int pl2 = fdiv(r2, 1.0f / (float)oh), ph2 = r2 - pl2 * oh;
  const int fy = first[1][pl2], ly = last[1][pl2];
  const int fyc = fy <= ly ? fy : y0;
  float wy[4];
  const int ny = max(ly - fy, 0);             // tap j reads row fy + min(j, ny): always a row this RoI wrote
#pragma unroll
  for (int j = 0; j < 4; ++j) wy[j] = (fy + j <= ly) ? T[1][pl2 * L + fy + j] : 0.0f;
  const bool morey = ly >= fy + 4;

  // ---- x role: the output elements this thread owns
  // (the taps themselves stay in LDS as one 16-byte record per pw: sixteen more live registers cost a wave per SIMD)
  // Outputs beyond NOUT * SEP_THREADS (14 bins: 2744 elements) are handled in CHUNKS of whole pw planes (round 5): a chunk
  // is a smaller problem of its own - its pw planes tap a sub-range of the x cells, passes 1 and 2 run over that
  // sub-range only (one or two cells shared with the neighbouring chunk) - so the accumulators stay at NOUT per thread.
  // The round-4 kernel carried 8 or 16 outputs per thread instead: 184 / 256 VGPRs, two / ONE wave per SIMD, and 14 bins
  // cost 3.3-5x the 10-bin time for 2.7x the outputs (profiles/r04_NOTES.txt 11).
  const int pch = max(1, min(ow, (NOUT * SEP_THREADS) / olh));     // pw planes per chunk
  int orr[NOUT], opw[NOUT];
  bool morex = false;
  if (t < ow) {
    const int f = first[0][t], l = last[0][t];
    f32x4 w;
    w.x = (f <= l) ? T[0][t * W + f] : 0.0f;
    w.y = (f + 1 <= l) ? T[0][t * W + f + 1] : 0.0f;
    w.z = (f + 2 <= l) ? T[0][t * W + f + 2] : 0.0f;
    w.w = (f + 3 <= l) ? T[0][t * W + f + 3] : 0.0f;
    xt[t] = w;
    if (f > l) first[0][t] = x0;              // a row without taps: all weights zero, any cell of the region will do
    morex = l >= f + 4;
  }
  morex = __syncthreads_or(morex);            // uniform: the rare path is skipped with one scalar branch; and xt is visible

  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(in + (int64_t)roi_inds[k] * A.C * WLH), 0, (int)((int64_t)A.C * WLH * 4), 0x00020000);

  // the chunk's outputs and x cells (rows start and end monotonically in pw; rows without taps were pointed at x0 above)
  constexpr bool multi = kMulti;
  int o_lo = 0, o_hi = nout, cx0 = x0, cx1 = x0 + sx - 1;
  auto chunk_setup = [&](int p0) {
    const int p1 = min(p0 + pch, ow);
    o_lo = p0 * olh;
    o_hi = p1 * olh;
    cx0 = x0 + sx;
    cx1 = x0 - 1;
    for (int pp = p0; pp < p1; ++pp) {
      const int l = last[0][pp];
      if (l >= first[0][pp]) { cx0 = min(cx0, (int)first[0][pp]); cx1 = max(cx1, l); }
    }
    cx0 = max(cx0, x0);
    cx1 = min(cx1, x0 + sx - 1);
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int o = o_lo + t + i * SEP_THREADS;
      opw[i] = o < o_hi ? fdiv(o, 1.0f / (float)olh) : p0;
      orr[i] = o < o_hi ? o - opw[i] * olh : 0;
    }
  };
  if (!multi) chunk_setup(0);
  for (int c = c0; c < cend; c += SEP_CH) {
    // a run shorter than SEP_CH re-reads its last channel and skips the stores
    int soff[SEP_CH];
#pragma unroll
    for (int ch = 0; ch < SEP_CH; ++ch) soff[ch] = min(c + ch, A.C - 1) * WLH * 4;
   for (int p0 = 0; p0 < (kMulti ? ow : 1); p0 += pch) {
    if constexpr (kMulti) chunk_setup(p0);   // (a single chunk - every output up to 10 bins - was set up once, above)
    f32x4 acc[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int xs = cx0 - x0; xs <= cx1 - x0; xs += XB) {
      const int nx = min(XB, cx1 - x0 + 1 - xs);
      // ---- pass 1 (z): tmp1[x][y][ph] = sum_z Tz[ph][z] * in[x0+xs+x][y0+y][z]
      const int nrows = act1 ? nx * sy : 0;
      for (int r = slot1; r < nrows; r += NS1) {
        const int x = fdiv(r, inv_sy), y = r - x * sy;
        const int voff = (((x0 + xs + x) * L + (y0 + y)) * H + jz) * 4;
        u32x4 q[SEP_CH];
#pragma unroll
        for (int ch = 0; ch < SEP_CH; ++ch) q[ch] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff[ch], 0);
        if (maskz) {
#pragma unroll
          for (int ch = 0; ch < SEP_CH; ++ch) { q[ch].x &= zm[0]; q[ch].y &= zm[1]; q[ch].z &= zm[2]; q[ch].w &= zm[3]; }
        }
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        fma4(a, wz[0], f32x4{__uint_as_float(q[0].x), __uint_as_float(q[1].x), __uint_as_float(q[2].x), __uint_as_float(q[3].x)});
        fma4(a, wz[1], f32x4{__uint_as_float(q[0].y), __uint_as_float(q[1].y), __uint_as_float(q[2].y), __uint_as_float(q[3].y)});
        fma4(a, wz[2], f32x4{__uint_as_float(q[0].z), __uint_as_float(q[1].z), __uint_as_float(q[2].z), __uint_as_float(q[3].z)});
        fma4(a, wz[3], f32x4{__uint_as_float(q[0].w), __uint_as_float(q[1].w), __uint_as_float(q[2].w), __uint_as_float(q[3].w)});
        if (morez) {
          for (int j = jz + 4; j <= lz; ++j) {
            const float w = T[2][ph1 * H + j];
            const int vo = voff + (j - jz) * 4;
            fma4(a, w, f32x4{__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, soff[0], 0)),
                                   __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, soff[1], 0)),
                                   __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, soff[2], 0)),
                                   __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, soff[3], 0))});
          }
        }
        tmp1[r * oh + ph1] = a;
      }
      __syncthreads();
      // ---- pass 2 (y): tmp2[x][pl][ph] = sum_y Ty[pl][y] * tmp1[x][y - y0][ph]
      const int nx2 = act2 ? nx : 0;
      for (int x = slot2; x < nx2; x += NS2) {
        const lfloat4* src = tmp1 + (x * sy + fyc - y0) * oh + ph2;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) fma4(a, wy[j], src[min(j, ny) * oh]);
        if (morey)
          for (int j = fy + 4; j <= ly; ++j) fma4(a, T[1][pl2 * L + j], src[(j - fy) * oh]);
        tmp2[x * olh + r2] = a;
      }
      __syncthreads();
      // ---- pass 3 (x), this slab's share: acc[o] += sum_{x in slab} Tx[pw][x] * tmp2[x][pl][ph]
      const int xlo = x0 + xs, xhi = x0 + xs + nx - 1;
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const lfloat4* src = tmp2 + orr[i] - xlo * olh;
        const f32x4 w = xt[opw[i]];
        const int f = first[0][opw[i]];
        const float wj[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cell = f + j, cc = min(max(cell, xlo), xhi);
          fma4(acc[i], cell == cc ? wj[j] : 0.0f, src[cc * olh]);
        }
      }
      if (morex) {        // rows longer than the four register taps (sampling grid > 2): the rest from the table
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
          if (o_lo + t + i * SEP_THREADS >= o_hi) continue;
          const int pw = opw[i], lx = last[0][pw];
          const lfloat4* src = tmp2 + orr[i] - xlo * olh;
          for (int xx = max(first[0][pw] + 4, xlo); xx <= min(lx, xhi); ++xx) fma4(acc[i], T[0][pw * W + xx], src[xx * olh]);
        }
      }
      // no barrier here: the next slab's pass 1 writes tmp1 only, and its barrier orders the writes of tmp2 behind
      // these reads
    }
    // streaming stores (SEP_NT_STORE): the 262 MB of a configs[4] call would otherwise walk through the L2 and push out
    // the few channels of the volume its workgroups keep re-reading
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int o = o_lo + t + i * SEP_THREADS;
      if (o < o_hi) {
        float* dst = obase + (int64_t)c * nout + o;
        SEP_STORE(dst, acc[i].x * inv_count);
        if (c + 1 < cend) SEP_STORE(dst + nout, acc[i].y * inv_count);
        if (c + 2 < cend) SEP_STORE(dst + 2 * (int64_t)nout, acc[i].z * inv_count);
        if (c + 3 < cend) SEP_STORE(dst + 3 * (int64_t)nout, acc[i].w * inv_count);
      }
    }
    // (no barrier between chunks either: the next chunk's pass 1 writes tmp1 only, and its first barrier is reached by
    //  a wave after it has finished this chunk's pass 3)
   }
#if SEP_RUN_BARRIER
    __syncthreads();
#endif
    // no barrier between channel runs: the next run's pass 1 writes tmp1, which nobody reads any more (every wave left
    // this run's pass 2 through the barrier behind it), and its pass 2 writes tmp2 only behind the NEXT barrier, which
    // a wave reaches after it has finished this run's pass 3
  }
}

// Backward = the transpose, pass by pass: gout [ow][ol][oh] -(x^T)-> [sx][ol][oh] -(y^T)-> [sx][sy][oh] -(z^T)->
// [sx][sy][sz], added into grad_input with one atomic per touched cell and channel (the lane-per-output kernel
// issues 8 g^3 per output element); lanes run along z, so the atomics of a wave hit adjacent addresses.
// What bounds it (round 5, tools/micro/roialign_bench bwd, configs[4]): with the fixed roles below the three passes take
// 0.13-0.16 ms (plain stores instead of atomics: 0.16; the first version: 0.40 / 0.45) and the launch still takes 0.58 ms
// whatever the channels per workgroup (8..32) or the LDS budget (2..5 workgroups per CU): a region row along z is
// ~10 floats = 1.56 64-byte segments per (x, y, channel), 10.7 M atomic requests per call, and the memory-side atomic
// unit takes ~21 G requests/s whatever their width (profiles/r03_NOTES.txt 1) = 0.51 ms.  Fewer requests need another
// layout of the accumulation target: k_roi_align3d_sep_bwd_cl below (channels fastest: one 64-byte request per voxel and
// 16 channels, plus a transposing copy into grad_input; 0.34 ms for the whole call) is what inr_roi_align_3d_backward_ws
// runs; this kernel stays behind inr_roi_align_3d_backward (no workspace, accumulates in place, any channel count).
__global__ void __launch_bounds__(SEP_THREADS) k_roi_align3d_sep_bwd(const float* __restrict__ gout,
                                                                     const float* __restrict__ rois,
                                                                     const int32_t* __restrict__ roi_inds, SepArgs A,
                                                                     float* __restrict__ gin) {
  extern __shared__ f32x4 lds4[];
  int k, c0;
  if (!sep_assign(A, &k, &c0)) return;
  lfloat* lds = (lfloat*)lds4;
  const int t = threadIdx.x;
  const int W = A.W, L = A.L, H = A.H, ow = A.ow, ol = A.ol, oh = A.oh;
  lfloat* T[3];
  lint* first[3];
  lint* last[3];
  T[0] = lds; T[1] = T[0] + ow * W; T[2] = T[1] + ol * L;
  lint* ip = (lint*)(T[2] + oh * H);
  first[0] = ip; last[0] = first[0] + ow; first[1] = last[0] + ow; last[1] = first[1] + ol;
  first[2] = last[1] + ol; last[2] = first[2] + oh;
  // per cell: the output indices whose rows reach it (rows start and end monotonically in p)
  lint* pfirst[3];
  lint* plast[3];
  pfirst[0] = last[2] + oh; plast[0] = pfirst[0] + W; pfirst[1] = plast[0] + W; plast[1] = pfirst[1] + L;
  pfirst[2] = plast[1] + L; plast[2] = pfirst[2] + H;
  lSepRoi* R = (lSepRoi*)(plast[2] + H);
  const int fixed_words = ow * W + ol * L + oh * H + 2 * (ow + ol + oh) + 2 * (W + L + H) + (int)(sizeof(SepRoi) / 4);
  lfloat* tmpbase = lds + ((fixed_words + 3) & ~3);

  sep_build_tables(rois + (int64_t)k * 6, A, R, T, first, last);
  {
    const int n[3] = {W, L, H};
    const int o[3] = {ow, ol, oh};
    for (int r = t; r < W + L + H; r += SEP_THREADS) {
      const int a = r < W ? 0 : (r < W + L ? 1 : 2);
      const int cell = r - (a == 0 ? 0 : (a == 1 ? W : W + L));
      int f = o[a], l = -1;
      for (int p = 0; p < o[a]; ++p)
        if (first[a][p] <= cell && cell <= last[a][p]) { f = min(f, p); l = max(l, p); }
      pfirst[a][cell] = f;
      plast[a][cell] = l;
      (void)n;
    }
  }
  __syncthreads();

  const int olh = ol * oh, nout = ow * olh;
  const int64_t WLH = (int64_t)W * L * H;
  const int cend = min(c0 + A.cpb, A.C);
  const int x0 = R->lo[0], y0 = R->lo[1], z0 = R->lo[2];
  const int sx = R->hi[0] - x0 + 1, sy = R->hi[1] - y0 + 1, sz = R->hi[2] - z0 + 1;
  if (sx <= 0 || sy <= 0 || sz <= 0) return;
  const float inv_count = R->inv_count;
  // LDS: go [ow][ol][oh], then per x slab t2 [XB][ol][oh] and t1 [XB][sy][oh]
  lfloat4* go = (lfloat4*)tmpbase;
  int XB = (A.tmp_floats - SEP_CH * nout) / (SEP_CH * (sy * oh + olh));
  XB = max(1, min(XB, sx));
  lfloat4* t2 = go + nout;
  lfloat4* t1 = t2 + XB * olh;
  // integer division in the index decodes below (round-4 advisor: the float-reciprocal fdiv() is exact only for small
  // operands, and nothing bounds the extents here - L, H up to what the LDS window holds)
  const float* gbase = gout + ((int64_t)k * A.C) * nout;
  float* vol0 = gin + ((int64_t)roi_inds[k] * A.C) * WLH;

  // ---- roles (round 5).  The first version of this kernel decoded (x, y, z) from a flat item index in every pass - two
  // integer divisions per item, ~60 vector instructions around ~20 useful ones - and read taps and tap ranges from the
  // tables per item.  Now a thread's role is fixed for the whole RoI, as in the forward: (pl, ph) in the x^T pass, cell y
  // and ph in the y^T pass, cell z in the z^T pass; the taps of its cell (first output index + four weights; longer
  // ranges - boxes thinner than the output - take the rest from the table) sit in registers and the loops run over
  // x planes / (x, y) rows with a carried increment.  Needs ol*oh, sy*oh and sz within the workgroup (every RoI up to
  // 25 cells across at 10 bins); other RoIs take the generic loops below.
  const bool fast = olh <= SEP_THREADS && sy * oh <= SEP_THREADS && sz <= SEP_THREADS;      // RoI-uniform
  int NSA = 1, slotA = 0, rA = 0, NSB = 1, slotB = 0, yB = 0, phB = 0, pfB = 0, plB = -1, nB = 0;
  int NSC = 1, slotC = 0, zC = 0, pfC = 0, plC = -1, nC = 0, xC0 = 0, yC0 = 0, dxC = 0, dyC = 0;
  bool actA = false, actB = false, actC = false, moreB = false, moreC = false;
  float wB[4] = {0.f, 0.f, 0.f, 0.f}, wC[4] = {0.f, 0.f, 0.f, 0.f};
  if (fast) {
    NSA = SEP_THREADS / olh; slotA = t / olh; rA = t - slotA * olh; actA = slotA < NSA;
    const int syoh = sy * oh;
    NSB = SEP_THREADS / syoh; slotB = t / syoh;
    const int rB = t - slotB * syoh;
    yB = rB / oh; phB = rB - yB * oh; actB = slotB < NSB;
    {
      const int cell = y0 + yB;
      const int pf = pfirst[1][cell], pl = plast[1][cell];
      pfB = pf <= pl ? pf : 0; plB = pf <= pl ? pl : -1;
      nB = max(plB - pfB, 0);                  // tap j reads output index pfB + min(j, nB): always one that reaches the cell
#pragma unroll
      for (int j = 0; j < 4; ++j) wB[j] = (pfB + j <= plB) ? T[1][(pfB + j) * L + cell] : 0.0f;
      moreB = plB >= pfB + 4;
    }
    NSC = SEP_THREADS / sz; slotC = t / sz; zC = t - slotC * sz; actC = slotC < NSC;
    {
      const int cell = z0 + zC;
      const int pf = pfirst[2][cell], pl = plast[2][cell];
      pfC = pf <= pl ? pf : 0; plC = pf <= pl ? pl : -1;
      nC = max(plC - pfC, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) wC[j] = (pfC + j <= plC) ? T[2][(pfC + j) * H + cell] : 0.0f;
      moreC = plC >= pfC + 4;
    }
    xC0 = slotC / sy; yC0 = slotC - xC0 * sy;     // row = x * sy + y, rows slotC, slotC + NSC, ...
    dxC = NSC / sy; dyC = NSC - dxC * sy;
  }

  // The gout values of the NEXT channel run are requested before this run's passes (round 5): the kernel is a chain of
  // dependent steps per run - global read, barrier, three LDS passes - and the read used to start only when the
  // previous run had finished.  Up to GOP elements per thread ride in registers (outputs up to 1024 elements: 10^3); larger
  // outputs read in place as before.
  constexpr int GOP = 4;
  const bool pre_ok = nout <= GOP * SEP_THREADS;
  f32x4 gnext[GOP];
  auto go_request = [&](int c) {
#pragma unroll
    for (int i = 0; i < GOP; ++i) {
      const int o = min(t + i * SEP_THREADS, nout - 1);
      const float* src = gbase + (int64_t)c * nout + o;
      f32x4 g;
      g.x = src[0];
      g.y = src[(int64_t)min(1, cend - 1 - c) * nout];
      g.z = src[(int64_t)min(2, cend - 1 - c) * nout];
      g.w = src[(int64_t)min(3, cend - 1 - c) * nout];
      gnext[i] = g;
      if ((i + 1) * SEP_THREADS >= nout) break;
    }
  };
  if (pre_ok) go_request(c0);
  for (int c = c0; c < cend; c += SEP_CH) {
    if (pre_ok) {
#pragma unroll
      for (int i = 0; i < GOP; ++i) {
        const int o = t + i * SEP_THREADS;
        if (o < nout) {
          f32x4 g = gnext[i];
          g.x *= inv_count;
          g.y = c + 1 < cend ? g.y * inv_count : 0.0f;
          g.z = c + 2 < cend ? g.z * inv_count : 0.0f;
          g.w = c + 3 < cend ? g.w * inv_count : 0.0f;
          go[o] = g;
        }
        if ((i + 1) * SEP_THREADS >= nout) break;
      }
      if (c + SEP_CH < cend) go_request(c + SEP_CH);
    } else {
      for (int o = t; o < nout; o += SEP_THREADS) {
        const float* src = gbase + (int64_t)c * nout + o;
        f32x4 g;
        g.x = src[0] * inv_count;
        g.y = c + 1 < cend ? src[nout] * inv_count : 0.0f;
        g.z = c + 2 < cend ? src[2 * (int64_t)nout] * inv_count : 0.0f;
        g.w = c + 3 < cend ? src[3 * (int64_t)nout] * inv_count : 0.0f;
        go[o] = g;
      }
    }
    __syncthreads();
    if (fast) {
    // ---- fixed roles (round 5): no index decoding and no table look-ups in the inner loops (see the roles above)
    for (int xs = 0; xs < sx; xs += XB) {
      const int nx = min(XB, sx - xs);
      // ---- x^T: t2[x][pl][ph] = sum_pw Tx[pw][x] * go[pw][pl][ph]       (x is wave-uniform up to the slot: broadcasts)
      const int nxA = actA ? nx : 0;
      for (int x = slotA; x < nxA; x += NSA) {
        const int cell = x0 + xs + x;
        const int pf = pfirst[0][cell], pl = plast[0][cell];
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = pf; p <= pl; ++p) fma4(a, T[0][p * W + cell], go[p * olh + rA]);
        t2[x * olh + rA] = a;
      }
      __syncthreads();
      // ---- y^T: t1[x][y][ph] = sum_pl Ty[pl][y] * t2[x][pl][ph]        (taps of cell y in registers)
      const int nxB = actB ? nx : 0;
      for (int x = slotB; x < nxB; x += NSB) {
        const lfloat4* src = t2 + (x * ol + pfB) * oh + phB;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) fma4(a, wB[j], src[min(j, nB) * oh]);
        if (moreB)
          for (int p = pfB + 4; p <= plB; ++p) fma4(a, T[1][p * L + y0 + yB], src[(p - pfB) * oh]);
        t1[(x * sy + yB) * oh + phB] = a;
      }
      __syncthreads();
      // ---- z^T: grad_input[x][y][z] += sum_ph Tz[ph][z] * t1[x][y][ph]   (taps of cell z in registers; lanes along z)
      const int nrowsC = actC ? nx * sy : 0;
      int x = xC0, y = yC0;
      for (int row = slotC; row < nrowsC; row += NSC) {
        const lfloat4* src = t1 + row * oh + pfC;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) fma4(a, wC[j], src[min(j, nC)]);
        if (moreC)
          for (int p = pfC + 4; p <= plC; ++p) fma4(a, T[2][p * H + z0 + zC], src[p - pfC]);
        float* dst = vol0 + (int64_t)c * WLH + ((x0 + xs + x) * L + (y0 + y)) * H + z0 + zC;
        atomicAdd(dst, a.x);
        if (c + 1 < cend) atomicAdd(dst + WLH, a.y);
        if (c + 2 < cend) atomicAdd(dst + 2 * WLH, a.z);
        if (c + 3 < cend) atomicAdd(dst + 3 * WLH, a.w);
        y += dyC; x += dxC;
        if (y >= sy) { y -= sy; ++x; }
      }
    }
    } else {
    // ---- any extents: every item decodes its indices and reads its taps from the tables
    for (int xs = 0; xs < sx; xs += XB) {
      const int nx = min(XB, sx - xs);
      // ---- x^T: t2[x][pl][ph] = sum_pw Tx[pw][x] * go[pw][pl][ph]
      const int n2 = nx * olh;
      for (int item = t; item < n2; item += SEP_THREADS) {
        const int x = item / olh, r = item - x * olh;
        const int cell = x0 + xs + x;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = pfirst[0][cell]; p <= plast[0][cell]; ++p) {
          const float w = T[0][p * W + cell];
          const f32x4 v = go[p * olh + r];
          a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
        }
        t2[item] = a;
      }
      __syncthreads();
      // ---- y^T: t1[x][y][ph] = sum_pl Ty[pl][y] * t2[x][pl][ph]
      const int n1 = nx * sy * oh;
      for (int item = t; item < n1; item += SEP_THREADS) {
        const int row = item / oh, ph = item - row * oh;
        const int x = row / sy, y = row - x * sy;
        const int cell = y0 + y;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = pfirst[1][cell]; p <= plast[1][cell]; ++p) {
          const float w = T[1][p * L + cell];
          const f32x4 v = t2[(x * ol + p) * oh + ph];
          a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
        }
        t1[item] = a;
      }
      __syncthreads();
      // ---- z^T: grad_input[x][y][z] += sum_ph Tz[ph][z] * t1[x][y][ph]
      const int n0 = nx * sy * sz;
      for (int item = t; item < n0; item += SEP_THREADS) {
        const int row = item / sz, z = item - row * sz;
        const int x = row / sy, y = row - x * sy;
        const int cell = z0 + z;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = pfirst[2][cell]; p <= plast[2][cell]; ++p) {
          const float w = T[2][p * H + cell];
          const f32x4 v = t1[row * oh + p];
          a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
        }
        float* dst = vol0 + (int64_t)c * WLH + ((x0 + xs + x) * L + (y0 + y)) * H + cell;
        atomicAdd(dst, a.x);
        if (c + 1 < cend) atomicAdd(dst + WLH, a.y);
        if (c + 2 < cend) atomicAdd(dst + 2 * WLH, a.z);
        if (c + 3 < cend) atomicAdd(dst + 3 * WLH, a.w);
      }
      // t2 is rewritten by the next slab's first pass while slow waves may still read t1 here, never t2: fine; t1 is
      // rewritten after that pass's barrier
    }
    }
    __syncthreads();   // go is reloaded
  }
}

// ---- backward with a channels-fastest accumulation target (round 5) ---------------------------------------------------
// The backward above is bound by the memory-side atomic unit, which takes ~21 G requests of up to 64 bytes per second
// whatever they carry: a region row along z of the gradient's own [C][W][L][H] layout is ~10 floats in 1.56 segments.
// Accumulating into a scratch volume laid out [W][L][H][C] lets one instruction cover the 16 channels of a workgroup at
// a voxel with ONE full 64-byte request (tools/micro/atomic_layout_bench.hip: the atomics of a configs[4] call alone
// 0.64 -> 0.30 ms, 0.25 being the floor of packed requests); a transposing copy then writes grad_input once (no zero
// fill of grad_input, no accumulation into it: the call OVERWRITES it).  configs[4]: 0.293 ms + 0.015 (zero fill of the
// scratch) + 0.030 (copy) = 0.34 ms against 0.58 + 0.02 in place.  The passes are the ones above with 16 channels
// in flight per slab instead of 4: x^T reads the output gradient straight from global memory (its 64 KB per workgroup
// do not fit beside the slabs), y^T as above, z^T with one channel per lane (16 lanes = the 16 channels of a voxel).
__global__ void __launch_bounds__(SEP_THREADS) k_roi_align3d_sep_bwd_cl(const float* __restrict__ gout,
                                                                        const float* __restrict__ rois,
                                                                        const int32_t* __restrict__ roi_inds, SepArgs A,
                                                                        float* __restrict__ gt) {
  extern __shared__ f32x4 lds4[];
  int k, c0;
  if (!sep_assign(A, &k, &c0)) return;
  lfloat* lds = (lfloat*)lds4;
  const int t = threadIdx.x;
  const int W = A.W, L = A.L, H = A.H, ow = A.ow, ol = A.ol, oh = A.oh;
  lfloat* T[3];
  lint* first[3];
  lint* last[3];
  T[0] = lds; T[1] = T[0] + ow * W; T[2] = T[1] + ol * L;
  lint* ip = (lint*)(T[2] + oh * H);
  first[0] = ip; last[0] = first[0] + ow; first[1] = last[0] + ow; last[1] = first[1] + ol;
  first[2] = last[1] + ol; last[2] = first[2] + oh;
  lint* pfirst[3];
  lint* plast[3];
  pfirst[0] = last[2] + oh; plast[0] = pfirst[0] + W; pfirst[1] = plast[0] + W; plast[1] = pfirst[1] + L;
  pfirst[2] = plast[1] + L; plast[2] = pfirst[2] + H;
  lSepRoi* R = (lSepRoi*)(plast[2] + H);
  const int fixed_words = ow * W + ol * L + oh * H + 2 * (ow + ol + oh) + 2 * (W + L + H) + (int)(sizeof(SepRoi) / 4);
  lfloat* tmpbase = lds + ((fixed_words + 3) & ~3);

  sep_build_tables(rois + (int64_t)k * 6, A, R, T, first, last);
  {
    const int o[3] = {ow, ol, oh};
    for (int r = t; r < W + L + H; r += SEP_THREADS) {
      const int a = r < W ? 0 : (r < W + L ? 1 : 2);
      const int cell = r - (a == 0 ? 0 : (a == 1 ? W : W + L));
      int f = o[a], l = -1;
      for (int p = 0; p < o[a]; ++p)
        if (first[a][p] <= cell && cell <= last[a][p]) { f = min(f, p); l = max(l, p); }
      pfirst[a][cell] = f;
      plast[a][cell] = l;
    }
  }
  __syncthreads();

  constexpr int RUNS = 4;                       // channel runs of SEP_CH in flight: cpb = 16 channels per workgroup
  const int olh = ol * oh, nout = ow * olh;
  const int64_t WLH = (int64_t)W * L * H;
  const int x0 = R->lo[0], y0 = R->lo[1], z0 = R->lo[2];
  const int sx = R->hi[0] - x0 + 1, sy = R->hi[1] - y0 + 1, sz = R->hi[2] - z0 + 1;
  if (sx <= 0 || sy <= 0 || sz <= 0) return;
  const float inv_count = R->inv_count;
  const int syoh = sy * oh;
  int XB = A.tmp_floats / (SEP_CH * RUNS * (olh + syoh));
  XB = max(1, min(XB, sx));
  lfloat4* t2 = (lfloat4*)tmpbase;              // [RUNS][XB][ol][oh]
  lfloat4* t1 = t2 + RUNS * XB * olh;           // [RUNS][XB][sy][oh]
  const float* gbase = gout + ((int64_t)k * A.C + c0) * nout;
  float* vol = gt + ((int64_t)roi_inds[k] * WLH) * A.C + c0;

  // x^T role: (pl, ph) fixed, slots over x (host: ol * oh <= SEP_THREADS)
  const int NSA = SEP_THREADS / olh, slotA = t / olh, rA = t - slotA * olh;
  const bool actA = slotA < NSA;
  // y^T roles: (y, ph); one per thread with slots over (run, x) when sy * oh fits the workgroup, else several per thread
  const bool oneB = syoh <= SEP_THREADS;
  const int NSB = oneB ? SEP_THREADS / syoh : 1, slotB = oneB ? t / syoh : 0, roleB0 = oneB ? t - slotB * syoh : t;
  const bool actB = slotB < NSB;
  // z^T role: channel (16 lanes = the workgroup's 16 channels), z slot
  const int chC = t & 15, runC = chC >> 2, compC = chC & 3;

  for (int xs = 0; xs < sx; xs += XB) {
    const int nx = min(XB, sx - xs);
    // ---- x^T: t2[run][x][pl][ph] = sum_pw Tx[pw][x] * gout[run][pw][pl][ph] / count     (gout from global memory)
    if (actA) {
      for (int x = slotA; x < nx; x += NSA) {
        const int cell = x0 + xs + x;
        const int pf0 = pfirst[0][cell], pl0 = plast[0][cell];
        const int pf = pf0 <= pl0 ? pf0 : 0, pl = pf0 <= pl0 ? pl0 : -1, np = max(pl - pf, 0);
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (pf + j <= pl) ? T[0][(pf + j) * W + cell] * inv_count : 0.0f;
#pragma unroll
        for (int run = 0; run < RUNS; ++run) {
          const float* src = gbase + (int64_t)(run * SEP_CH) * nout + rA;
          float g[4][SEP_CH];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int po = (pf + min(j, np)) * olh;          // tap j reads an output index that reaches the cell
#pragma unroll
            for (int ch = 0; ch < SEP_CH; ++ch) g[j][ch] = src[(int64_t)ch * nout + po];
          }
          f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) fma4(a, w[j], f32x4{g[j][0], g[j][1], g[j][2], g[j][3]});
          for (int p = pf + 4; p <= pl; ++p) {
            const float wp = T[0][p * W + cell] * inv_count;
            fma4(a, wp, f32x4{src[p * olh], src[(int64_t)nout + p * olh], src[2 * (int64_t)nout + p * olh],
                              src[3 * (int64_t)nout + p * olh]});
          }
          t2[(run * XB + x) * olh + rA] = a;
        }
      }
    }
    __syncthreads();
    // ---- y^T: t1[run][x][y][ph] = sum_pl Ty[pl][y] * t2[run][x][pl][ph]
    if (actB) {
      for (int role = roleB0; role < syoh; role += SEP_THREADS) {
        const int yB = role / oh, phB = role - yB * oh;
        const int cell = y0 + yB;
        const int pf0 = pfirst[1][cell], pl0 = plast[1][cell];
        const int pf = pf0 <= pl0 ? pf0 : 0, pl = pf0 <= pl0 ? pl0 : -1, np = max(pl - pf, 0);
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (pf + j <= pl) ? T[1][(pf + j) * L + cell] : 0.0f;
        for (int q = slotB; q < RUNS * nx; q += NSB) {
          const int run = q / nx, x = q - run * nx;
          const lfloat4* src = t2 + ((run * XB + x) * ol + pf) * oh + phB;
          f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) fma4(a, w[j], src[min(j, np) * oh]);
          for (int p = pf + 4; p <= pl; ++p) fma4(a, T[1][p * L + cell], src[(p - pf) * oh]);
          t1[((run * XB + x) * sy + yB) * oh + phB] = a;
        }
      }
    }
    __syncthreads();
    // ---- z^T: scratch[x][y][z][c] += sum_ph Tz[ph][z] * t1[run(c)][x][y][ph][c & 3]     (one 64-byte request per voxel)
    for (int z = t >> 4; z < sz; z += SEP_THREADS / 16) {
      const int cell = z0 + z;
      const int pf0 = pfirst[2][cell], pl0 = plast[2][cell];
      const int pf = pf0 <= pl0 ? pf0 : 0, pl = pf0 <= pl0 ? pl0 : -1, np = max(pl - pf, 0);
      float w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (pf + j <= pl) ? T[2][(pf + j) * H + cell] : 0.0f;
      const lfloat* src0 = (const lfloat*)(t1 + runC * XB * syoh) + compC + 4 * pf;
      const int o1 = 4 * min(1, np), o2 = 4 * min(2, np), o3 = 4 * min(3, np);
      for (int x = 0; x < nx; ++x) {
        float* dst = vol + ((int64_t)((x0 + xs + x) * L + y0) * H + cell) * A.C + chC;
        const lfloat* src = src0 + 4 * (x * syoh);
        for (int y = 0; y < sy; ++y) {
          float a = w[0] * src[0];
          a = __builtin_fmaf(w[1], src[o1], a);
          a = __builtin_fmaf(w[2], src[o2], a);
          a = __builtin_fmaf(w[3], src[o3], a);
          for (int p = pf + 4; p <= pl; ++p) a = __builtin_fmaf(T[2][p * H + cell], src[4 * (p - pf)], a);
          atomicAdd(dst, a);
          src += 4 * oh;
          dst += (int64_t)H * A.C;
        }
      }
    }
    // the next slab's x^T pass writes t2 only; its barrier orders the rewrite of t1 behind these reads
  }
}

// scratch [N][V][C] -> grad_input [N][C][V] (V = W*L*H): 32 x 32 tiles through LDS, both sides 128-byte rows
__global__ void __launch_bounds__(256) k_channels_last_to_planes(const float* __restrict__ src, float* __restrict__ dst,
                                                                 int64_t V, int C) {
  __shared__ float tile[32][33];
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const float* s = src + (int64_t)blockIdx.z * V * C;
  float* d = dst + (int64_t)blockIdx.z * V * C;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t v = v0 + ty + 8 * i;
    const int c = c0 + tx;
    tile[ty + 8 * i][tx] = (v < V && c < C) ? s[v * C + c] : 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i;
    const int64_t v = v0 + tx;
    if (v < V && c < C) d[(int64_t)c * V + v] = tile[tx][ty + 8 * i];
  }
}

static int g_roi_mode = 0;   // 0 auto, 1 one lane per output element, 2 separable (error if it does not fit)

// LDS bytes of the separable kernels for these extents (0: does not fit the device's 64 KB default window)
static int sep_lds_bytes(int W, int L, int H, int ow, int ol, int oh, bool bwd, int* tmp_floats) {
  const int64_t tab = (int64_t)ow * W + (int64_t)ol * L + (int64_t)oh * H;
  const int64_t ints = 2 * (int64_t)(ow + ol + oh) + (bwd ? 2 * (int64_t)(W + L + H) : 0);
  const int64_t fixed = (tab + ints) * 4 + (int64_t)sizeof(SepRoi) + 16;
  // one x plane of the widest region must fit: SEP_CH * (L*oh + ol*oh) floats (+ gout in the backward)
  const int64_t need = (int64_t)SEP_CH * ((int64_t)L * oh + (int64_t)ol * oh) +
                       (bwd ? (int64_t)SEP_CH * ow * ol * oh : 4 * (int64_t)ow);
  int64_t tmp = std::max<int64_t>(need, SEP_TMP_FLOATS + (bwd ? (int64_t)SEP_CH * ow * ol * oh : 0));
  const int64_t total = fixed + tmp * 4;
  if (total > 64 * 1024) {
    tmp = need;
    if (fixed + tmp * 4 > 64 * 1024) return 0;
  }
  *tmp_floats = (int)tmp;
  return (int)(fixed + tmp * 4);
}

static int sep_channels_per_block(int C, int64_t K) {
  // enough workgroups to fill the chip several times over (256 CUs x 3 resident), whole SEP_CH runs; the tables of a
  // RoI are built once per workgroup, so longer runs amortise them (16 channels: 0.152 ms, 8: 0.168, 32: 0.162)
  int cpb = 16;
  while (cpb > SEP_CH && K * ((C + cpb - 1) / cpb) < 4096) cpb /= 2;
  return cpb;
}

static bool sep_grid_fits(int C, int64_t K) {
  if (K >= (1 << 24)) return false;
  const int cpb = sep_channels_per_block(C, K);
  const int64_t ngroups = (C + cpb - 1) / cpb;
  return 8 * K * ((ngroups + 7) / 8) < (1ll << 31);
}

}  // namespace inr

using namespace inr;

extern "C" {

int inr_roi_align_3d_set_mode(int32_t mode) {
  if (mode < 0 || mode > 3) {
    set_error("inr_roi_align_3d_set_mode: mode must be 0 (auto), 1 (lane per output), 2 (separable) or 3 (separable, no workspace)");
    return INR_EINVAL;
  }
  g_roi_mode = mode;
  return INR_OK;
}

int inr_roi_align_3d_forward(const float* input, const float* rois, const int32_t* roi_inds, int32_t N, int32_t C,
                             int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w, int32_t out_l, int32_t out_h,
                             float spatial_scale, float* out, inr_stream_t s) {
  INR_REQUIRE(K >= 0 && N >= 0 && C > 0 && W > 0 && L > 0 && H > 0 && out_w > 0 && out_l > 0 && out_h > 0, "bad sizes");
  if (K == 0) return INR_OK;
  INR_REQUIRE(input && rois && roi_inds && out && N > 0, "null pointer");
  const int64_t nout = (int64_t)out_w * out_l * out_h;
  const int64_t total = K * C * nout;
  int tmp_floats = 0;
  const bool sep_ok = nout <= 16 * SEP_THREADS && (int64_t)out_l * out_h <= SEP_THREADS && H >= 4 &&
                      sep_grid_fits(C, K) && (int64_t)C * W * L * H * 4 < (1ll << 31);
  const int lds = sep_ok ? sep_lds_bytes(W, L, H, out_w, out_l, out_h, false, &tmp_floats) : 0;
  INR_REQUIRE(g_roi_mode != 2 || lds > 0, "separable kernel: tables do not fit the LDS window for these extents");
  if (g_roi_mode != 1 && lds > 0) {
    SepArgs A;
    A.C = C; A.W = W; A.L = L; A.H = H; A.ow = out_w; A.ol = out_l; A.oh = out_h; A.scale = spatial_scale;
    A.cpb = sep_channels_per_block(C, K);
    A.ngroups = (C + A.cpb - 1) / A.cpb;
    A.K = (int)K;
    A.tmp_floats = tmp_floats;
    const unsigned grid = 8u * (unsigned)K * (unsigned)((A.ngroups + 7) / 8);
    // four outputs per thread; larger outputs (11 bins and up) in chunks of whole pw planes inside the kernel
    if (nout <= 4 * SEP_THREADS)
      k_roi_align3d_sep_fwd<4, false><<<grid, SEP_THREADS, lds, as_stream(s)>>>(input, rois, roi_inds, A, out);
    else
      k_roi_align3d_sep_fwd<4, true><<<grid, SEP_THREADS, lds, as_stream(s)>>>(input, rois, roi_inds, A, out);
    return check_launch("roi_align_3d_forward (separable)");
  }
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
  const int64_t nout = (int64_t)out_w * out_l * out_h;
  const int64_t total = K * C * nout;
  int tmp_floats = 0;
  const int lds = (sep_grid_fits(C, K) && (int64_t)W * L * H < (1ll << 30))
                      ? sep_lds_bytes(W, L, H, out_w, out_l, out_h, true, &tmp_floats) : 0;
  INR_REQUIRE(g_roi_mode != 2 || lds > 0, "separable kernel: tables do not fit the LDS window for these extents");
  if (g_roi_mode != 1 && lds > 0) {
    SepArgs A;
    A.C = C; A.W = W; A.L = L; A.H = H; A.ow = out_w; A.ol = out_l; A.oh = out_h; A.scale = spatial_scale;
    A.cpb = sep_channels_per_block(C, K);
    A.ngroups = (C + A.cpb - 1) / A.cpb;
    A.K = (int)K;
    A.tmp_floats = tmp_floats;
    const unsigned grid = 8u * (unsigned)K * (unsigned)((A.ngroups + 7) / 8);
    k_roi_align3d_sep_bwd<<<grid, SEP_THREADS, lds, as_stream(s)>>>(grad_out, rois, roi_inds, A, grad_input);
    return check_launch("roi_align_3d_backward (separable)");
  }
  k_roi_align3d_bwd<<<blocks_for(total, 256), 256, 0, as_stream(s)>>>(grad_out, rois, roi_inds, C, W, L, H, total,
                                                                      out_w, out_l, out_h, spatial_scale, grad_input);
  return check_launch("roi_align_3d_backward");
}

// LDS of the channels-last backward: the backward's tables + the two slab intermediates for 16 channels (no staged gout)
static int sep_cl_lds_bytes(int W, int L, int H, int ow, int ol, int oh, int* tmp_floats) {
  const int64_t tab = (int64_t)ow * W + (int64_t)ol * L + (int64_t)oh * H;
  const int64_t ints = 2 * (int64_t)(ow + ol + oh) + 2 * (int64_t)(W + L + H);
  const int64_t fixed = (tab + ints) * 4 + (int64_t)sizeof(SepRoi) + 16;
  const int64_t need = 4 * (int64_t)SEP_CH * ((int64_t)L * oh + (int64_t)ol * oh);     // one x plane of the widest region
  // 8192 floats of slabs: ~39 KB in all, four workgroups per CU (tools/micro/roialign_bench bwd: 0.33 ms against 0.35 at
  // 10240 and 0.37-0.39 at 6144 / 4096, where the slabs get thin)
  int64_t tmp = std::max<int64_t>(need, 8192);
  if (fixed + tmp * 4 > 64 * 1024) {
    tmp = need;
    if (fixed + tmp * 4 > 64 * 1024) return 0;
  }
  *tmp_floats = (int)tmp;
  return (int)(fixed + tmp * 4);
}

int64_t inr_roi_align_3d_backward_workspace_bytes(int32_t N, int32_t C, int32_t W, int32_t L, int32_t H, int64_t K,
                                                  int32_t out_w, int32_t out_l, int32_t out_h) {
  if (N <= 0 || C <= 0 || W <= 0 || L <= 0 || H <= 0 || K <= 0 || out_w <= 0 || out_l <= 0 || out_h <= 0) return 0;
  if (g_roi_mode == 1 || g_roi_mode == 3) return 0;
  int tmp_floats = 0;
  const int64_t V = (int64_t)W * L * H;
  // 16 channels per workgroup = one 64-byte request per voxel; every output column (pl, ph) needs a thread in the x^T pass
  if (C % 16 != 0 || (int64_t)out_l * out_h > SEP_THREADS || V >= (1ll << 30) || (V + 31) / 32 >= (1ll << 31) ||
      (C + 31) / 32 > 65535 || N > 65535 || K >= (1 << 24) || 8 * K * (((int64_t)C / 16 + 7) / 8) >= (1ll << 31))
    return 0;
  if (sep_cl_lds_bytes(W, L, H, out_w, out_l, out_h, &tmp_floats) == 0) return 0;
  return (int64_t)N * C * V * 4;
}

// Cost model of the two backward forms (round-5 advisor: the workspace form was taken wherever it was AVAILABLE, and it
// adds a zero fill of the scratch volume plus a transposing copy - three passes over N*C*V floats - to every call: 2.2x
// slower than in place for 64 small boxes on a [1,256,80^3] pyramid level, 2.7x faster for 512 large boxes on [1,256,20^3]).
// Both forms are bound by the memory-side atomic unit; in place a request carries a run of ~6 floats along H of one
// channel, the channels-fastest scratch packs 16 channels of one voxel into it.  Times in ms as linear models in
//   O = K * bins * C (output elements), R = voxels inside the RoIs' regions summed over the RoIs, N*C*V (volume passes),
//   K * bins (per-RoI table set-up, which the 16-channel groups of the workspace kernel amortise worse),
// fitted (non-negative least squares on relative error) to 28 measured shapes - C 64/256, 20^3..80^3, 64/512 boxes, 7^3 and
// 10^3 bins, small and large boxes: tools/roialign_shapes_probe.py -> profiles/r06_roialign_bwd_shapes.txt.  Taking the
// model's pick costs 4.5 % over the better form on average (always-workspace: 17 %, worst 2.2x; always-in-place: 35 %).
// covered_voxels: R if the caller knows it (the Python wrapper measures it once per call shape, asynchronously), < 0 =
// unknown: the lower bound K * min(bins, V) is used, which biases towards the in-place form.
int inr_roi_align_3d_backward_prefers_workspace(int32_t N, int32_t C, int32_t W, int32_t L, int32_t H, int64_t K,
                                                int32_t out_w, int32_t out_l, int32_t out_h, int64_t covered_voxels) {
  if (inr_roi_align_3d_backward_workspace_bytes(N, C, W, L, H, K, out_w, out_l, out_h) <= 0) return 0;
  if (g_roi_mode == 2) return 1;            // mode 2 forces the separable kernels in their workspace form (A/B tests)
  const double V = (double)W * L * H;
  const double bins = (double)out_w * out_l * out_h;
  const double R = covered_voxels >= 0 ? (double)covered_voxels : (double)K * (bins < V ? bins : V);
  const double O = (double)K * bins * C, RC = R * C, NCV = (double)N * C * V, KB = (double)K * bins;
  const double t_in_place = 2.013e-9 * O + 6.284e-9 * RC + 6.81e-10 * NCV + 1.148e-7 * R + 0.0191;
  const double t_workspace = 1.007e-9 * O + 2.670e-9 * RC + 2.439e-9 * NCV + 1.446e-7 * KB + 1.595e-7 * R + 0.0152;
  return t_workspace < 0.9 * t_in_place ? 1 : 0;
}

int inr_roi_align_3d_backward_ws(const float* grad_out, const float* rois, const int32_t* roi_inds, int32_t N, int32_t C,
                                 int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w, int32_t out_l, int32_t out_h,
                                 float spatial_scale, float* grad_input, void* workspace, int64_t workspace_bytes,
                                 inr_stream_t s) {
  INR_REQUIRE(K >= 0 && N >= 0 && C > 0 && W > 0 && L > 0 && H > 0 && out_w > 0 && out_l > 0 && out_h > 0, "bad sizes");
  INR_REQUIRE(grad_input || N == 0, "null pointer");
  const int64_t V = (int64_t)W * L * H;
  if (N == 0) return INR_OK;
  if (K == 0) {                                    // the call overwrites grad_input: nothing pooled, nothing flows back
    const hipError_t e = hipMemsetAsync(grad_input, 0, (size_t)N * C * V * 4, as_stream(s));
    INR_REQUIRE(e == hipSuccess, "hipMemsetAsync failed");
    return INR_OK;
  }
  INR_REQUIRE(grad_out && rois && roi_inds, "null pointer");
  const int64_t need = inr_roi_align_3d_backward_workspace_bytes(N, C, W, L, H, K, out_w, out_l, out_h);
  INR_REQUIRE(need > 0, "inr_roi_align_3d_backward_ws: not available for these extents / this mode (workspace_bytes == 0: "
                        "call inr_roi_align_3d_backward)");
  INR_REQUIRE(workspace && workspace_bytes >= need, "workspace too small");
  int tmp_floats = 0;
  const int lds = sep_cl_lds_bytes(W, L, H, out_w, out_l, out_h, &tmp_floats);
  float* gt = static_cast<float*>(workspace);
  const hipError_t e = hipMemsetAsync(gt, 0, (size_t)need, as_stream(s));
  INR_REQUIRE(e == hipSuccess, "hipMemsetAsync failed");
  SepArgs A;
  A.C = C; A.W = W; A.L = L; A.H = H; A.ow = out_w; A.ol = out_l; A.oh = out_h; A.scale = spatial_scale;
  A.cpb = 16;
  A.ngroups = C / 16;
  A.K = (int)K;
  A.tmp_floats = tmp_floats;
  const unsigned grid = 8u * (unsigned)K * (unsigned)((A.ngroups + 7) / 8);
  k_roi_align3d_sep_bwd_cl<<<grid, SEP_THREADS, lds, as_stream(s)>>>(grad_out, rois, roi_inds, A, gt);
  const int rc = check_launch("roi_align_3d_backward (separable, channels-last accumulation)");
  if (rc != INR_OK) return rc;
  const dim3 tg((unsigned)((V + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
  k_channels_last_to_planes<<<tg, 256, 0, as_stream(s)>>>(gt, grad_input, V, C);
  return check_launch("roi_align_3d_backward (transposing copy)");
}

}  // extern "C"
