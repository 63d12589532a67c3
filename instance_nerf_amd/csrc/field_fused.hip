// Fused field evaluation for gfx950 (SURVEY a7 + a9 + a10 + a13): hash-grid gather, SH-4 and
// the sigma / colour / instance MLPs in ONE kernel; the 32 encoded features and all hidden
// activations live in registers and never touch HBM.
//
// MLP on the matrix cores with fp32-class accuracy at bf16 rate: v_mfma_f32_16x16x32_bf16 with a
// 3-term split  x*w ~= xh*wh + xh*wl + xl*wh  (xh/xl, wh/wl = bf16 head and remainder; the dropped
// xl*wl term is 2^-16 relative; products are exact and accumulate in fp32).  60 MFMAs of ~16 cycles
// per 16-sample tile instead of 160 fp32 MFMAs (v_mfma_f32_16x16x4_f32) of 32 cycles.  The exact-fp32
// variant is kept behind -DINR_MLP_FP32=1 for A/B runs.
// Operand maps (both shapes): A[i=lane&15][k-slot lane>>4], B[k-slot lane>>4][j=lane&15],
// D[i=4*(lane>>4)+r][j=lane&15]; the bf16 shape carries 8 k values per slot, the f32 shape one.
//
// Transposed formulation - H^T = W * X^T - with a 16-sample tile per wave:
//   * B operand = activations: lane (q = lane>>4, j = lane&15) holds input k-slot q of sample j;
//   * A operand = weights:     lane holds W[out = 16*mt + j][k-slot q], pre-packed on the host
//     into fragment order and staged once per workgroup in LDS (40 KB, ds_read_b128);
//   * D = 16 outputs x 16 samples: lane (q, j), register r holds output 4q+r of sample j.
// A dot product does not care in which order k is visited, so the k-step (nt, r) of the next
// layer is DEFINED to cover inputs {16nt + 4q + r : q = 0..3}: that is exactly register r of
// output tile nt of the previous layer.  Layer outputs therefore feed the next layer's B
// operand in place - no LDS transpose, no cross-lane traffic between layers.
// The same trick places the encoder: lane q of a sample owns levels {2q, 2q+1, 8+2q, 9+2q}
// (features 16*(s>>2) + 4q + (s&3), s = 0..7), so four lanes share one sample's 128 gathers.
//
// Compiled with -ffp-contract=off (cell selection is the oracle's decision); blending and the
// MFMA chain are explicit fmaf / MFMA.
#include <atomic>
#include <mutex>

#include "common.h"
#include "grid_common.h"

namespace inr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef INR_MLP_FP32
#define INR_MLP_FP32 0
#endif

// Ablation variants for profiling (no MLP, per-workgroup time stamps, static tile deal, one slow XCD) produce WRONG
// results.  Their code lives outside the sources build.py compiles - csrc/probe/field_probe.h, which only
// tools/build_probe.py adds (-DINR_PROBE_BUILD) - and hooks in through the INR_PROBE_* macros below, all empty here
// (round-5 verdict item 7a: a stray -D must not be able to turn the product into a silently wrong library).
#ifdef INR_PROBE_BUILD
#include "probe/field_probe.h"
#else
#if defined(INR_PROBE_MODE) || defined(INR_PROBE_STATIC) || defined(INR_PROBE_SLOW_XCD)
#error "INR_PROBE_* switches need -DINR_PROBE_BUILD (tools/build_probe.py): those variants produce wrong results"
#endif
#define INR_PROBE_PROLOGUE()
#define INR_PROBE_SCHEDULE(steal)
#define INR_PROBE_TILE()
#define INR_PROBE_AFTER_GATHER()
#define INR_PROBE_GEO_IS_OUTPUT 1
#define INR_PROBE_EPILOGUE()
#endif

// (ns-1, BASELINE north star "LDS staging of per-level feature tiles": besides the weights and the level records,
//  level 0 of the table - dense, 4 920 rows = 39 KB - was made LDS-resident for the eval kernel in round 3, its eight
//  look-ups per sample served by ds_read_b64.  Bit-identical, measured, rejected: 5.37-5.40 vs 5.20-5.25 ms per 37 M
//  samples with the select placed after all loads are out (-3.3 % L1 look-ups, +8 % VALU instructions: the look-ups it
//  removes were the cheapest - 16 samples of a tile share one or two lines of level 0), 6.73 ms with the select inside
//  the issue loop.  profiles/r03_NOTES.txt section 6, r03f / r03g PMC files.)

// packed-buffer section offsets, in floats
constexpr int kSig0 = 0;                       // 64 x 32  : 4 mt x 8 ks
constexpr int kSig1 = kSig0 + 64 * 32;         // 16 x 64  : 1 mt x 16 ks
constexpr int kCol0 = kSig1 + 16 * 64;         // 64 x 32  : 4 mt x 8 ks   (16 sh + 16 sigma-net rows)
constexpr int kCol1 = kCol0 + 64 * 32;         // 64 x 64  : 4 mt x 16 ks
constexpr int kCol2 = kCol1 + 64 * 64;         // 16 x 64  : 1 mt x 16 ks  (3 live rows)
constexpr int kNerfFloats = kCol2 + 16 * 64;   // 10240 floats = 40 KB
constexpr int kIns0 = 0;                       // 64 x 32
constexpr int kIns1 = kIns0 + 64 * 32;         // 64 x 64
constexpr int kIns2 = kIns1 + 64 * 64;         // K x 64   : K/16 mt x 16 ks

// position of fragment value (mt, ks, lane) inside a section with n_ks k-steps
__host__ __device__ inline int frag_pos(int mt, int ks, int lane, int n_ks) {
  return ((mt * (n_ks / 4) + ks / 4) * 64 + lane) * 4 + (ks & 3);
}

__device__ __forceinline__ f32x4 mfma4(const float4 a, const f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], c, 0, 0, 0);
  return c;
}
// ReLU as ONE integer max on the bit pattern (negative floats, -0 and negative NaNs are negative integers);
// fmaxf() costs two v_max_f32 here because the compiler canonicalises MFMA outputs first.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v[0] = relu1(v[0]); v[1] = relu1(v[1]); v[2] = relu1(v[2]); v[3] = relu1(v[3]);
  return v;
}

// ---- encoder, one lane's quarter of a sample -----------------------------------------------------
// Lane q owns levels {2q, 2q+1, 8+2q, 9+2q}; their constants are loaded ONCE per wave (they do
// not depend on the tile).  Indexing is branch-free: the dense index cx + cy*s + cz*s^2 and the
// hash cx ^ cy*p1 ^ cz*p2 share the two multiplies (pa, pb = strides or primes) and differ only
// in the combining operator, selected per lane with v_cndmask; (cy+1)*pa = cy*pa + pa (mod 2^32).
// Gathers are buffer loads: one 32-bit byte offset per corner against a wave-uniform descriptor
// of the whole table (49 MB < 4 GB), so a corner costs ONE address VGPR and is bounds-checked.
// Per-(q, slot) level constants live in LDS (16 records of 8 words, written once per workgroup) and are
// re-read per tile right before use - two ds_read_b128 per level - instead of occupying ~24 VGPRs
// for the whole kernel: the gather phase already holds 64 data + 12 fraction registers per lane and
// must stay <= 128 VGPRs for 4 waves/SIMD.
//   word 0 scale (f32)   1 byte offset of the level's first row   2 pa   3 pb   4 mask   5 hashed(0/1)
// A level beyond num_levels needs no predicate: all its words are 0, so its gathers read row 0 of the
// table (valid memory, finite) and its two features meet all-zero weight columns (the packer zero-fills
// inputs >= 2 * num_levels).
struct LevelRec {
  uint4 a, b;
};
constexpr int kLevelRecBytes = 16 * sizeof(LevelRec);   // 512 B after the weights

__device__ __forceinline__ void stage_level_recs(const GridDesc& G, LevelRec* recs) {
  const int t = threadIdx.x;
  if (t < 16) {
    const int q = t >> 2, li = t & 3;
    const int l = (li >> 1) * 8 + 2 * q + (li & 1);
    const bool live = l < G.num_levels;
    const int lc = live ? l : 0;
    const uint32_t m = G.mask[lc], s = G.res1[lc];
    LevelRec r;
    r.a.x = live ? __float_as_uint(G.scales[lc]) : 0u;
    r.a.y = live ? G.offsets[lc] * 8u : 0u;
    r.a.z = !live ? 0u : (m ? 2654435761u : s);
    r.a.w = !live ? 0u : (m ? 805459861u : s * s);
    r.b.x = !live ? 0u : (m ? m : 0xFFFFFFFFu);
    r.b.y = (live && m) ? 1u : 0u;
    r.b.z = 0u; r.b.w = 0u;
    recs[t] = r;
  }
}

// wave-uniform (kernel arguments only): slot li is served by hashed, present levels for every q
__device__ __forceinline__ bool slot_all_hashed(const GridDesc& G, int li) {
  bool ah = true;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int lq = (li >> 1) * 8 + 2 * qq + (li & 1);
    ah = ah && (lq < G.num_levels) && (G.mask[lq < G.num_levels ? lq : 0] != 0);
  }
  return ah;
}

typedef unsigned int u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned int))));

// Default cache policy on the gathers (raw_buffer_load aux 0).  Measured and rejected in round 2:
// nt on the fine levels 15.5 ms, sc1 9.25 ms, sc0 = default 6.07 ms per 37 M samples.
// kHalf: the table is a half-precision copy ([T,2] fp16, 4 bytes per row - upstream's `-O` / fp16 storage as an
// OPT-IN for inference, NeRFNetwork.half_table): a row is one dword, its byte offset half the fp32 one, and the two
// features are widened when they are blended (v_cvt_f32_f16 x 2 per corner).  512 instead of 1024 algorithmic bytes
// per sample; 32 instead of 16 rows per 128-byte line.
template <bool kHalf = false>
__device__ __forceinline__ u32x2 gather_row(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
  if constexpr (kHalf) {
    u32x2 r;
    r[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(byte_off >> 1), 0, 0);
    r[1] = 0u;
    return r;
  } else {
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)byte_off, 0, 0);
  }
}

// Lane-paired gather for the FINE slots (levels 8..15).  Measured on MI355X (tools/micro/gather_bench.hip): the vector
// memory path prices a gather instruction by the number of DISTINCT 128-byte lines its 64 lanes touch - any two lanes
// of the wave that hit the same line share one look-up (1.6 distinct lines per clock per CU out of the L1, 0.45 out of
// the L2, 0.10 out of the Infinity Cache) - while the same line touched by two different instructions is looked up
// twice.  The two x-neighbour corners of a cell sit in one line 15 times out of 16 (x is the fastest index in dense
// and hashed levels), and on the fine levels neighbouring samples share nothing, so there every corner is its own
// look-up.  Hence: lanes q and q^1 of a sample (one "pair") together own the four fine levels {8+4p .. 11+4p},
// p = q>>1; each fetches ITS x side (q&1) of all four - 16 loads per lane as before - so that the x-neighbour rows are
// requested by two lanes of ONE instruction and share a look-up.  Each lane blends its side's four corners per level;
// the two halves meet through v_permlane16_swap_b32 (lanes 16 apart exchange registers): 4 swaps + 4 adds per tile,
// after which lane q holds the complete features of its own two fine levels exactly where the MLP expects them.
// The coarse slots (levels 0..7) keep the one-lane-per-level scheme: there neighbouring samples already share lines.
//
// Instruction diet (the kernel is co-limited by VALU issue: 775 VALU instructions per tile before):
//  * cell = (uint)pos and fraction = v_fract_f32(pos) instead of floor / subtract / convert: pos >= 0.5, so truncation
//    IS the floor, and pos - floor(pos) is exact in binary32, so v_fract returns the same bits;
//  * weights and blending on the packed-fp32 pipe: the two features of a row arrive as a register pair and take the
//    same weight (v_pk_fma_f32 with a broadcast operand), x-neighbour weights are formed two at a time (v_pk_mul_f32);
//    every component is the same IEEE operation as before - (wx*wy)*wz, then fma in corner order - so nothing changes
//    numerically.
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Gathered {
  u32x2 c[2][8];      // coarse slots: raw rows [level][corner]
  u32x2 f[4][4];      // fine slots: this lane's x side, [level of the pair][yz corner]
  float cfx[2], cfy[2], cfz[2];
  float fwx[4], ffy[4], ffz[4];   // fine: weight of this lane's x side, y / z fractions
};

// ONE wave-uniform branch around the whole gather sequence, none inside it.  Round 1 chose between the xor-only and
// the general index maths per level: every such diamond is a control-flow merge at which the compiler's wait-count
// pass must assume that the loads of the side NOT taken are still outstanding into the registers the other side
// writes, so it put `s_waitcnt vmcnt(2)` in front of every level's block - the 32 gathers of a tile went out in five
// batches, each waiting for the previous one, instead of all at once (found in the ISA in round 2).
// Staging of the 32 gathers of a tile (measured, profiles/r02_NOTES.txt section 9): all 32 in flight at once is NOT
// the fastest - on the table feed 5.71 ms against 5.50 ms when the fine levels' loads go out only after all but 8
// of the 16 coarse loads have landed and each later fine level waits until at most 4 loads are outstanding.  The
// kernel is throughput-bound (look-ups, VALU issue), not latency-bound, and fewer lines in flight per wave leave
// more of the 32 KB L1 to the other seven waves of the CU.  (-1 = no wait.)
constexpr int kSliceLevels = 3;           // XCD-sliced frame path (k_grid_fine_slices): levels 13..15 arrive precomputed
constexpr int kWaitAfterCoarse = 8;
constexpr int kWaitBetweenFine = 4;
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N >= 0) __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

template <bool kFineHashed, bool kHalf = false, bool kPre = false>
__device__ __forceinline__ void issue_gathers_impl(const LevelRec* __restrict__ my_recs, __amdgpu_buffer_rsrc_t rsrc,
                                                   float x0, float x1, float x2, Gathered& g) {
  const int q = (threadIdx.x >> 4) & 3;
  const uint32_t side = (uint32_t)(q & 1);
  const LevelRec* pair_recs = my_recs - 4 * (q & 1);     // records of the even lane of this pair
#pragma unroll
  for (int li = 0; li < 2; ++li) {
    const uint4 ra = my_recs[li].a;
    const float s = __uint_as_float(ra.x);
    const uint32_t base = ra.y, pa = ra.z, pb = ra.w;
    const uint32_t mask = my_recs[li].b.x;
    const float px = x0 * s + 0.5f, py = x1 * s + 0.5f, pz = x2 * s + 0.5f;   // mul, add: not fused
    g.cfx[li] = __builtin_amdgcn_fractf(px); g.cfy[li] = __builtin_amdgcn_fractf(py); g.cfz[li] = __builtin_amdgcn_fractf(pz);
    const uint32_t cx = (uint32_t)px, cy = (uint32_t)py, cz = (uint32_t)pz;
    const uint32_t hy0 = cy * pa, hy1 = hy0 + pa;
    const uint32_t hz0 = cz * pb, hz1 = hz0 + pb;
    const bool h = my_recs[li].b.y != 0;                 // per lane: the four q of a coarse slot mix dense and hashed
    const uint32_t yz[4] = {h ? (hy0 ^ hz0) : (hy0 + hz0), h ? (hy1 ^ hz0) : (hy1 + hz0),
                            h ? (hy0 ^ hz1) : (hy0 + hz1), h ? (hy1 ^ hz1) : (hy1 + hz1)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t c = cx + (k & 1);
      const uint32_t idx = (h ? (c ^ yz[k >> 1]) : (c + yz[k >> 1])) & mask;
      g.c[li][k] = gather_row<kHalf>(rsrc, base + idx * 8u);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const LevelRec* rec = pair_recs + (i >> 1) * 4 + 2 + (i & 1);    // level 8 + 4p + i
    const uint4 ra = rec->a;
    const float s = __uint_as_float(ra.x);
    const uint32_t base = ra.y, pa = ra.z, pb = ra.w;
    const uint32_t mask = rec->b.x;
    const float px = x0 * s + 0.5f, py = x1 * s + 0.5f, pz = x2 * s + 0.5f;
    const float fx = __builtin_amdgcn_fractf(px);
    g.fwx[i] = side ? fx : 1.0f - fx;
    g.ffy[i] = __builtin_amdgcn_fractf(py); g.ffz[i] = __builtin_amdgcn_fractf(pz);
    const uint32_t c = (uint32_t)px + side, cy = (uint32_t)py, cz = (uint32_t)pz;
    const uint32_t hy0 = cy * pa, hy1 = hy0 + pa;
    const uint32_t hz0 = cz * pb, hz1 = hz0 + pb;
    if (i == 0) wait_vmcnt<kWaitAfterCoarse>();
    else wait_vmcnt<kWaitBetweenFine>();
    if constexpr (kPre) {
      // sliced feed: the finest kSliceLevels levels (of pair 1's 12..15) arrive precomputed; its lanes ask for row 0 of
      // the table instead (all of them the SAME line: one look-up per instruction, no divergence) and the caller
      // overwrites those features
      const uint32_t keep = ((q >> 1) && i >= 4 - kSliceLevels) ? 0u : 0xFFFFFFFFu;
      g.f[i][0] = gather_row<kHalf>(rsrc, (base + ((c ^ (hy0 ^ hz0)) & mask) * 8u) & keep);
      g.f[i][1] = gather_row<kHalf>(rsrc, (base + ((c ^ (hy1 ^ hz0)) & mask) * 8u) & keep);
      g.f[i][2] = gather_row<kHalf>(rsrc, (base + ((c ^ (hy0 ^ hz1)) & mask) * 8u) & keep);
      g.f[i][3] = gather_row<kHalf>(rsrc, (base + ((c ^ (hy1 ^ hz1)) & mask) * 8u) & keep);
    } else if constexpr (kFineHashed) {       // every fine level of both pairs is hashed: xor-only index maths
      g.f[i][0] = gather_row<kHalf>(rsrc, base + ((c ^ (hy0 ^ hz0)) & mask) * 8u);
      g.f[i][1] = gather_row<kHalf>(rsrc, base + ((c ^ (hy1 ^ hz0)) & mask) * 8u);
      g.f[i][2] = gather_row<kHalf>(rsrc, base + ((c ^ (hy0 ^ hz1)) & mask) * 8u);
      g.f[i][3] = gather_row<kHalf>(rsrc, base + ((c ^ (hy1 ^ hz1)) & mask) * 8u);
    } else {
      const bool h = rec->b.y != 0;
      const uint32_t yz[4] = {h ? (hy0 ^ hz0) : (hy0 + hz0), h ? (hy1 ^ hz0) : (hy1 + hz0),
                              h ? (hy0 ^ hz1) : (hy0 + hz1), h ? (hy1 ^ hz1) : (hy1 + hz1)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t idx = (h ? (c ^ yz[k]) : (c + yz[k])) & mask;
        g.f[i][k] = gather_row<kHalf>(rsrc, base + idx * 8u);
      }
    }
  }
}

template <bool kHalf = false, bool kPre = false>
__device__ __forceinline__ void issue_gathers(const LevelRec* __restrict__ my_recs, const bool (&all_hashed)[4],
                                              __amdgpu_buffer_rsrc_t rsrc, float x0, float x1, float x2, Gathered& g) {
  if constexpr (kPre) issue_gathers_impl<true, kHalf, true>(my_recs, rsrc, x0, x1, x2, g);   // (the host checks: fine levels hashed)
  else if (all_hashed[2] && all_hashed[3]) issue_gathers_impl<true, kHalf>(my_recs, rsrc, x0, x1, x2, g);
  else issue_gathers_impl<false, kHalf>(my_recs, rsrc, x0, x1, x2, g);
}

template <bool kHalf = false>
__device__ __forceinline__ f32x2 row2(const u32x2 v) {
  const unsigned bx = v[0], by = v[1];     // (scalars first: see the note on __builtin_bit_cast below)
  if constexpr (kHalf) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_bit_cast(h2, bx);
    return f32x2{(float)h[0], (float)h[1]};
  } else {
    return f32x2{__uint_as_float(bx), __uint_as_float(by)};
  }
}

// trilinear blend: weight = (wx*wy)*wz, accumulated with fma - coarse levels in corner order 0..7, fine levels as
// (this side's corners in yz order) and then x side 0 + x side 1.
// out[s], s = 0..7 <-> feature 16*(s>>2) + 4q + (s&3).
template <bool kHalf = false>
__device__ __forceinline__ void blend(const Gathered& g, f32x4& lo, f32x4& hi) {
#pragma unroll
  for (int li = 0; li < 2; ++li) {
    const f32x2 wx = {1.0f - g.cfx[li], g.cfx[li]};
    const float wy0 = 1.0f - g.cfy[li], wz0 = 1.0f - g.cfz[li];
    const f32x2 a = wx * wy0, b = wx * g.cfy[li];                     // (wx*wy) for y side 0 / 1, both x sides
    const f32x2 w[4] = {a * wz0, b * wz0, a * g.cfz[li], b * g.cfz[li]};   // corners (0,1) (2,3) (4,5) (6,7)
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x2 wk = (k & 1) ? w[k >> 1].yy : w[k >> 1].xx;
      acc = __builtin_elementwise_fma(wk, row2<kHalf>(g.c[li][k]), acc);
    }
    lo[2 * li] = acc.x;
    lo[2 * li + 1] = acc.y;
  }
  f32x2 part[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 wy = {1.0f - g.ffy[i], g.ffy[i]};
    const f32x2 xy = wy * g.fwx[i];                                    // (wx*wy): y side 0 / 1 (product commutes)
    const f32x2 w0 = xy * (1.0f - g.ffz[i]), w1 = xy * g.ffz[i];       // corners (y0z0, y1z0), (y0z1, y1z1)
    f32x2 acc = {0.f, 0.f};
    acc = __builtin_elementwise_fma(w0.xx, row2<kHalf>(g.f[i][0]), acc);
    acc = __builtin_elementwise_fma(w0.yy, row2<kHalf>(g.f[i][1]), acc);
    acc = __builtin_elementwise_fma(w1.xx, row2<kHalf>(g.f[i][2]), acc);
    acc = __builtin_elementwise_fma(w1.yy, row2<kHalf>(g.f[i][3]), acc);
    part[i] = acc;
  }
  // v_permlane16_swap_b32 vdst, src: the odd 16-lane rows of vdst trade places with the even rows of src.  With
  // vdst = the partial of an even-lane level and src = the partial of an odd-lane level, afterwards BOTH registers of
  // every lane belong to that lane's own level: one is its own half, the other the partner's.
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int lv = t >> 1, ft = t & 1;
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(part[lv][ft]), __float_as_uint(part[2 + lv][ft]), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    hi[t] = __uint_as_float(r0) + __uint_as_float(r1);      // = x side 0 + x side 1 in both lanes
  }
}

// (Round 1 had tried an x-split with EIGHT lanes per sample - twice the index maths per sample - and lost 11 % in
// the VALU-bound regime.  The pairing above keeps four lanes per sample and 32 loads per lane; only the fine levels'
// cell location is duplicated.)

// One MLP layer: out[mt] (N_MT output tiles) = W * in, K = 16 * N_G inputs.  in[g] is the B
// operand of k-steps 4g..4g+3.  The N_MT accumulator chains are interleaved (independent
// MFMAs back to back: the 16x16x4 f32 MFMA issues every 32 cycles but has a 40-cycle
// dependent latency); a single-tile layer splits K over two accumulators instead.
template <int N_MT, int N_G>
__device__ __forceinline__ void layer(const float4* __restrict__ w, int lane, const f32x4* in, f32x4* out) {
  if constexpr (N_MT == 1) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < N_G; g += 2) {
      __builtin_amdgcn_sched_barrier(0);
      const float4 w0 = w[g * 64 + lane], w1 = w[(g + 1) * 64 + lane];
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, in[g][0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, in[g + 1][0], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, in[g][1], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, in[g + 1][1], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, in[g][2], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, in[g + 1][2], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, in[g][3], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, in[g + 1][3], a1, 0, 0, 0);
    }
    out[0] = a0 + a1;
  } else {
    f32x4 acc[N_MT];
#pragma unroll
    for (int mt = 0; mt < N_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < N_G; ++g) {
      __builtin_amdgcn_sched_barrier(0);   // keep weight reads next to their MFMAs (register pressure)
      float4 wv[N_MT];
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) wv[mt] = w[(mt * N_G + g) * 64 + lane];
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mt].x, in[g][0], acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mt].y, in[g][1], acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mt].z, in[g][2], acc[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mt].w, in[g][3], acc[mt], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < N_MT; ++mt) out[mt] = acc[mt];
  }
}

// ---- split-bf16 MLP ----------------------------------------------------------------------------------
// Packed weights (host, inr_*_pack_weights): per section [mt][step][hi|lo][lane][8 bf16]; the 8 values
// of lane (q, i) in step s are W[16mt + i][kidx(s, q, e)], e = 0..7, with
//   hidden / encoder inputs: kidx = 16*(2s + (e>>2)) + 4q + (e&3)   (= D register e&3 of output tile 2s + (e>>2))
// so the B operand of step s is registers {h[2s][0..3], h[2s+1][0..3]} of the previous layer, in place.
struct SplitB {
  uint4 hi, lo;   // 8 bf16 each
};

// bf16 head by truncation (v_perm_b32 packs the two high halves), remainder exactly in fp32, then truncated.
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float l0 = x0 - __uint_as_float(u0 & 0xFFFF0000u);
  const float l1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
  lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}
__device__ __forceinline__ SplitB split8(const f32x4 a, const f32x4 b) {
  SplitB r;
  split2(a[0], a[1], r.hi.x, r.lo.x);
  split2(a[2], a[3], r.hi.y, r.lo.y);
  split2(b[0], b[1], r.hi.z, r.lo.z);
  split2(b[2], b[3], r.hi.w, r.lo.w);
  return r;
}
__device__ __forceinline__ f32x4 mfma_bf(const uint4 a, const uint4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// out[mt] = W * in over N_S steps of 32 inputs.  w: section base (uint4 = 8 bf16), fragment (mt, s, hl) at
// ((mt*N_S + s)*2 + hl)*64 + lane.  Independent accumulator chains are interleaved.
template <int N_MT, int N_S>
__device__ __forceinline__ void layer_bf(const uint4* __restrict__ w, int lane, const SplitB* in, f32x4* out) {
  if constexpr (N_MT == 1) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
    for (int st = 0; st < N_S; ++st) {
      __builtin_amdgcn_sched_barrier(0);
      const uint4 wh = w[(st * 2 + 0) * 64 + lane], wlo = w[(st * 2 + 1) * 64 + lane];
      a0 = mfma_bf(wh, in[st].hi, a0);
      a1 = mfma_bf(wlo, in[st].hi, a1);
      a2 = mfma_bf(wh, in[st].lo, a2);
    }
    out[0] = a0 + (a1 + a2);
  } else {
    f32x4 acc[N_MT];
#pragma unroll
    for (int mt = 0; mt < N_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < N_S; ++st) {
      __builtin_amdgcn_sched_barrier(0);   // keep weight reads next to their MFMAs (register pressure)
      uint4 wh[N_MT], wlo[N_MT];
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) {
        wh[mt] = w[((mt * N_S + st) * 2 + 0) * 64 + lane];
        wlo[mt] = w[((mt * N_S + st) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = mfma_bf(wh[mt], in[st].hi, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = mfma_bf(wlo[mt], in[st].hi, acc[mt]);
#pragma unroll
      for (int mt = 0; mt < N_MT; ++mt) acc[mt] = mfma_bf(wh[mt], in[st].lo, acc[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < N_MT; ++mt) out[mt] = acc[mt];
  }
}

// Opt-in single-pass variant (kFast; upstream's `-O` numerics: fp16 MLP operands, fp32 accumulation): weights packed as
// fp16 in the head slots of the same fragment layout (pack_section_f16), activations rounded to fp16
// (v_cvt_pk_f16_f32, round to nearest even), ONE v_mfma_f32_16x16x32_f16 per (tile, step) instead of three bf16 passes,
// one conversion per activation pair instead of the head / remainder split.  2^-12 relative per operand: close to, but
// NOT, the fp32-class default.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
template <int N_MT, int N_S>
__device__ __forceinline__ void layer_h1(const uint4* __restrict__ w, int lane, const uint4* in, f32x4* out) {
  f32x4 acc[N_MT];
#pragma unroll
  for (int mt = 0; mt < N_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < N_S; ++st) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < N_MT; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, w[((mt * N_S + st) * 2 + 0) * 64 + lane]),
                                                       __builtin_bit_cast(f16x8_t, in[st]), acc[mt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < N_MT; ++mt) out[mt] = acc[mt];
}

// Layer entry points used by the kernels: `in` is N_G groups of 4 registers (N_G = K/16).
template <int N_MT, int N_G, bool kFast = false>
__device__ __forceinline__ void mlp_layer(const float4* __restrict__ wsec, int lane, const f32x4* in, f32x4* out) {
#if INR_MLP_FP32
  layer<N_MT, N_G>(wsec, lane, in, out);
#else
  if constexpr (kFast) {
    uint4 b[N_G / 2];
#pragma unroll
    for (int st = 0; st < N_G / 2; ++st)
      b[st] = uint4{pack_f16x2(in[2 * st][0], in[2 * st][1]), pack_f16x2(in[2 * st][2], in[2 * st][3]),
                    pack_f16x2(in[2 * st + 1][0], in[2 * st + 1][1]), pack_f16x2(in[2 * st + 1][2], in[2 * st + 1][3])};
    layer_h1<N_MT, N_G / 2>(reinterpret_cast<const uint4*>(wsec), lane, b, out);
  } else {
    SplitB b[N_G / 2];
#pragma unroll
    for (int st = 0; st < N_G / 2; ++st) b[st] = split8(in[2 * st], in[2 * st + 1]);
    layer_bf<N_MT, N_G / 2>(reinterpret_cast<const uint4*>(wsec), lane, b, out);
  }
#endif
}

__device__ __forceinline__ float select4(int q, float a, float b, float c, float d) {
  return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}

struct TileIn {
  float x0, x1, x2;   // x01
  float d0, d1, d2;
  bool oob;           // x01 left [0,1]: features are zero (upstream's flag_oob); x01 is then replaced by 0
};

// x01 = (x + bound) / (2 bound): a true IEEE division, as in the oracle.  (Replacing it by a multiplication
// with the exact reciprocal when 2*bound is a power of two - selected by a wave-uniform flag - measured 7 %
// SLOWER: the extra control flow in front of the gathers costs more than the ~30 instructions it saves.)
__device__ __forceinline__ float to_x01(float x, float bound, float rb, float) { return (x + bound) / rb; }

template <bool kDir>
__device__ __forceinline__ void load_tile_in(const float* __restrict__ x, const float* __restrict__ d, int64_t m,
                                             float bound, float rb, float rb_inv, TileIn& t) {
  t.x0 = to_x01(x[m * 3 + 0], bound, rb, rb_inv);
  t.x1 = to_x01(x[m * 3 + 1], bound, rb, rb_inv);
  t.x2 = to_x01(x[m * 3 + 2], bound, rb, rb_inv);
  t.oob = oob01(t.x0, t.x1, t.x2);
  if (t.oob) t.x0 = t.x1 = t.x2 = 0.0f;
  if constexpr (kDir) {
    t.d0 = d[m * 3]; t.d1 = d[m * 3 + 1]; t.d2 = d[m * 3 + 2];
  }
}

// Latency hiding is by wave-level parallelism, not by a per-wave software pipeline: a tile's 32 gathers
// per lane are all issued back to back (addresses first, then one wait) and while one wave of a SIMD
// runs its 60 MFMAs the other has its table reads in flight.  A per-wave double-buffered variant
// (next tile's gathers issued before this tile's MLP, 168 VGPRs) was built and measured twice: it is
// not faster (6.95 vs 6.81 ms at 8 waves/CU) because it enlarges the per-CU working set exactly like a
// higher occupancy does - see grid_for() - so it was removed.
// Workgroup b runs on XCD b % 8 (observed dispatch order, confirmed with HW_REG_XCC_ID; used for speed only - any
// placement gives the same results).  The sample stream is ray-major and rays are image-ordered, so a contiguous run of
// tiles is a compact region of space: the waves of an XCD sweep such runs together, which keeps the XCD's private 4 MB L2
// on one part of the hash table's working set instead of all of it.
// XCD-aware persistent schedule.  Every XCD gets the same number of tiles, but NOT one contiguous eighth of the stream
// (rounds 1 and most of 2 did that): equal sample counts are not equal work - a tile costs what its region of the
// scene costs in cache misses, and the frame is ray-ordered top to bottom - so the XCD with the expensive eighth set
// the kernel's time while the others idled (field kernel 5.53 -> 5.18 ms for 37 M samples once the eighths were
// interleaved, profiles/r02_NOTES.txt section 28).  The stream is cut into chunks of 2^kXcdChunkLog2 tiles dealt round
// robin to the XCDs (a chunk is still a compact run of patches: the XCD's private L2 keeps its locality); what is left
// after the last complete round of eight chunks is split into eighths as before, so the counts stay equal.  Launches
// with fewer than four rounds (training batches: random rays, no regional structure) keep the contiguous split.
constexpr int kXcdChunkLog2 = 10;
#ifndef INR_STEAL_SHIFT
#define INR_STEAL_SHIFT 1
#endif
constexpr int kStealRoundsShift = INR_STEAL_SHIFT;     // hybrid schedule: the last rounds >> shift rounds are drawn (TileWalk)
struct TileSched {
  int64_t lo, hi;      // hi: one past the last tile this wave may take
  int64_t first;       // first tile (contiguous split) or first XCD-local index (interleaved split)
  int64_t stride;      // distance between consecutive tiles / local indices of this wave
  int64_t main_local = 0, tail_base = 0;   // interleaved split: local indices below main_local map into the chunked part
  int64_t n_main = 0, tail = 0;            // tiles in complete rounds of eight chunks / behind them (interleaved split)
  int64_t static_local = 0;                // hybrid: local indices from here on are handed out by cursors (TileWalk)
  uint32_t xcd = 0;
  bool interleaved = false, hybrid = false;
  // local index u of XCD y -> tile; one past XCD y's last local index
  __device__ __forceinline__ int64_t tile_of(int y, int64_t u) const {
    if (u < main_local) return ((((u >> kXcdChunkLog2) << 3) + y) << kXcdChunkLog2) + (u & ((1 << kXcdChunkLog2) - 1));
    return n_main + tail * y / 8 + (u - main_local);
  }
  __device__ __forceinline__ int64_t local_end(int y) const { return main_local + tail * (y + 1) / 8 - tail * y / 8; }
  __device__ __forceinline__ int64_t tile(int64_t i) const {
    if (interleaved) {
      const int64_t u = first + i * stride;
      if (u < main_local)
        return ((((u >> kXcdChunkLog2) << 3) + xcd) << kXcdChunkLog2) + (u & ((1 << kXcdChunkLog2) - 1));
      return tail_base + (u - main_local);
    }
    return first + i * stride;
  }
};
__device__ __forceinline__ TileSched make_sched(int64_t n_tiles, int waves_per_block, bool steal = false) {
  const int nb = gridDim.x, b = blockIdx.x;
  const int w = threadIdx.x >> 6;
  TileSched s;
  if (nb % 8 == 0) {
    const int xcd = b & 7, local = b >> 3, per = nb >> 3;
    const int64_t rounds = n_tiles >> (kXcdChunkLog2 + 3);
    if (rounds >= 4) {
      const int64_t n_main = rounds << (kXcdChunkLog2 + 3), tail = n_tiles - n_main;
      s.interleaved = true;
      s.xcd = (uint32_t)xcd;
      s.main_local = rounds << kXcdChunkLog2;
      s.n_main = n_main;
      s.tail = tail;
      s.hybrid = steal;
      s.static_local = steal ? (rounds - max((int64_t)1, rounds >> kStealRoundsShift)) << kXcdChunkLog2 : s.main_local;
      s.tail_base = n_main + tail * xcd / 8;
      s.lo = 0;
      s.hi = n_main + tail * (xcd + 1) / 8;
      s.first = (int64_t)local * waves_per_block + w;
      s.stride = (int64_t)per * waves_per_block;
      return s;
    }
    s.lo = n_tiles * xcd / 8;
    s.hi = n_tiles * (xcd + 1) / 8;
    s.first = s.lo + (int64_t)local * waves_per_block + w;
    s.stride = (int64_t)per * waves_per_block;
  } else {
    s.lo = 0;
    s.hi = n_tiles;
    s.first = (int64_t)b * waves_per_block + w;
    s.stride = (int64_t)nb * waves_per_block;
  }
  return s;
}

constexpr int kFieldThreads = 512;     // waves of a workgroup share one 40 KB weight image in LDS
constexpr int kGroupChunkLog2 = 6;     // dynamic group schedules: 2^k consecutive 16-ray groups per cursor and round
constexpr int kGroupDraw = 1;          // groups an owner takes per draw (must divide the chunk)
constexpr int kGroupPools = 4;                            // cursors per XCD, wave-owned groups (see GroupDraw)
// DYNAMIC schedule over 16-ray groups (kernels in which a wave or a workgroup owns a group for all its steps).  A group
// costs what its rays' sample counts and termination points make it cost (the ceiling rows of a frame: a dozen steps;
// the middle rows: seventy; an opaque group: two), so a fixed deal of ~20 groups per wave leaves the launch waiting for
// its unluckiest wave (round 2 measured 7.2 - 9.4 ms for the same work under four static deals).  Instead the owner of
// a group draws the next one from a cursor (zeroed by the caller): chunks of 2^kGroupChunkLog2 consecutive groups are
// dealt round robin to the XCDs - neighbouring patches still share an L2, every XCD sees every region of the frame -
// and inside an XCD groups go to whoever is free.  Same-address atomics are applied one after the other (~60 ns each):
// with one cursor per XCD an opaque frame of the wave-owned kernel - two steps per group, 2048 waves drawing - spent a
// quarter of its time queueing for it (1.22 vs 0.98 ms), so that kernel uses kGroupPools cursors per XCD, one per
// quarter of its workgroups (one chunk at a time each: the same window of the frame); the workgroup-owned kernel draws
// an eighth as often and is faster with ONE window per XCD (instance render 11.4 ms; four pools 12.0; the static deal
// 11.6).  Measured and not kept: 4 groups per draw (9.8 instead of 6.9 ms: the waves of an XCD then span four times the
// window and the L2 hit rate collapses), chunks of 16 / 256 / 1024 groups (6.9-7.0 / 7.1-7.5 / 7.2-8.0 ms), super-tiled
// pixel order (within noise).  Workgroup b runs on XCD b % 8 (observed; speed only - any placement gives the same results).
struct GroupDraw {
  unsigned long long* cursor;
  int owner, n_owner;
  __device__ __forceinline__ void init(unsigned long long* cursors, int pools) {
    const int n_xcd = (gridDim.x % 8 == 0) ? 8 : 1;
    const int n_pool = (gridDim.x % (8 * pools) == 0) ? pools : 1;
    const int xcd = (n_xcd == 8) ? (int)(blockIdx.x & 7) : 0;
    const int pool = (n_pool > 1) ? (int)((blockIdx.x >> 3) % pools) : 0;
    owner = pool * n_xcd + xcd;
    n_owner = n_pool * n_xcd;
    cursor = cursors + owner;
  }
  __device__ __forceinline__ int64_t group_of(uint32_t u) const {
    const int64_t chunk = (int64_t)(u >> kGroupChunkLog2) * n_owner + owner;
    return (chunk << kGroupChunkLog2) + (u & ((1u << kGroupChunkLog2) - 1u));
  }
  // called by ONE lane: the first of the kGroupDraw consecutive groups it now owns
  __device__ __forceinline__ int64_t draw() const { return group_of((uint32_t)atomicAdd(cursor, (unsigned long long)kGroupDraw)); }
};
// HYBRID schedule of the big field launches (frames): equal shares per XCD are equal WORK only while the eight XCDs
// run at the same speed, and on some boxes they do not - the same build, the same frame: 4.6 ms on one call, 5.1 ms on
// another, the trained scene 11.6 against 15.0 ms, while the kernels that draw their work (k_nerf_render,
// k_instance_render) and the ones that are not bound by the memory path (-O numerics) read the same on both
// (profiles/r03_NOTES.txt 21): with a static deal the launch waits for its slowest XCD.  So the second half of the
// rounds (and the ragged tail) is not dealt but DRAWN: every XCD's part of it sits behind kStealPools cursors
// (blocks of 64 local tiles dealt round robin to the pools, kStealDraw tiles per draw, the next draw in flight while
// the current tiles are evaluated), a wave draws from the cursor of its own XCD and pool, and when that runs dry it
// looks at all cursors at once (one 256-byte load) and moves to the first live one - its XCD's other pools first, then
// the next XCD's.  A fast XCD finishes its own part early and takes over what a slow one has left; with equal XCDs
// the drawn part is the same tiles in the same windows as before.  Cursors are zeroed by the launcher; nullptr (all
// other launches) is the static schedule.  Any order gives the same results: a tile's outputs depend on nothing else.
// Measured (view 0, 37 M samples, tools/field_probe.py on the profiling builds of tools/build_probe.py): static deal
// 5.12 ms, a quarter / half / all of the rounds drawn 4.88 / 4.85 / 4.91 ms; with one XCD made ~64 % slower
// (s_sleep per tile) 8.51 / 6.35 / 5.17 / 5.23 ms - half it is (kStealRoundsShift = 1).
constexpr int kStealPools = 4, kStealOwners = 8 * kStealPools, kStealDraw = 4, kStealBlockLog2 = 6;
struct TileWalk {
  TileSched s;
  unsigned long long* cur;
  int lane, own, owner, sub;
  int64_t it, k_base;
  unsigned long long raw_next;           // the draw in flight (lane 0), not looked at until it is needed
  bool dyn;
  __device__ __forceinline__ void init(const TileSched& sched, unsigned long long* cursors) {
    s = sched; cur = cursors; lane = threadIdx.x & 63; it = 0; dyn = false; sub = kStealDraw; k_base = 0; raw_next = 0;
    own = owner = (int)(((blockIdx.x >> 3) % kStealPools) * 8 + (blockIdx.x & 7));
    if (!cursors) s.hybrid = false;
    if (!s.hybrid) s.static_local = s.main_local;
  }
  // owner o = pool * 8 + xcd: number of cursor positions that map to local indices below the XCD's end (a multiple of
  // the block), and the local index of position k
  __device__ __forceinline__ int64_t pool_len(int o) const {
    const int64_t nb = (s.local_end(o & 7) - s.static_local + (1 << kStealBlockLog2) - 1) >> kStealBlockLog2;
    const int p = o >> 3;
    return nb > p ? ((nb - p + kStealPools - 1) / kStealPools) << kStealBlockLog2 : 0;
  }
  __device__ __forceinline__ int64_t local_of(int o, int64_t k) const {
    return s.static_local + ((((k >> kStealBlockLog2) * kStealPools) + (o >> 3)) << kStealBlockLog2) +
           (k & ((1 << kStealBlockLog2) - 1));
  }
  __device__ __forceinline__ unsigned long long issue(int o) const {
    unsigned long long v = 0;
    if (lane == 0) v = atomicAdd(cur + o, (unsigned long long)kStealDraw);
    return v;
  }
  __device__ __forceinline__ static int64_t uniform(unsigned long long v) {
    return (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)v));
  }
  // first live owner in this wave's order (own XCD's pools, then the following XCDs'), -1: every cursor has run dry
  __device__ __forceinline__ int scan() const {
    bool live = false;
    if (lane < kStealOwners) live = (int64_t)__hip_atomic_load(cur + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < pool_len(lane);
    const uint32_t mask = (uint32_t)__ballot(live);
    for (int i = 0; i < kStealOwners; ++i) {
      const int o = (((own >> 3) + i) % kStealPools) * 8 + (((own & 7) + i / kStealPools) & 7);
      if ((mask >> o) & 1u) return o;
    }
    return -1;
  }
  // the wave's next tile, -1 when there is none
  __device__ __forceinline__ int64_t next() {
    if (!dyn) {
      if (!s.hybrid) {
        const int64_t t = s.tile(it++);
        return t < s.hi ? t : -1;
      }
      const int64_t u = s.first + (it++) * s.stride;
      if (u < s.static_local) return s.tile_of((int)s.xcd, u);
      dyn = true;
      raw_next = issue(owner);
    }
    for (;;) {
      if (sub < kStealDraw) {
        const int64_t u = local_of(owner, k_base + sub);
        ++sub;
        if (u < s.local_end(owner & 7)) return s.tile_of(owner & 7, u);
        continue;                                          // behind the ragged end of the XCD's part
      }
      int64_t k = uniform(raw_next);
      while (k >= pool_len(owner)) {
        const int o = scan();
        if (o < 0) return -1;
        owner = o;
        k = uniform(issue(owner));
      }
      k_base = k;
      sub = 0;
      raw_next = issue(owner);
    }
  }
};
constexpr int kFieldMinWaves = 2;      // __launch_bounds__ second argument: <= 128 VGPRs

// kTable: fused-frame fast path - x is already normalised to [0,1] by the march writer and the direction
// encoding comes from a per-ray SH table (shq[ray][q] = this lane's four components) via a per-sample ray id:
// ~100 VALU instructions per tile less than dividing and evaluating 16 polynomials in every lane.
__device__ __forceinline__ f32x4 load4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return f32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ f32x4 mask4(f32x4 g, f32x4 h) {
  return f32x4{h[0] > 0.f ? g[0] : 0.f, h[1] > 0.f ? g[1] : 0.f, h[2] > 0.f ? g[2] : 0.f, h[3] > 0.f ? g[3] : 0.f};
}

// kSave (training) 1: activations the backward needs, row-major - enc [M,32], h1 [M,64], so [M,16] (raw sigma-net
// output: density logit + 15 geo features), cin [M,32] (colour-net input in weight-column order: 16 SH, 15 geo, 0),
// c1, c2 [M,64]; 2: enc only (k_nerf_head_bwd recomputes the rest).
struct NerfSave {
  float *enc, *h1, *so, *cin, *c1, *c2;
};

// kPre (table feed only; round 5, sliced frame path): the three FINEST levels (13..15, fine slots of lanes q = 2, 3) were
// evaluated by k_grid_fine_slices; `d` (unused by the table feed) then points to their features, float2 [3][m_pad],
// m_pad = M rounded up to 32; lanes q = 2, 3 read theirs from there and gather level 12 themselves.
template <bool kColor, bool kTable = false, int kSave = 0, bool kHalf = false, bool kFast = false, bool kPre = false>
__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_nerf_fwd(const float* __restrict__ x, const float* __restrict__ d,
                                                               int64_t M, const int32_t* __restrict__ n_dev, float bound,
                                                               const float2* __restrict__ emb, uint32_t emb_bytes, GridDesc G,
                                                               const float4* __restrict__ packed, float density_scale,
                                                               float* __restrict__ sigma, float* __restrict__ rgb,
                                                               float* __restrict__ geo, const int32_t* __restrict__ ray_ids,
                                                               const float4* __restrict__ shq, NerfSave sv,
                                                               unsigned long long* __restrict__ steal) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  INR_PROBE_PROLOGUE();
  // the field kernel's waves outrank whatever shares the CU with them (FramePipeline: the next view's marchers run
  // at priority 0 and take the issue slots this kernel leaves free); alone on the chip it changes nothing
  __builtin_amdgcn_s_setprio(3);
  constexpr int kStage = (kColor ? kNerfFloats : kCol0) / 4;
  for (int i = threadIdx.x; i < kStage; i += kFieldThreads) wl[i] = packed[i];
  LevelRec* recs = reinterpret_cast<LevelRec*>(wl + kStage);
  stage_level_recs(G, recs);
  __syncthreads();

  constexpr int kWaves = kFieldThreads / 64;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  int64_t n = M;
  if (n_dev) n = min((int64_t)*n_dev, M);
  const int64_t n_tiles = (n + 15) >> 4;
  const float rb = 2.0f * bound;
  const float rb_inv = 0.0f;
  INR_PROBE_SCHEDULE(steal);
  TileWalk walk;
  walk.init(make_sched(n_tiles, kWaves, steal != nullptr), steal);

  const bool all_hashed[4] = {slot_all_hashed(G, 0), slot_all_hashed(G, 1), slot_all_hashed(G, 2), slot_all_hashed(G, 3)};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);

  // (Requesting the NEXT tile's coordinates and ray id one tile ahead - so that neither the gather addresses nor the
  //  dependent direction-table load wait for a fresh trip through the memory system - was measured twice, in round 1
  //  on the plain feed and in round 2 on this table feed: 0 % and -2 %; removed.)
  for (int64_t tile = walk.next(); tile >= 0; tile = walk.next()) {
    INR_PROBE_TILE();
    const int64_t m = tile * 16 + j;
    const bool valid = m < n;
    TileIn me;
    f32x4 sh_in = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kTable) {
      const int64_t mm = valid ? m : n - 1;
      me.x0 = x[mm * 3 + 0]; me.x1 = x[mm * 3 + 1]; me.x2 = x[mm * 3 + 2];
      const float4 t4 = shq[(int64_t)ray_ids[mm] * 4 + q];
      sh_in = f32x4{t4.x, t4.y, t4.z, t4.w};
    } else {
      load_tile_in<kColor>(x, d, valid ? m : n - 1, bound, rb, rb_inv, me);
    }

    f32x4 enc[2];
    {
      Gathered g;
      uint32_t rec_off = (uint32_t)q * 4u * (uint32_t)sizeof(LevelRec);
      asm volatile("" : "+v"(rec_off));    // opaque per tile: keeps the records in LDS, not hoisted into VGPRs
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f fine_a = {0.f, 0.f}, fine_b = {0.f, 0.f};
      if constexpr (kPre) {                // lane q = 2: levels 12 (own gathers), 13 (slice 0); q = 3: 14, 15 (slices 1, 2)
        static_assert(!kPre || kTable, "the sliced feed is a variant of the table feed");
        static_assert(kSliceLevels == 3, "lane map below");
        const v2f* pre = reinterpret_cast<const v2f*>(d);
        const int64_t m_pad = (M + 31) & ~(int64_t)31, mm = valid ? m : n - 1;
        const int sl = (q & 1) ? 1 : 0;    // (lanes q = 0, 1 load the same rows: their values are not used)
        fine_a = __builtin_nontemporal_load(pre + (int64_t)sl * m_pad + mm);
        fine_b = __builtin_nontemporal_load(pre + (int64_t)(sl + 1) * m_pad + mm);
      }
      issue_gathers<kHalf, kPre>(reinterpret_cast<const LevelRec*>(reinterpret_cast<const char*>(recs) + rec_off), all_hashed,
                                 rsrc, me.x0, me.x1, me.x2, g);
      __builtin_amdgcn_sched_barrier(0);   // all 32 gathers in flight before the first blend waits
      blend<kHalf>(g, enc[0], enc[1]);
      if constexpr (kPre) {
        if (q == 2) enc[1] = f32x4{enc[1][0], enc[1][1], fine_a.x, fine_a.y};          // level 12 gathered here, 13 precomputed
        else if (q == 3) enc[1] = f32x4{fine_a.x, fine_a.y, fine_b.x, fine_b.y};       // 14, 15
      }
    }
    if constexpr (!kTable) {               // the table feed comes from the marcher, which clamps to the volume
      if (me.oob) enc[0] = enc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    INR_PROBE_AFTER_GATHER();

    f32x4 h1[4];
    mlp_layer<4, 2, kFast>(wl + kSig0 / 4, lane, enc, h1);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
    f32x4 h2[1];
    mlp_layer<1, 4, kFast>(wl + kSig1 / 4, lane, h1, h2);      // row 0 = raw density, rows 1..15 = geo features

    if (valid) {
      if (q == 0) sigma[m] = __expf(h2[0][0]) * density_scale;
      if (INR_PROBE_GEO_IS_OUTPUT && geo) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * q + r;
          if (row >= 1) geo[m * 15 + row - 1] = h2[0][r];
        }
      }
    }

    if constexpr (kColor) {
      f32x4 cin[2];
      if constexpr (kTable) {
        cin[0] = sh_in;
      } else {
        float sh[16];
        sh4(me.d0, me.d1, me.d2, sh);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cin[0][ks] = select4(q, sh[4 * ks], sh[4 * ks + 1], sh[4 * ks + 2], sh[4 * ks + 3]);
      }
      cin[1] = h2[0];                                 // k-slot (q, r) = sigma-net row 4q+r (row 0 has zero weight)
      f32x4 c1[4], c2[4], o[1];
      mlp_layer<4, 2, kFast>(wl + kCol0 / 4, lane, cin, c1);
#pragma unroll
      for (int t = 0; t < 4; ++t) c1[t] = relu4(c1[t]);
      mlp_layer<4, 4, kFast>(wl + kCol1 / 4, lane, c1, c2);
#pragma unroll
      for (int t = 0; t < 4; ++t) c2[t] = relu4(c2[t]);
      mlp_layer<1, 4, kFast>(wl + kCol2 / 4, lane, c2, o);
      if constexpr (kSave == 2) {          // k_nerf_head_bwd recomputes everything else from the encoder output
        if (valid) {
#pragma unroll
          for (int t = 0; t < 2; ++t) store4(sv.enc + m * 32 + 16 * t + 4 * q, enc[t]);
        }
      }
      if constexpr (kSave == 1) {
        if (valid) {
#pragma unroll
          for (int t = 0; t < 2; ++t) store4(sv.enc + m * 32 + 16 * t + 4 * q, enc[t]);
          store4(sv.so + m * 16 + 4 * q, h2[0]);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            store4(sv.h1 + m * 64 + 16 * t + 4 * q, h1[t]);
            store4(sv.c1 + m * 64 + 16 * t + 4 * q, c1[t]);
            store4(sv.c2 + m * 64 + 16 * t + 4 * q, c2[t]);
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) sv.cin[m * 32 + 4 * ks + q] = cin[0][ks];      // SH component 4ks+q
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * q + r;                                                   // sigma-net output row
            sv.cin[m * 32 + (row >= 1 ? 15 + row : 31)] = row >= 1 ? h2[0][r] : 0.f;       // geo row-1 | pad column
          }
        }
      }
      if (valid && q == 0) {
        rgb[m * 3 + 0] = __frcp_rn(1.0f + __expf(-o[0][0]));     // v_exp_f32 / v_rcp_f32: ~1e-7 relative
        rgb[m * 3 + 1] = __frcp_rn(1.0f + __expf(-o[0][1]));
        rgb[m * 3 + 2] = __frcp_rn(1.0f + __expf(-o[0][2]));
      }
    }
  }
  INR_PROBE_EPILOGUE();
}

// ---- sliced frame path: the finest levels level by level (round 5) ---------------------------------------------------
// A hashed level is 2^19 rows x 8 B = 4 MiB - exactly one XCD's L2.  Where the finest levels are finer than the spacing
// of the frame's samples (a scene that fills a bound >= 4 volume: finest level 8192), no two samples share a cell on
// them: each (sample, level) costs four 128-byte lines whatever the order, and the only cache that can serve them is
// one that holds the WHOLE level.  The fused kernel touches all 16 levels in every tile, so exactly those levels miss
// the L2 altogether - PMC at bound 4: 12.35 fabric requests per sample = 3 levels x 4 lines, 58 G requests/s against
// the 69 G random lines/s the XCD fabric ports deliver (profiles/r05_NOTES.txt 1-2).  This pre-pass evaluates the THREE
// finest levels (13..15) one level at a time over all samples - the level is the slow index of the launch, so at any
// moment every XCD's L2 holds the one level the chip is working on (tools/micro/level_xcd_bench.hip: 259 against 69 G
// lines/s; the kernel reaches 200) - and leaves 24 bytes per sample for k_nerf_fwd<.., kPre>, which gathers the other
// thirteen levels and reads these back, streamed.  2.8 ms per level and 141 M samples against 3.7 ms inside the fused
// kernel; level 12 costs 0.7 ms inside and stays there (history of the cut, from all eight fine levels on their own XCDs
// down to this: notes 3).  Lanes 2i, 2i+1 of a wave are the two x sides of sample i (one look-up for the x-neighbour
// rows).  Arithmetic = the fused kernel's fine slots, operation by operation: results are bit-identical.
constexpr int kSliceThreads = 256;
constexpr int kSliceTilesPerIter = 4;     // 32-sample tiles a wave has in flight (4 loads per lane each)
struct SliceX {
  float x0, x1, x2;
};
__device__ __forceinline__ SliceX slice_load_x(const float* __restrict__ x01, int64_t m, int64_t M) {
  const int64_t mm = m < M ? m : M - 1;
  const float* p = x01 + mm * 3;
  return SliceX{__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2)};
}
// grid (workgroups per level, kSliceLevels): the level is the SLOW index of the launch, so the chip sweeps the chunk's
// samples level by level and at any time (nearly) all workgroups of every XCD gather from the same 4 MiB
__global__ void __launch_bounds__(kSliceThreads) k_grid_fine_slices(const float* __restrict__ x01, int64_t M,
                                                                    const float2* __restrict__ emb, uint32_t emb_bytes,
                                                                    GridDesc G, float* __restrict__ pre) {
  const int slice = blockIdx.y;
  const int l = 16 - kSliceLevels + slice;
  const float sc = G.scales[l];
  const uint32_t m_ = G.mask[l], res1 = G.res1[l];
  const bool h = m_ != 0;
  const uint32_t base = G.offsets[l] * 8u, pa = h ? 2654435761u : res1, pb = h ? 805459861u : res1 * res1;
  const uint32_t mask = h ? m_ : 0xFFFFFFFFu;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t side = (uint32_t)(lane & 1);
  constexpr int kWaves = kSliceThreads / 64;
  constexpr int kT = kSliceTilesPerIter;
  const int64_t m_pad = (M + 31) & ~(int64_t)31;
  const int64_t n_iter = (M + 32 * kT - 1) / (32 * kT);
  float* __restrict__ out = pre + (int64_t)slice * m_pad * 2;
  for (int64_t it = (int64_t)blockIdx.x * kWaves + w; it < n_iter; it += (int64_t)gridDim.x * kWaves) {
    u32x2 rows[kT][4];
    float wx[kT], fy[kT], fz[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      const SliceX X = slice_load_x(x01, (it * kT + t) * 32 + (lane >> 1), M);
      const float px = X.x0 * sc + 0.5f, py = X.x1 * sc + 0.5f, pz = X.x2 * sc + 0.5f;       // mul, add: not fused
      const float fx = __builtin_amdgcn_fractf(px);
      wx[t] = side ? fx : 1.0f - fx;
      fy[t] = __builtin_amdgcn_fractf(py); fz[t] = __builtin_amdgcn_fractf(pz);
      const uint32_t c = (uint32_t)px + side, cy = (uint32_t)py, cz = (uint32_t)pz;
      const uint32_t hy0 = cy * pa, hy1 = hy0 + pa;
      const uint32_t hz0 = cz * pb, hz1 = hz0 + pb;
      const uint32_t yz[4] = {h ? (hy0 ^ hz0) : (hy0 + hz0), h ? (hy1 ^ hz0) : (hy1 + hz0),
                              h ? (hy0 ^ hz1) : (hy0 + hz1), h ? (hy1 ^ hz1) : (hy1 + hz1)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t idx = (h ? (c ^ yz[k]) : (c + yz[k])) & mask;
        rows[t][k] = gather_row<false>(rsrc, base + idx * 8u);
      }
    }
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      const f32x2 wy = {1.0f - fy[t], fy[t]};
      const f32x2 xy = wy * wx[t];
      const f32x2 w0 = xy * (1.0f - fz[t]), w1 = xy * fz[t];
      f32x2 acc = {0.f, 0.f};
      acc = __builtin_elementwise_fma(w0.xx, row2<false>(rows[t][0]), acc);
      acc = __builtin_elementwise_fma(w0.yy, row2<false>(rows[t][1]), acc);
      acc = __builtin_elementwise_fma(w1.xx, row2<false>(rows[t][2]), acc);
      acc = __builtin_elementwise_fma(w1.yy, row2<false>(rows[t][3]), acc);
      const float ox = __shfl_xor(acc.x, 1), oy = __shfl_xor(acc.y, 1);
      const float v = side ? acc.y + oy : acc.x + ox;          // both lanes hold side 0 + side 1; lane `side` stores feature `side`
      const int64_t m = (it * kT + t) * 32 + (lane >> 1);
      if (m < M) __builtin_nontemporal_store(v, out + m * 2 + side);
    }
  }
}

// ---- rgb-sigma lattice extraction (BASELINE configs[4]) ---------------------------------------
// out[m] = (mean over D fixed view directions of rgb(x_m, dir), raw density logit h0(x_m)): ONE gather and ONE
// sigma-net pass per point, then the colour net once per direction with the direction's SH row as a wave-uniform
// operand (staged in LDS).  Replaces density() + D x color() through the unfused encoder / BLAS path.
// kLattice (round 4): the points are a [W, L, H] lattice given by its three coordinate axes (x = ax_w[iw], y = ax_l[il],
// z = ax_h[ih]: exactly the caller's floats, no meshgrid tensor) and a 16-sample tile is a RUN ALONG W at fixed (il, ih).
// The writer's own order has h fastest; x is the fastest index of the table's rows (dense and hashed levels alike), so
// 16 consecutive h are 16 different 128-byte lines per level and corner where 16 consecutive w share one or two: the
// kernel is purely gather-bound (1 or 4 colour-net passes: 2.546 / 2.545 ms) and went 2.55 -> 1.07 ms for 160^3 points
// on the order alone (tools/extract_order_probe.py, profiles/r04_NOTES.txt 6).  out stays [W, L, H, 4].
struct LatticeDesc {
  const float* ax_w;
  const float* ax_l;
  const float* ax_h;
  int W, L, H, Wp;          // Wp = W rounded up to 16: a tile never straddles two (il, ih)
};
constexpr int kMaxExtractDirs = 8;
template <bool kLattice>
__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_nerf_fwd_dirs(
    const float* __restrict__ x, int64_t M, float bound, const float2* __restrict__ emb, uint32_t emb_bytes, GridDesc G,
    const float4* __restrict__ packed, const float* __restrict__ sh_dirs /*[D,16]*/, int D, float4* __restrict__ out,
    LatticeDesc lat, float logit_min) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  constexpr int kStage = kNerfFloats / 4;
  for (int i = threadIdx.x; i < kStage; i += kFieldThreads) wl[i] = packed[i];
  LevelRec* recs = reinterpret_cast<LevelRec*>(wl + kStage);
  stage_level_recs(G, recs);
  float* shl = reinterpret_cast<float*>(recs + 16);                   // [D][q][ks]: lane q's four components per direction
  for (int i = threadIdx.x; i < D * 16; i += kFieldThreads) {
    const int d = i >> 4, qq = (i >> 2) & 3, ks = i & 3;
    shl[i] = sh_dirs[d * 16 + 4 * ks + qq];
  }
  __syncthreads();
  constexpr int kWaves = kFieldThreads / 64;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int64_t n_tiles = (M + 15) >> 4;
  const float rb = 2.0f * bound;
  const TileSched sched = make_sched(n_tiles, kWaves);
  const bool all_hashed[4] = {slot_all_hashed(G, 0), slot_all_hashed(G, 1), slot_all_hashed(G, 2), slot_all_hashed(G, 3)};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);
  const float inv_d = 1.0f / (float)D;
  for (int64_t it = 0, tile = sched.tile(0); tile < sched.hi; tile = sched.tile(++it)) {
    int64_t m = tile * 16 + j;
    bool valid = m < M;
    TileIn me;
    if constexpr (kLattice) {
      // traversal index -> (il, ih, iw) with iw fastest; the caller clamps to [-bound, bound] like extract_rgbsigma did
      const int64_t row = m / lat.Wp;
      const int iw = (int)(m - row * lat.Wp);
      const int il = (int)(row / lat.H), ih = (int)(row - (int64_t)il * lat.H);
      valid = valid && iw < lat.W;
      const int cw = min(iw, lat.W - 1), cl = min(il, lat.L - 1);
      me.x0 = to_x01(fminf(fmaxf(lat.ax_w[cw], -bound), bound), bound, rb, 0.0f);
      me.x1 = to_x01(fminf(fmaxf(lat.ax_l[cl], -bound), bound), bound, rb, 0.0f);
      me.x2 = to_x01(fminf(fmaxf(lat.ax_h[ih], -bound), bound), bound, rb, 0.0f);
      me.oob = oob01(me.x0, me.x1, me.x2);
      if (me.oob) me.x0 = me.x1 = me.x2 = 0.0f;
      m = ((int64_t)cw * lat.L + cl) * lat.H + ih;      // where the point lives in [W, L, H]
    } else {
      load_tile_in<false>(x, nullptr, valid ? m : M - 1, bound, rb, 0.0f, me);
    }
    f32x4 enc[2];
    {
      Gathered g;
      uint32_t rec_off = (uint32_t)q * 4u * (uint32_t)sizeof(LevelRec);
      asm volatile("" : "+v"(rec_off));
      issue_gathers(reinterpret_cast<const LevelRec*>(reinterpret_cast<const char*>(recs) + rec_off), all_hashed, rsrc,
                    me.x0, me.x1, me.x2, g);
      __builtin_amdgcn_sched_barrier(0);
      blend(g, enc[0], enc[1]);
    }
    if (me.oob) enc[0] = enc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 h1[4], h2[1];
    mlp_layer<4, 2>(wl + kSig0 / 4, lane, enc, h1);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
    mlp_layer<1, 4>(wl + kSig1 / 4, lane, h1, h2);
    float r = 0.f, g = 0.f, b = 0.f;
    for (int d = 0; d < D; ++d) {                                       // wave-uniform trip count
      f32x4 cin[2], c1[4], c2[4], o[1];
      const float4 s4 = *reinterpret_cast<const float4*>(shl + d * 16 + q * 4);
      cin[0] = f32x4{s4.x, s4.y, s4.z, s4.w};
      cin[1] = h2[0];
      mlp_layer<4, 2>(wl + kCol0 / 4, lane, cin, c1);
#pragma unroll
      for (int t = 0; t < 4; ++t) c1[t] = relu4(c1[t]);
      mlp_layer<4, 4>(wl + kCol1 / 4, lane, c1, c2);
#pragma unroll
      for (int t = 0; t < 4; ++t) c2[t] = relu4(c2[t]);
      mlp_layer<1, 4>(wl + kCol2 / 4, lane, c2, o);
      r += __frcp_rn(1.0f + __expf(-o[0][0]));
      g += __frcp_rn(1.0f + __expf(-o[0][1]));
      b += __frcp_rn(1.0f + __expf(-o[0][2]));
    }
    if (valid && q == 0) out[m] = make_float4(r * inv_d, g * inv_d, b * inv_d, fmaxf(h2[0][0], logit_min));
  }
}

// kSave (training) 1: the encoder output and the two hidden activations are also written, row-major, for the
// weight gradients and the ReLU masks of k_instance_bwd; 2: the encoder output only (k_instance_head_bwd recomputes
// the hidden layers: 512 B per sample less to write here and to read there).
template <int K_MT, int kSave = 0>
__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_instance_fwd(const float* __restrict__ x, int64_t M,
                                                                   const int32_t* __restrict__ n_dev, float bound,
                                                                   const float2* __restrict__ emb, uint32_t emb_bytes,
                                                                   GridDesc G, const float4* __restrict__ packed,
                                                                   float* __restrict__ logits, float* __restrict__ enc_out,
                                                                   float* __restrict__ h1_out, float* __restrict__ h2_out) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  constexpr int K = K_MT * 16;
  constexpr int kStage = (kIns2 + K * 64) / 4;
  for (int i = threadIdx.x; i < kStage; i += kFieldThreads) wl[i] = packed[i];
  LevelRec* recs = reinterpret_cast<LevelRec*>(wl + kStage);
  stage_level_recs(G, recs);
  __syncthreads();

  constexpr int kWaves = kFieldThreads / 64;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  int64_t n = M;
  if (n_dev) n = min((int64_t)*n_dev, M);
  const int64_t n_tiles = (n + 15) >> 4;
  const float rb = 2.0f * bound;
  const float rb_inv = 0.0f;
  const TileSched sched = make_sched(n_tiles, kWaves);

  const bool all_hashed[4] = {slot_all_hashed(G, 0), slot_all_hashed(G, 1), slot_all_hashed(G, 2), slot_all_hashed(G, 3)};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);

  for (int64_t it = 0, tile = sched.tile(0); tile < sched.hi; tile = sched.tile(++it)) {
    const int64_t m = tile * 16 + j;
    const bool valid = m < n;
    TileIn me;
    load_tile_in<false>(x, nullptr, valid ? m : n - 1, bound, rb, rb_inv, me);
    f32x4 enc[2];
    {
      Gathered g;
      uint32_t rec_off = (uint32_t)q * 4u * (uint32_t)sizeof(LevelRec);
      asm volatile("" : "+v"(rec_off));    // opaque per tile: keeps the records in LDS, not hoisted into VGPRs
      issue_gathers(reinterpret_cast<const LevelRec*>(reinterpret_cast<const char*>(recs) + rec_off), all_hashed, rsrc,
                    me.x0, me.x1, me.x2, g);
      __builtin_amdgcn_sched_barrier(0);   // all 32 gathers in flight before the first blend waits
      blend(g, enc[0], enc[1]);
    }
    if (me.oob) enc[0] = enc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 h1[4], h2[4], o[K_MT];
    mlp_layer<4, 2>(wl + kIns0 / 4, lane, enc, h1);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
    mlp_layer<4, 4>(wl + kIns1 / 4, lane, h1, h2);
#pragma unroll
    for (int t = 0; t < 4; ++t) h2[t] = relu4(h2[t]);
    mlp_layer<K_MT, 4>(wl + kIns2 / 4, lane, h2, o);
    if (valid) {
#pragma unroll
      for (int mt = 0; mt < K_MT; ++mt) {
        float4 v = make_float4(o[mt][0], o[mt][1], o[mt][2], o[mt][3]);
        *reinterpret_cast<float4*>(logits + m * K + 16 * mt + 4 * q) = v;
      }
      if constexpr (kSave != 0) {
        // lane q holds encoder features 4q..4q+3 and 16+4q..16+4q+3, and rows 16t+4q..+3 of every hidden tile
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<float4*>(enc_out + m * 32 + 16 * t + 4 * q) = make_float4(enc[t][0], enc[t][1], enc[t][2], enc[t][3]);
      }
      if constexpr (kSave == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          *reinterpret_cast<float4*>(h1_out + m * 64 + 16 * t + 4 * q) = make_float4(h1[t][0], h1[t][1], h1[t][2], h1[t][3]);
          *reinterpret_cast<float4*>(h2_out + m * 64 + 16 * t + 4 * q) = make_float4(h2[t][0], h2[t][1], h2[t][2], h2[t][3]);
        }
      }
    }
  }
}

// ---- instance MLP input gradients (training) ----------------------------------------------------------------
// The chain dL/dlogits -> dL/dz2 -> dL/dz1 -> dL/denc is the same transposed formulation with W^T as the A
// operand: G^T = W^T . dY^T, so every layer's D registers are again the next layer's B operand.  Sections of
// `packed`: W2^T (64 x 64, inputs beyond K zero), W1^T (64 x 64), W0^T (32 x 64).  ReLU masks come from the saved
// activations (h > 0), loaded in the D-register layout with one float4 per tile.
constexpr int kBwd2 = 0;
constexpr int kBwd1 = kBwd2 + 64 * 64;
constexpr int kBwd0 = kBwd1 + 64 * 64;
constexpr int kBwdFloats = kBwd0 + 32 * 64;

__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_instance_bwd(const float* __restrict__ dlogits, int K,
                                                                   const float* __restrict__ h1, const float* __restrict__ h2,
                                                                   int64_t M, const float4* __restrict__ packed,
                                                                   float* __restrict__ dz2, float* __restrict__ dz1,
                                                                   float* __restrict__ denc) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  for (int i = threadIdx.x; i < kBwdFloats / 4; i += kFieldThreads) wl[i] = packed[i];
  __syncthreads();
  constexpr int kWaves = kFieldThreads / 64;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int64_t n_tiles = (M + 15) >> 4;
  const TileSched sched = make_sched(n_tiles, kWaves);
  for (int64_t it = 0, tile = sched.tile(0); tile < sched.hi; tile = sched.tile(++it)) {
    const int64_t m = tile * 16 + j;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    f32x4 g[4], hh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 16 * t + 4 * q;
      g[t] = (valid && c < K) ? load4(dlogits + mm * K + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      hh[t] = load4(h2 + mm * 64 + c);
    }
    f32x4 a[4], b[4], c0[2];
    mlp_layer<4, 4>(wl + kBwd2 / 4, lane, g, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = mask4(a[t], hh[t]);
      hh[t] = load4(h1 + mm * 64 + 16 * t + 4 * q);
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) store4(dz2 + m * 64 + 16 * t + 4 * q, a[t]);
    }
    mlp_layer<4, 4>(wl + kBwd1 / 4, lane, a, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = mask4(b[t], hh[t]);
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) store4(dz1 + m * 64 + 16 * t + 4 * q, b[t]);
    }
    mlp_layer<2, 4>(wl + kBwd0 / 4, lane, b, c0);
    if (valid) {
#pragma unroll
      for (int t = 0; t < 2; ++t) store4(denc + m * 32 + 16 * t + 4 * q, c0[t]);
    }
  }
}

// ---- instance head: the WHOLE backward in one launch (round 3) -----------------------------------------------
// Replaces, for the instance stage, k_composite_train_extra_bwd (dL/dlogits rows [M,K] written and read back),
// k_instance_bwd (which read the saved h1 / h2 and wrote dz2 / dz1 for the weight gradients) and the three
// k_linear_wgrad launches (which read all of them again): per sample it reads the saved encoder output (128 B), the
// compositing weight and the ray's row of dL/d(rendered logits), and writes dL/denc (128 B) - ~0.3 KB instead of
// ~3.5 KB of HBM traffic per sample.  Per 16-sample tile a wave
//   1. forms g = w * dL/dpix[ray] (the K-channel compositing backward, weights detached),
//   2. recomputes h1 = relu(W0 enc), h2 = relu(W1 h1) with the forward's own weights and code (same bits: the ReLU
//      masks are the forward's),
//   3. runs the input-gradient chain a = (W2^T g) . [h2 > 0], b = (W1^T a) . [h1 > 0], denc = W0^T b,
//   4. and the weight gradients dW2 += g^T h2, dW1 += a^T h1, dW0 += b^T enc on the fp32 matrix cores
//      (v_mfma_f32_16x16x4_f32, exact products, k = 4 samples per instruction).  The contraction runs over SAMPLES,
//      which sit in lane & 15 of the D-layout registers: the (G, X) tiles of the workgroup's eight waves are staged in
//      LDS row-major (row pitch 68 floats: the 16-byte writes of the 16 lanes of a q group fall into different
//      banks) and read back as A[i = channel][k = sample], B[k = sample][j = channel].  The 40 accumulator tiles
//      (16 + 16 + 8 of 16x16) are OWNED by the waves - wave w: tiles 2w, 2w+1 of dW2 and dW1, tile w of dW0, 20
//      VGPRs - and every wave consumes all eight staged tiles for its own accumulators.
// The workgroup's accumulators go to partial[block] (fragment order); k_head_wgrad_reduce sums over the workgroups.
// Rows >= n (the device-side sample count; the buffer is sized from mean_count) are not read: their dL/denc is zero.
constexpr int kHbPitch = 68;
constexpr int kHbWaves = kFieldThreads / 64;
constexpr int kHbTileFloats = 16 * kHbPitch;
constexpr int kHbStageFloats = 2 * kHbWaves * kHbTileFloats;            // G and X of one round: 69.6 KB
constexpr int kHbFwdFloats = kIns2;                                      // forward sections kIns0, kIns1
constexpr int kHbAccTiles = 40;

// The table-gradient buffer of the scatter that FOLLOWS a head-backward launch is zero-filled by that launch (every
// workgroup a slice, before its first tile): one launch and one exposed 49 MB fill less per training step.
__device__ __forceinline__ void hb_zero_fill(float* __restrict__ buf, int64_t n) {
  if (!buf) return;
  float4* b4 = reinterpret_cast<float4*>(buf);
  const int64_t n4 = n >> 2;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) b4[i] = z;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) buf[(n4 << 2) + threadIdx.x] = 0.f;
}

#if INR_MLP_FP32
// exact-fp32 build: the (G, X) tiles staged row-major as fp32, contraction with v_mfma_f32_16x16x4_f32
typedef float HbStage;
__device__ __forceinline__ HbStage* hb_my_tile(float* plane, int wave) { return plane + wave * kHbTileFloats; }
template <int N_T>
__device__ __forceinline__ void hb_stage(HbStage* __restrict__ tile_base, int q, int j, const f32x4* v) {
#pragma unroll
  for (int t = 0; t < N_T; ++t) store4(tile_base + j * kHbPitch + 16 * t + 4 * q, v[t]);
}
__device__ __forceinline__ void hb_stage_value(HbStage* __restrict__ tile_base, int j, int channel, float v) {
  tile_base[j * kHbPitch + channel] = v;
}

// acc0 += G[:, ot]^T X[:, it0], acc1 += G[:, ot]^T X[:, it1] over the 8 x 16 staged samples
__device__ __forceinline__ void hb_accum(const float* __restrict__ gS, const float* __restrict__ xS, int q, int j, int ot,
                                         int it0, int it1, bool two, f32x4& acc0, f32x4& acc1) {
#pragma unroll 2
  for (int T = 0; T < kHbWaves; ++T) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = T * kHbTileFloats + (4 * u + q) * kHbPitch;
      const float a = gS[row + 16 * ot + j];
      const float b0 = xS[row + 16 * it0 + j];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
      if (two) {
        const float b1 = xS[row + 16 * it1 + j];
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
      }
    }
  }
}
#else
// Default build (late round 3): the weight gradients take the bf16 matrix cores too.  The fp32 contraction
// (v_mfma_f32_16x16x4_f32: 4 samples per 32-cycle instruction) was 62 % of these kernels' MFMA time; split into bf16
// head + remainder like the MLP layers (hi*hi + lo*hi + hi*lo, fp32 accumulation: 2^-16 relative per product, random
// over ~2e5 samples) the same 32 samples cost three 16-cycle v_mfma_f32_16x16x32_bf16.  The operands of that
// instruction hold 8 CONSECUTIVE k (= samples) per lane, so the tiles are staged TRANSPOSED - [channel][sample], a head
// plane and a remainder plane of 64 x 136 bf16 each (pitch 136 halfwords = 68 dwords: rows land 4 banks apart, the
// 16-byte reads of a 16-lane row spread over all banks) - exactly the 34 816 bytes per operand of the fp32 staging.
typedef uint16_t HbStage;
constexpr int kHbPitchH = 136;                           // halfwords per channel row: 128 samples + 8
constexpr int kHbPlaneH = 64 * kHbPitchH;                // halfwords per plane (head | remainder)
static_assert(2 * kHbPlaneH * 2 == kHbWaves * kHbTileFloats * 4, "bf16 staging must fit the fp32 staging's footprint");
__device__ __forceinline__ HbStage* hb_my_tile(float* plane, int wave) {
  return reinterpret_cast<HbStage*>(plane) + wave * 16;  // this wave's 16 sample columns
}
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2h_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t hb_pack_bf16x2(float a, float b) {       // v_cvt_pk_bf16_f32: round to nearest even
  const f32x2h_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// head / remainder of two values -> (hi0 | hi1 << 16), (lo0 | lo1 << 16)
__device__ __forceinline__ void hb_split2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  hi = hb_pack_bf16x2(v0, v1);
  lo = hb_pack_bf16x2(v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xFFFF0000u));
}
__device__ __forceinline__ void hb_put(HbStage* __restrict__ tile_base, int j, int channel, uint32_t hi16, uint32_t lo16) {
  tile_base[channel * kHbPitchH + j] = (uint16_t)hi16;
  tile_base[kHbPlaneH + channel * kHbPitchH + j] = (uint16_t)lo16;
}
template <int N_T>
__device__ __forceinline__ void hb_stage(HbStage* __restrict__ tile_base, int q, int j, const f32x4* v) {
#pragma unroll
  for (int t = 0; t < N_T; ++t) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      uint32_t hi, lo;
      hb_split2(v[t][r], v[t][r + 1], hi, lo);
      hb_put(tile_base, j, 16 * t + 4 * q + r, hi & 0xFFFFu, lo & 0xFFFFu);
      hb_put(tile_base, j, 16 * t + 4 * q + r + 1, hi >> 16, lo >> 16);
    }
  }
}
__device__ __forceinline__ void hb_stage_value(HbStage* __restrict__ tile_base, int j, int channel, float v) {
  uint32_t hi, lo;
  hb_split2(v, 0.0f, hi, lo);
  hb_put(tile_base, j, channel, hi & 0xFFFFu, lo & 0xFFFFu);
}

// acc0 += G[:, ot]^T X[:, it0], acc1 += G[:, ot]^T X[:, it1] over the 128 staged samples (4 k-blocks of 32)
__device__ __forceinline__ void hb_accum(const float* __restrict__ gS, const float* __restrict__ xS, int q, int j, int ot,
                                         int it0, int it1, bool two, f32x4& acc0, f32x4& acc1) {
  const uint16_t* G = reinterpret_cast<const uint16_t*>(gS) + (16 * ot + j) * kHbPitchH + 8 * q;
  const uint16_t* X0 = reinterpret_cast<const uint16_t*>(xS) + (16 * it0 + j) * kHbPitchH + 8 * q;
  const uint16_t* X1 = reinterpret_cast<const uint16_t*>(xS) + (16 * it1 + j) * kHbPitchH + 8 * q;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const uint4 a_hi = *reinterpret_cast<const uint4*>(G + 32 * kb), a_lo = *reinterpret_cast<const uint4*>(G + kHbPlaneH + 32 * kb);
    const uint4 b_hi = *reinterpret_cast<const uint4*>(X0 + 32 * kb), b_lo = *reinterpret_cast<const uint4*>(X0 + kHbPlaneH + 32 * kb);
    acc0 = mfma_bf(a_hi, b_hi, acc0);
    acc0 = mfma_bf(a_lo, b_hi, acc0);
    acc0 = mfma_bf(a_hi, b_lo, acc0);
    if (two) {
      const uint4 c_hi = *reinterpret_cast<const uint4*>(X1 + 32 * kb), c_lo = *reinterpret_cast<const uint4*>(X1 + kHbPlaneH + 32 * kb);
      acc1 = mfma_bf(a_hi, c_hi, acc1);
      acc1 = mfma_bf(a_lo, c_hi, acc1);
      acc1 = mfma_bf(a_hi, c_lo, acc1);
    }
  }
}
#endif

__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_instance_head_bwd(
    const float* __restrict__ enc, const float* __restrict__ wbuf, const int32_t* __restrict__ ray_of,
    const float* __restrict__ g_pix, int Kp, int64_t M, const int32_t* __restrict__ n_dev,
    const float* __restrict__ scale_a, const float* __restrict__ scale_b,
    const float4* __restrict__ packed_fwd, const float4* __restrict__ packed_bwd, float* __restrict__ denc,
    float4* __restrict__ partial, float* __restrict__ zero_buf, int64_t zero_n) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  hb_zero_fill(zero_buf, zero_n);
  float4* wf = wl;
  float4* wb = wl + kHbFwdFloats / 4;
  float* gS = reinterpret_cast<float*>(wb + kBwdFloats / 4);
  float* xS = gS + kHbWaves * kHbTileFloats;
  for (int i = threadIdx.x; i < kHbFwdFloats / 4; i += kFieldThreads) wf[i] = packed_fwd[i];
  for (int i = threadIdx.x; i < kBwdFloats / 4; i += kFieldThreads) wb[i] = packed_bwd[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  int64_t n = M;
  if (n_dev) n = min((int64_t)*n_dev, M);
  // g_pix may be an unnormalised gradient (softmax - onehot of the fused cross entropy): the two device scalars -
  // 1 / kept rows and dL/dloss - scale every sample's weight instead of a pass over [N,K]
  const float g_scale = (scale_a ? *scale_a : 1.0f) * (scale_b ? *scale_b : 1.0f);
  const int64_t n_tiles = (M + 15) >> 4;
  const int64_t per_round = (int64_t)gridDim.x * kHbWaves;
  const int64_t n_rounds = (n_tiles + per_round - 1) / per_round;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[5] = {zero4, zero4, zero4, zero4, zero4};
  HbStage* my_g = hb_my_tile(gS, wave);
  HbStage* my_x = hb_my_tile(xS, wave);
  const int ot = wave >> 1, it0 = 2 * (wave & 1), it1 = it0 + 1;
  // Inputs of the NEXT round are requested while this round's weight-gradient phases run (the kernel is a chain of
  // dependent round trips otherwise: 7 rounds x (HBM load -> dependent row load -> MFMA chain -> 6 barriers)):
  // stage 1 (encoder row, weight, ray row) right after this round's inputs are consumed, stage 2 (the ray's
  // dL/dpix row, which needs the ray row) before the last phase.
  f32x4 e[2] = {zero4, zero4}, graw[4] = {zero4, zero4, zero4, zero4};
  float w = 0.0f;
  auto tile_row = [&](int64_t round) { return (round * per_round + (int64_t)blockIdx.x * kHbWaves + wave) * 16 + j; };
  {
    const int64_t m = tile_row(0);
    if (m < n) {
      e[0] = load4(enc + m * 32 + 4 * q);
      e[1] = load4(enc + m * 32 + 16 + 4 * q);
      w = wbuf[m] * g_scale;
      const float* gp = g_pix + (int64_t)ray_of[m] * Kp;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * t + 4 * q < Kp) graw[t] = load4(gp + 16 * t + 4 * q);
    }
  }
  for (int64_t round = 0; round < n_rounds; ++round) {
    const int64_t m = tile_row(round);
    f32x4 g[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = graw[t] * w;
    f32x4 h1[4], h2[4], a[4], b[4], c0[2];
    mlp_layer<4, 2>(wf + kIns0 / 4, lane, e, h1);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
    mlp_layer<4, 4>(wf + kIns1 / 4, lane, h1, h2);
#pragma unroll
    for (int t = 0; t < 4; ++t) h2[t] = relu4(h2[t]);
    hb_stage<4>(my_g, q, j, g);
    hb_stage<4>(my_x, q, j, h2);
    const f32x4 e_cur[2] = {e[0], e[1]};
    // stage 1 of the next round
    const int64_t mn = tile_row(round + 1);
    const bool next_valid = round + 1 < n_rounds && mn < n;
    int32_t rid_n = 0;
    e[0] = e[1] = zero4;
    w = 0.0f;
    if (next_valid) {
      e[0] = load4(enc + mn * 32 + 4 * q);
      e[1] = load4(enc + mn * 32 + 16 + 4 * q);
      w = wbuf[mn] * g_scale;
      rid_n = ray_of[mn];
    }
    mlp_layer<4, 4>(wb + kBwd2 / 4, lane, g, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = mask4(a[t], h2[t]);
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, it0, it1, true, acc[0], acc[1]);                 // dW2 tiles 2w, 2w+1
    mlp_layer<4, 4>(wb + kBwd1 / 4, lane, a, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = mask4(b[t], h1[t]);
    __syncthreads();
    hb_stage<4>(my_g, q, j, a);
    hb_stage<4>(my_x, q, j, h1);
    mlp_layer<2, 4>(wb + kBwd0 / 4, lane, b, c0);
    if (m < M) {
      store4(denc + m * 32 + 4 * q, c0[0]);
      store4(denc + m * 32 + 16 + 4 * q, c0[1]);
    }
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, it0, it1, true, acc[2], acc[3]);                 // dW1 tiles 2w, 2w+1
    __syncthreads();
    // stage 2 of the next round: the ray row has arrived long ago
#pragma unroll
    for (int t = 0; t < 4; ++t) graw[t] = zero4;
    if (next_valid) {
      const float* gp = g_pix + (int64_t)rid_n * Kp;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * t + 4 * q < Kp) graw[t] = load4(gp + 16 * t + 4 * q);
    }
    hb_stage<4>(my_g, q, j, b);
    hb_stage<2>(my_x, q, j, e_cur);
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, wave & 1, 0, false, acc[4], acc[4]);             // dW0 tile (ot, it = w & 1)
    __syncthreads();
  }
  float4* out = partial + (size_t)blockIdx.x * kHbAccTiles * 64;
  const int ids[5] = {2 * wave, 2 * wave + 1, 16 + 2 * wave, 16 + 2 * wave + 1, 32 + wave};
#pragma unroll
  for (int k = 0; k < 5; ++k) out[ids[k] * 64 + lane] = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
}

// gw2 [Kp,64], gw1 [64,64], gw0 [64,32] = sum over the workgroups' partials (written, not accumulated).
// Element idx of a partial: tile = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3 -> D[4 (lane >> 4) + r][lane & 15].
__global__ void __launch_bounds__(1024) k_head_wgrad_reduce(const float* __restrict__ partial, int n_groups, int Kp,
                                                            float* __restrict__ gw0, float* __restrict__ gw1,
                                                            float* __restrict__ gw2) {
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  constexpr int per = kHbAccTiles * 256;
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
    const int tile = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
    const int ol = 4 * (lane >> 4) + r, il = lane & 15;
    if (tile < 16) {
      const int o = 16 * (tile >> 2) + ol, i = 16 * (tile & 3) + il;
      if (o < Kp) gw2[o * 64 + i] = s;
    } else if (tile < 32) {
      const int t = tile - 16;
      gw1[(16 * (t >> 2) + ol) * 64 + 16 * (t & 3) + il] = s;
    } else {
      const int t = tile - 32;
      gw0[(16 * (t >> 1) + ol) * 32 + 16 * (t & 1) + il] = s;
    }
  }
}

// ---- NeRF field input gradients (training) -------------------------------------------------------------------
// (g_sigma, g_rgb) -> colour net -> geo features + density logit -> sigma net -> dL/denc, one launch, same
// transposed formulation as k_instance_bwd.  Sections of `packed` (W^T as the A operand, zero-padded):
//   C2T 64 x 32 (3 live inputs)   C1T 64 x 64   C0T 16 x 64 (row i = sigma-net output row i: row 0 zero, row i >= 1
//   is colour-net input column 15 + i)   S1T 64 x 32 (16 live inputs)   S0T 32 x 64.
constexpr int kNbC2 = 0;
constexpr int kNbC1 = kNbC2 + 64 * 32;
constexpr int kNbC0 = kNbC1 + 64 * 64;
constexpr int kNbS1 = kNbC0 + 16 * 64;
constexpr int kNbS0 = kNbS1 + 64 * 32;
constexpr int kNerfBwdFloats = kNbS0 + 32 * 64;      // 11264 floats = 44 KB

struct NerfBwdIO {
  const float *g_sigma, *g_rgb, *rgb, *so, *h1, *c1, *c2;     // inputs
  float *d_o, *dz_c2, *dz_c1, *d_so, *dz_h1, *d_enc;           // outputs: [M,4] [M,64] [M,64] [M,16] [M,64] [M,32]
};

__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_nerf_bwd(NerfBwdIO io, int64_t M, float density_scale,
                                                               const float4* __restrict__ packed) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  for (int i = threadIdx.x; i < kNerfBwdFloats / 4; i += kFieldThreads) wl[i] = packed[i];
  __syncthreads();
  constexpr int kWaves = kFieldThreads / 64;
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const int64_t n_tiles = (M + 15) >> 4;
  const TileSched sched = make_sched(n_tiles, kWaves);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int64_t it = 0, tile = sched.tile(0); tile < sched.hi; tile = sched.tile(++it)) {
    const int64_t m = tile * 16 + j;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    // d(rgb logits): rows 0..2 of a 16-row tile live in lane q == 0
    f32x4 g[2] = {zero4, zero4};
    if (valid && q == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float c = io.rgb[mm * 3 + r];
        g[0][r] = io.g_rgb[mm * 3 + r] * c * (1.0f - c);
      }
    }
    if (valid && q == 0) store4(io.d_o + m * 4, g[0]);
    f32x4 hh[4], a[4], b[4], ds[2], e0[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) hh[t] = load4(io.c2 + mm * 64 + 16 * t + 4 * q);
    mlp_layer<4, 2>(wl + kNbC2 / 4, lane, g, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t] = mask4(a[t], hh[t]);
      hh[t] = load4(io.c1 + mm * 64 + 16 * t + 4 * q);
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) store4(io.dz_c2 + m * 64 + 16 * t + 4 * q, a[t]);
    }
    mlp_layer<4, 4>(wl + kNbC1 / 4, lane, a, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      b[t] = mask4(b[t], hh[t]);
      hh[t] = load4(io.h1 + mm * 64 + 16 * t + 4 * q);
    }
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) store4(io.dz_c1 + m * 64 + 16 * t + 4 * q, b[t]);
    }
    mlp_layer<1, 4>(wl + kNbC0 / 4, lane, b, ds);      // rows 1..15: d(geo); row 0: 0
    ds[1] = zero4;
    if (q == 0) {
      // density: sigma = density_scale * exp(so0), trunc_exp backward = g * exp(clamp(so0, -15, 15))
      const float so0 = io.so[mm * 16];
      ds[0][0] = valid ? io.g_sigma[mm] * density_scale * __expf(fminf(fmaxf(so0, -15.0f), 15.0f)) : 0.f;
    }
    if (valid) store4(io.d_so + m * 16 + 4 * q, ds[0]);
    mlp_layer<4, 2>(wl + kNbS1 / 4, lane, ds, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = mask4(a[t], hh[t]);
    if (valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t) store4(io.dz_h1 + m * 64 + 16 * t + 4 * q, a[t]);
    }
    mlp_layer<2, 4>(wl + kNbS0 / 4, lane, a, e0);
    if (valid) {
#pragma unroll
      for (int t = 0; t < 2; ++t) store4(io.d_enc + m * 32 + 16 * t + 4 * q, e0[t]);
    }
  }
}

// ---- NeRF field: the WHOLE backward in one launch (round 3; the NeRF-stage twin of k_instance_head_bwd) ---------
// Per 16-sample tile a wave recomputes the forward from the saved encoder output and the view direction (sigma net,
// SH, colour net - the forward's own weights and code, so the ReLU masks and the sigmoid are the forward's), runs the
// input-gradient chain of k_nerf_bwd, and accumulates the five weight gradients on the fp32 matrix cores through the
// same LDS staging as k_instance_head_bwd: dWc2 += d_o^T c2 (4 tiles), dWc1 += dz_c2^T c1 (16), dWc0 += dz_c1^T cin (8),
// dWs1 += d_so^T h1 (4), dWs0 += dz_h1^T enc (8) = 40 accumulator tiles, five per wave (wave w: tiles 2w, 2w+1 of dWc1,
// tile w of dWc0 and dWs0, and tile w of dWc2 (w < 4) or tile w - 4 of dWs1 (w >= 4)).  Replaces the activation
// stores of the training forward (1088 -> 128 B per sample), k_nerf_bwd and five k_linear_wgrad launches.
constexpr int kNhFwdFloats = kNerfFloats;

// colour-net input in weight-column order (16 SH, 15 geo, 1 zero) from the register layout of the forward:
// cin0[ks] = SH component 4 ks + q, so[r] = sigma-net row 4 q + r (row 0 is the density logit: column 31 gets 0)
__device__ __forceinline__ void nh_stage_cin(HbStage* __restrict__ tile_base, int q, int j, const f32x4 cin0, const f32x4 so) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) hb_stage_value(tile_base, j, 4 * ks + q, cin0[ks]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int srow = 4 * q + r;
    hb_stage_value(tile_base, j, srow >= 1 ? 15 + srow : 31, srow >= 1 ? so[r] : 0.0f);
  }
}

__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_nerf_head_bwd(
    const float* __restrict__ enc, const float* __restrict__ dirs, const float* __restrict__ g_sigma,
    const float* __restrict__ g_rgb, int64_t M, float density_scale, const float4* __restrict__ packed_fwd,
    const float4* __restrict__ packed_bwd, float* __restrict__ d_enc, float4* __restrict__ partial,
    float* __restrict__ zero_buf, int64_t zero_n) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  hb_zero_fill(zero_buf, zero_n);
  float4* wf = wl;
  float4* wb = wl + kNhFwdFloats / 4;
  float* gS = reinterpret_cast<float*>(wb + kNerfBwdFloats / 4);
  float* xS = gS + kHbWaves * kHbTileFloats;
  for (int i = threadIdx.x; i < kNhFwdFloats / 4; i += kFieldThreads) wf[i] = packed_fwd[i];
  for (int i = threadIdx.x; i < kNerfBwdFloats / 4; i += kFieldThreads) wb[i] = packed_bwd[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  const int64_t n_tiles = (M + 15) >> 4;
  const int64_t per_round = (int64_t)gridDim.x * kHbWaves;
  const int64_t n_rounds = (n_tiles + per_round - 1) / per_round;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[5] = {zero4, zero4, zero4, zero4, zero4};
  HbStage* my_g = hb_my_tile(gS, wave);
  HbStage* my_x = hb_my_tile(xS, wave);
  const int ot = wave >> 1, it0 = 2 * (wave & 1), it1 = it0 + 1;
  for (int64_t round = 0; round < n_rounds; ++round) {
    const int64_t tile = round * per_round + (int64_t)blockIdx.x * kHbWaves + wave;
    const int64_t m = tile * 16 + j;
    const bool valid = m < M;
    f32x4 e[2] = {zero4, zero4};
    float d0 = 0.f, d1 = 0.f, d2 = 1.f, gs = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f;
    if (valid) {
      e[0] = load4(enc + m * 32 + 4 * q);
      e[1] = load4(enc + m * 32 + 16 + 4 * q);
      d0 = dirs[m * 3]; d1 = dirs[m * 3 + 1]; d2 = dirs[m * 3 + 2];
      if (q == 0) {
        gs = g_sigma[m];
        gr0 = g_rgb[m * 3]; gr1 = g_rgb[m * 3 + 1]; gr2 = g_rgb[m * 3 + 2];
      }
    }
    // ---- forward, recomputed (same code as k_nerf_fwd) ----
    f32x4 h1[4], so[1], cin[2], c1[4], c2[4], o[1];
    mlp_layer<4, 2>(wf + kSig0 / 4, lane, e, h1);
#pragma unroll
    for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
    mlp_layer<1, 4>(wf + kSig1 / 4, lane, h1, so);
    {
      float sh[16];
      sh4(d0, d1, d2, sh);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) cin[0][ks] = select4(q, sh[4 * ks], sh[4 * ks + 1], sh[4 * ks + 2], sh[4 * ks + 3]);
    }
    cin[1] = so[0];
    mlp_layer<4, 2>(wf + kCol0 / 4, lane, cin, c1);
#pragma unroll
    for (int t = 0; t < 4; ++t) c1[t] = relu4(c1[t]);
    mlp_layer<4, 4>(wf + kCol1 / 4, lane, c1, c2);
#pragma unroll
    for (int t = 0; t < 4; ++t) c2[t] = relu4(c2[t]);
    mlp_layer<1, 4>(wf + kCol2 / 4, lane, c2, o);
    // ---- d(rgb logits): rows 0..2 of a 16-row tile live in lanes q == 0 ----
    f32x4 g[2] = {zero4, zero4};
    if (valid && q == 0) {
      const float c0v = __frcp_rn(1.0f + __expf(-o[0][0])), c1v = __frcp_rn(1.0f + __expf(-o[0][1])),
                  c2v = __frcp_rn(1.0f + __expf(-o[0][2]));
      g[0][0] = gr0 * c0v * (1.0f - c0v);
      g[0][1] = gr1 * c1v * (1.0f - c1v);
      g[0][2] = gr2 * c2v * (1.0f - c2v);
    }
    // phase 1: dWc2 (16 x 64, 3 live rows) += d_o^T c2          [waves 0..3: tile it = wave]
    hb_stage<1>(my_g, q, j, g);
    hb_stage<4>(my_x, q, j, c2);
    f32x4 a[4], b[4], ds[2], e0[2];
    mlp_layer<4, 2>(wb + kNbC2 / 4, lane, g, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = mask4(a[t], c2[t]);
    __syncthreads();
    if (wave < 4) hb_accum(gS, xS, q, j, 0, wave, 0, false, acc[4], acc[4]);
    mlp_layer<4, 4>(wb + kNbC1 / 4, lane, a, b);
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = mask4(b[t], c1[t]);
    __syncthreads();
    // phase 2: dWc1 += dz_c2^T c1
    hb_stage<4>(my_g, q, j, a);
    hb_stage<4>(my_x, q, j, c1);
    mlp_layer<1, 4>(wb + kNbC0 / 4, lane, b, ds);      // rows 1..15: d(geo); row 0: 0
    ds[1] = zero4;
    if (q == 0) ds[0][0] = valid ? gs * density_scale * __expf(fminf(fmaxf(so[0][0], -15.0f), 15.0f)) : 0.f;
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, it0, it1, true, acc[0], acc[1]);
    mlp_layer<4, 2>(wb + kNbS1 / 4, lane, ds, a);
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = mask4(a[t], h1[t]);
    __syncthreads();
    // phase 3: dWc0 (64 x 32) += dz_c1^T cin
    hb_stage<4>(my_g, q, j, b);
    nh_stage_cin(my_x, q, j, cin[0], so[0]);
    mlp_layer<2, 4>(wb + kNbS0 / 4, lane, a, e0);
    if (valid) {
      store4(d_enc + m * 32 + 4 * q, e0[0]);
      store4(d_enc + m * 32 + 16 + 4 * q, e0[1]);
    }
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, wave & 1, 0, false, acc[2], acc[2]);
    __syncthreads();
    // phase 4: dWs1 (16 x 64) += d_so^T h1                     [waves 4..7: tile it = wave - 4]
    hb_stage<1>(my_g, q, j, ds);
    hb_stage<4>(my_x, q, j, h1);
    __syncthreads();
    if (wave >= 4) hb_accum(gS, xS, q, j, 0, wave - 4, 0, false, acc[4], acc[4]);
    __syncthreads();
    // phase 5: dWs0 (64 x 32) += dz_h1^T enc
    hb_stage<4>(my_g, q, j, a);
    hb_stage<2>(my_x, q, j, e);
    __syncthreads();
    hb_accum(gS, xS, q, j, ot, wave & 1, 0, false, acc[3], acc[3]);
    __syncthreads();
  }
  float4* out = partial + (size_t)blockIdx.x * kHbAccTiles * 64;
  // tile ids: 0..15 dWc1 (ot, it), 16..23 dWc0, 24..31 dWs0, 32..35 dWc2 (it), 36..39 dWs1 (it)
  const int ids[5] = {2 * wave, 2 * wave + 1, 16 + wave, 24 + wave, 32 + wave};
#pragma unroll
  for (int k = 0; k < 5; ++k) out[ids[k] * 64 + lane] = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
}

// sums the workgroups' partials into gwc2 [16,64] (3 live rows), gwc1 [64,64], gwc0 [64,31],
// gws1 [16,64], gws0 [64,32] - written, not accumulated.
__global__ void __launch_bounds__(1024) k_nerf_head_wgrad_reduce(const float* __restrict__ partial, int n_groups,
                                                                 float* __restrict__ gwc2, float* __restrict__ gwc1,
                                                                 float* __restrict__ gwc0, float* __restrict__ gws1,
                                                                 float* __restrict__ gws0) {
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  constexpr int per = kHbAccTiles * 256;
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
    const int tile = idx >> 8, lane = (idx >> 2) & 63, r = idx & 3;
    const int ol = 4 * (lane >> 4) + r, il = lane & 15;
    if (tile < 16) gwc1[(16 * (tile >> 2) + ol) * 64 + 16 * (tile & 3) + il] = s;
    else if (tile < 24) {                   // [64,31] as nn.Linear stores it (the padding column 31 is dropped)
      const int col = 16 * ((tile - 16) & 1) + il;
      if (col < 31) gwc0[(16 * ((tile - 16) >> 1) + ol) * 31 + col] = s;
    }
    else if (tile < 32) gws0[(16 * ((tile - 24) >> 1) + ol) * 32 + 16 * ((tile - 24) & 1) + il] = s;
    else if (tile < 36) gwc2[ol * 64 + 16 * (tile - 32) + il] = s;
    else gws1[ol * 64 + 16 * (tile - 36) + il] = s;
  }
}

// Device-side weight packing for training (the weights change every step; the host packers above would cost a
// device->host->device round trip per step).  One thread per packed bf16 pair position; same layout and
// rounding as pack_section_bf16.  transpose: the section holds W^T of the row-major [n_rows_w, n_cols_w] weight.
struct PackJob {
  const float* W;
  int w_rows, w_cols;      // row-major weight as stored by nn.Linear: [out, in]
  int n_out, n_in;         // logical section size (rows = MFMA output rows, cols = inputs)
  int n_mt, n_s;
  int transpose;
  int dst_off;             // float offset of the section in the packed buffer
  int kmode;               // 0: k-slot = feature 16*(ks>>2)+4q+(ks&3); 1: colour-net input (kidx_color_in)
  int row_min, row_shift;  // transpose only: rows below row_min are zero, row i reads source column i + row_shift
  int bwd;                 // which output buffer
};
struct PackJobs {
  PackJob j[10];
};

__device__ __forceinline__ uint16_t bf16_rne_dev(float x) {
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ void __launch_bounds__(256) k_pack_weights(PackJobs jobs, float* __restrict__ packed_fwd,
                                                      float* __restrict__ packed_bwd) {
  const PackJob J = jobs.j[blockIdx.y];
  const int total = J.n_mt * J.n_s * 64 * 8;
  float* base = (J.bwd ? packed_bwd : packed_fwd) + J.dst_off;
  uint16_t* dst = reinterpret_cast<uint16_t*>(base);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63, rest = i >> 9;
    const int st = rest % J.n_s, mt = rest / J.n_s;
    const int row = 16 * mt + (lane & 15);
    const int qq = lane >> 4;
    int col = 16 * (2 * st + (e >> 2)) + 4 * qq + (e & 3);
    if (J.kmode == 1) {                                  // colour-net input: 4 SH slots, then sigma-net rows (row 0 unused)
      const int r = 4 * qq + (e & 3);
      col = (e >> 2) == 0 ? 4 * (e & 3) + qq : (r >= 1 ? 15 + r : -1);
    }
    float v = 0.f;
    if (row < J.n_out && col >= 0 && col < J.n_in) {
      if (!J.transpose) v = J.W[(size_t)row * J.w_cols + col];
      else if (row >= J.row_min) v = J.W[(size_t)col * J.w_cols + row + J.row_shift];
    }
#if INR_MLP_FP32
    // exact-fp32 build (round 5: the training entry points too): the fragment order of pack_section_f32 - k-step
    // ks = 4 (2 st + (e >> 2)) + (e & 3) of the bf16 layout's element (st, e), value (mt, ks, lane) at frag_pos()
    (void)dst;
    base[frag_pos(mt, 4 * (2 * st + (e >> 2)) + (e & 3), lane, 8 * J.n_s)] = v;
#else
    const uint16_t hi = bf16_rne_dev(v);
    const uint16_t lo = bf16_rne_dev(v - __uint_as_float((uint32_t)hi << 16));
    dst[((((size_t)(mt * J.n_s + st) * 2 + 0) * 64 + lane) * 8) + e] = hi;
    dst[((((size_t)(mt * J.n_s + st) * 2 + 1) * 64 + lane) * 8) + e] = lo;
#endif
  }
}


// ---- instance logits rendered in place (inference) ---------------------------------------------------------
// out[ray][ch] = sum_k w(ray,k) * logits(x(ray,k))[ch] without ever writing the [M, K] logits.  A WORKGROUP owns a
// 16-ray group of the patch-interleaved layout; its eight waves take the group's steps round robin (wave w: steps w,
// w + 8, ...), so that at any time the CU works on eight consecutive depths of one 4x4 patch - the access pattern the
// plain field kernel owes its L1 hit rate to (round 1 / early round 2 gave each wave a group of its own: eight
// unrelated patches per CU, 7.8 ms for 37 M samples).  MFMA column j is FIXED to ray j (the slot of its k-th sample is
// base + sum_i min(c_i, k) + #{i < j: c_i > k}), w * logits accumulates in the MFMA output registers (16 per lane), and
// the eight partial sums meet in LDS when the group is done: one barrier per group, two LDS buffers by group parity
// (waves 0..3 add up and store group g while waves 4..7 already run group g + 1).  Steps where every live ray of the
// group has w == 0 (behind the termination point) skip the gather and the MLP.  x_is_01: the march writer already
// normalised the coordinates (table feed).
// kO: upstream's -O numerics (opt-in) - the table is the fp16 copy and the MLP GEMMs take one fp16 MFMA pass.
template <int K_MT, bool kO = false>
__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_instance_render(
    const float* __restrict__ x, const int32_t* __restrict__ rays, const float* __restrict__ wbuf, int64_t N, int64_t M,
    float bound, const float2* __restrict__ emb, uint32_t emb_bytes, GridDesc G, const float4* __restrict__ packed,
    float* __restrict__ extra_out, int x_is_01, unsigned long long* __restrict__ cursors) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  __shared__ int64_t drawn[2];
  constexpr int K = K_MT * 16;
  constexpr int kStage = (kIns2 + K * 64) / 4;
  constexpr int kWaves = kFieldThreads / 64;
  for (int i = threadIdx.x; i < kStage; i += kFieldThreads) wl[i] = packed[i];
  LevelRec* recs = reinterpret_cast<LevelRec*>(wl + kStage);
  float4* red = wl + kStage + kLevelRecBytes / 16;        // [2][kWaves][K_MT][64] float4
  stage_level_recs(G, recs);
  __syncthreads();

  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15, wave = threadIdx.x >> 6;
  const float rb = 2.0f * bound;
  const float x_add = x_is_01 ? 0.0f : bound, x_div = x_is_01 ? 1.0f : rb;
  // a WORKGROUP per group, drawn dynamically (GroupDraw): thread 0 requests the next group while this one is evaluated
  // and hands it over through LDS behind the barrier every group ends with.  (kGroupDraw == 1 here.)
  const int64_t n_groups = (N + 15) >> 4;
  GroupDraw gd;
  gd.init(cursors, 1);
  const bool all_hashed[4] = {slot_all_hashed(G, 0), slot_all_hashed(G, 1), slot_all_hashed(G, 2), slot_all_hashed(G, 3)};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);

  if (threadIdx.x == 0) drawn[0] = gd.draw();
  __syncthreads();
  int parity = 0;
  for (int64_t grp = drawn[0]; grp < n_groups; grp = drawn[parity ^= 1]) {
    if (threadIdx.x == 0) drawn[parity ^ 1] = gd.draw();
    const int64_t ray = grp * 16 + j;
    const int cnt = ray < N ? rays[ray * 3 + 2] : 0;
    const int64_t S0 = rays[grp * 16 * 3 + 1];             // slot base of the group (offset of its first ray)
    int gtot = cnt, kmax = cnt;                            // group total (to honour a dropped group), longest ray
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      gtot += __shfl_xor(gtot, d, 64);
      kmax = max(kmax, __shfl_xor(kmax, d, 64));
    }
    if (S0 + gtot > M) kmax = 0;
    f32x4 acc[K_MT];
#pragma unroll
    for (int mt = 0; mt < K_MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the sample of (ray j, step k): requested one step of this wave ahead of its use
    auto request = [&](int k, float& w, float& x0, float& x1, float& x2) {
      int below = min(cnt, k);                             // samples of ray i before step k, summed over the group
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) below += __shfl_xor(below, d, 64);
      const unsigned field = (unsigned)(__ballot(k < cnt) & 0xFFFFull);
      const bool active = k < cnt;
      const int64_t slot = active ? S0 + below + __popc(field & ((1u << j) - 1u)) : 0;
      w = active ? wbuf[slot] : 0.0f;
      x0 = x[slot * 3 + 0]; x1 = x[slot * 3 + 1]; x2 = x[slot * 3 + 2];
    };
    float w = 0.f, x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (wave < kmax) request(wave, w, x0, x1, x2);
    for (int k = wave; k < kmax; k += kWaves) {
      float wn = 0.f, xn0 = 0.f, xn1 = 0.f, xn2 = 0.f;
      if (k + kWaves < kmax) request(k + kWaves, wn, xn0, xn1, xn2);
      if (__ballot(w != 0.0f) != 0ull) {                   // else: the whole group is past its termination points
        const float p0 = (x0 + x_add) / x_div, p1 = (x1 + x_add) / x_div, p2 = (x2 + x_add) / x_div;
        f32x4 enc[2];
        {
          Gathered g;
          uint32_t rec_off = (uint32_t)q * 4u * (uint32_t)sizeof(LevelRec);
          asm volatile("" : "+v"(rec_off));
          issue_gathers<kO>(reinterpret_cast<const LevelRec*>(reinterpret_cast<const char*>(recs) + rec_off), all_hashed,
                            rsrc, p0, p1, p2, g);
          __builtin_amdgcn_sched_barrier(0);
          blend<kO>(g, enc[0], enc[1]);
        }
        f32x4 h1[4], h2[4], o[K_MT];
        mlp_layer<4, 2, kO>(wl + kIns0 / 4, lane, enc, h1);
#pragma unroll
        for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
        mlp_layer<4, 4, kO>(wl + kIns1 / 4, lane, h1, h2);
#pragma unroll
        for (int t = 0; t < 4; ++t) h2[t] = relu4(h2[t]);
        mlp_layer<K_MT, 4, kO>(wl + kIns2 / 4, lane, h2, o);
#pragma unroll
        for (int mt = 0; mt < K_MT; ++mt) {
          acc[mt][0] = fmaf(w, o[mt][0], acc[mt][0]); acc[mt][1] = fmaf(w, o[mt][1], acc[mt][1]);
          acc[mt][2] = fmaf(w, o[mt][2], acc[mt][2]); acc[mt][3] = fmaf(w, o[mt][3], acc[mt][3]);
        }
      }
      w = wn; x0 = xn0; x1 = xn1; x2 = xn2;
    }
    float4* mine = red + ((size_t)(parity * kWaves + wave) * K_MT) * 64;
#pragma unroll
    for (int mt = 0; mt < K_MT; ++mt) mine[mt * 64 + lane] = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    __syncthreads();                                       // measured free: 14.5 ms per frame with and without it
    if (threadIdx.x < K_MT * 64 && ray < N) {              // thread t: tile row block t >> 6, lane position t & 63
      const float4* part = red + (size_t)parity * kWaves * K_MT * 64 + threadIdx.x;
      float4 sum = part[0];
#pragma unroll
      for (int wv = 1; wv < kWaves; ++wv) {
        const float4 v = part[(size_t)wv * K_MT * 64];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      const int32_t rid = rays[ray * 3];
      *reinterpret_cast<float4*>(extra_out + (int64_t)rid * K + 16 * (threadIdx.x >> 6) + 4 * q) = sum;
    }
  }
}


// ---- whole-ray rendering with early termination (inference) --------------------------------------------------
// Field evaluation AND alpha compositing in one kernel: one wave owns a 16-ray group of the patch-interleaved
// layout for all its steps, MFMA column j is fixed to ray j, the compositing state (T, colour, opacity, depth)
// lives in the registers of lane (q = 0, j), sigma / rgb never reach HBM, and - the point - a step whose rays
// have all dropped below T_thresh is not evaluated at all: the group stops as soon as its last ray is opaque.
// This is what upstream's alive-ray loop buys on trained (opaque) scenes, without its per-iteration host sync.
// On a transparent scene it evaluates exactly the samples of the two-kernel path.  weights (nullable) receives
// w per sample (0 for skipped samples) for k_instance_render; evaluated[0] += samples of the steps that were evaluated
// (all rays of the group that have a sample at such a step, terminated or not).
// (Round 3 tried the k_instance_render pattern here - a WORKGROUP owns the group, its eight waves take eight consecutive
//  steps, wave 0 composites a round at a time from LDS, one round evaluated speculatively: 1.18x instead of 1.25-1.37x
//  the two-kernel path's time per evaluated sample, but termination at 8-16-step granularity: opaque scene 2.91 vs 0.94
//  ms (10.2 M instead of 0.6 M samples evaluated), half-trained scene 15.3 vs 17.1 ms with the two-kernel path at 13.6.
//  There is no skippable fraction at which it is the best of the three paths - profiles/r03_NOTES.txt 8 - so the
//  wave-owned kernel stays.)
template <bool kO = false>      // kO: upstream's -O numerics (opt-in) - fp16 table copy, one fp16 MFMA pass per MLP GEMM
__global__ void __launch_bounds__(kFieldThreads, kFieldMinWaves) k_nerf_render(
    const float* __restrict__ x, const float* __restrict__ deltas, const int32_t* __restrict__ rays,
    const float* __restrict__ rays_d, int64_t N, int64_t M, float bound, const float2* __restrict__ emb,
    uint32_t emb_bytes, GridDesc G, const float4* __restrict__ packed, float density_scale, float T_thresh,
    float* __restrict__ weights_sum, float* __restrict__ depth, float* __restrict__ image, float* __restrict__ wbuf,
    unsigned long long* __restrict__ evaluated, int x_is_01) {
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  constexpr int kStage = kNerfFloats / 4;
  for (int i = threadIdx.x; i < kStage; i += kFieldThreads) wl[i] = packed[i];
  LevelRec* recs = reinterpret_cast<LevelRec*>(wl + kStage);
  stage_level_recs(G, recs);
  __syncthreads();

  const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
  const float rb = 2.0f * bound;
  const float x_add = x_is_01 ? 0.0f : bound, x_div = x_is_01 ? 1.0f : rb;   // table feed: already normalised
  const bool all_hashed[4] = {slot_all_hashed(G, 0), slot_all_hashed(G, 1), slot_all_hashed(G, 2), slot_all_hashed(G, 3)};
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, (int)emb_bytes, 0x00020000);
  unsigned long long n_eval = 0;
  // dynamic group schedule (GroupDraw): the wave draws its next group(s) while it evaluates the current one
  const int64_t n_groups = (N + 15) >> 4;
  GroupDraw gd;
  gd.init(evaluated + 1, kGroupPools);
  auto draw = [&]() -> int64_t {
    int64_t g = 0;
    if (lane == 0) g = gd.draw();
    return ((int64_t)__builtin_amdgcn_readfirstlane((int)(g >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
  };

  int64_t base = draw(), base_next = draw();      // the next draw is always in flight while a batch is evaluated
  for (int sub = 0;;) {
    if (sub == kGroupDraw) {
      base = base_next;
      base_next = draw();
      sub = 0;
    }
    const int64_t grp = base + sub++;
    if (grp >= n_groups) {
      if (base >= n_groups) break;                 // the XCD's share is used up
      sub = kGroupDraw;                            // the frame ends inside this batch
      continue;
    }
    const int64_t ray = grp * 16 + j;
    const bool has_ray = ray < N;
    const int cnt = has_ray ? rays[ray * 3 + 2] : 0;
    int64_t S = rays[grp * 16 * 3 + 1];
    int gtot = cnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) gtot += __shfl_xor(gtot, d, 64);
    const bool fits = S + gtot <= M;
    const int64_t rc = has_ray ? ray : N - 1;
    float sh[16];
    sh4(rays_d[rc * 3], rays_d[rc * 3 + 1], rays_d[rc * 3 + 2], sh);   // the direction is constant along a ray
    f32x4 cin0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) cin0[ks] = select4(q, sh[4 * ks], sh[4 * ks + 1], sh[4 * ks + 2], sh[4 * ks + 3]);
    float T = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, ws = 0.f, tt = 0.f, dsum = 0.f;   // meaningful in lanes q == 0
    bool done = false;
    // the slot of a step depends on the counts only: the coordinates of step k + 1 are requested while step k is
    // evaluated (they stream from HBM; a step used to start with that round trip)
    float xn0 = 0.f, xn1 = 0.f, xn2 = 0.f;
    if (fits && gtot > 0) {
      const unsigned f0 = (unsigned)(__ballot(0 < cnt) & 0xFFFFull);
      const int64_t m0 = ((f0 >> j) & 1u) ? S + __popc(f0 & ((1u << j) - 1u)) : 0;
      xn0 = x[m0 * 3 + 0]; xn1 = x[m0 * 3 + 1]; xn2 = x[m0 * 3 + 2];
    }
    for (int k = 0; fits; ++k) {
      const unsigned long long bal = __ballot(k < cnt);
      const unsigned field = (unsigned)(bal & 0xFFFFull);
      if (field == 0) break;
      const unsigned donef = (unsigned)(__ballot(done) & 0xFFFFull);
      const unsigned livef = field & ~donef;
      const bool active = (field >> j) & 1u, live = (livef >> j) & 1u;
      const int64_t slot = S + __popc(field & ((1u << j) - 1u));
      S += __popc(field);
      const float xr0 = xn0, xr1 = xn1, xr2 = xn2;
      {
        // only while a ray of the group is alive (alive at k + 1 implies alive at k): an opaque group must not stream
        // the coordinates of the samples it skips
        const unsigned fn = livef != 0 ? (unsigned)(__ballot(k + 1 < cnt) & 0xFFFFull) : 0u;
        if (fn != 0) {
          const int64_t mn = ((fn >> j) & 1u) ? S + __popc(fn & ((1u << j) - 1u)) : 0;
          xn0 = x[mn * 3 + 0]; xn1 = x[mn * 3 + 1]; xn2 = x[mn * 3 + 2];
        }
      }
      if (livef == 0) {                                   // every remaining ray of the group is opaque
        if (!wbuf) break;                                 // nothing left to do for this group
        if (active && q == 0) wbuf[slot] = 0.0f;          // the instance render reads a weight for every sample
        continue;
      }
      // group-level accounting, the same quantity the two-kernel path's compositing counts as NOT skippable: every
      // sample of a step that is evaluated at all (a terminated ray inside a live group still rides through the
      // MFMA tile).  Counting live rays only made infer_mode="auto" compare two different fractions (round-2 advisor).
      if (lane == 0) n_eval += __popc(field);
      const float2 dl = (q == 0 && live) ? reinterpret_cast<const float2*>(deltas)[slot] : make_float2(0.f, 0.f);
      const float x0 = (xr0 + x_add) / x_div, x1 = (xr1 + x_add) / x_div, x2 = (xr2 + x_add) / x_div;
      f32x4 enc[2];
      {
        Gathered g;
        uint32_t rec_off = (uint32_t)q * 4u * (uint32_t)sizeof(LevelRec);
        asm volatile("" : "+v"(rec_off));
        issue_gathers<kO>(reinterpret_cast<const LevelRec*>(reinterpret_cast<const char*>(recs) + rec_off), all_hashed, rsrc,
                          x0, x1, x2, g);
        __builtin_amdgcn_sched_barrier(0);
        blend<kO>(g, enc[0], enc[1]);
      }
      f32x4 h1[4], h2[1];
      mlp_layer<4, 2, kO>(wl + kSig0 / 4, lane, enc, h1);
#pragma unroll
      for (int t = 0; t < 4; ++t) h1[t] = relu4(h1[t]);
      mlp_layer<1, 4, kO>(wl + kSig1 / 4, lane, h1, h2);
      f32x4 cin[2], c1[4], c2[4], o[1];
      cin[0] = cin0;
      cin[1] = h2[0];
      mlp_layer<4, 2, kO>(wl + kCol0 / 4, lane, cin, c1);
#pragma unroll
      for (int t = 0; t < 4; ++t) c1[t] = relu4(c1[t]);
      mlp_layer<4, 4, kO>(wl + kCol1 / 4, lane, c1, c2);
#pragma unroll
      for (int t = 0; t < 4; ++t) c2[t] = relu4(c2[t]);
      mlp_layer<1, 4, kO>(wl + kCol2 / 4, lane, c2, o);
      if (q == 0 && active) {
        float w = 0.0f;
        if (live) {
          const float sg = __expf(h2[0][0]) * density_scale;
          const float alpha = 1.0f - expf(-sg * dl.x);
          w = alpha * T;
          cr += w * __frcp_rn(1.0f + __expf(-o[0][0]));
          cg += w * __frcp_rn(1.0f + __expf(-o[0][1]));
          cb += w * __frcp_rn(1.0f + __expf(-o[0][2]));
          tt += dl.y;
          dsum += w * tt;
          ws += w;
          T *= 1.0f - alpha;
          if (T < T_thresh) done = true;
        }
        if (wbuf) wbuf[slot] = w;
      }
    }
    if (q == 0 && has_ray) {
      const int32_t rid = rays[ray * 3];
      weights_sum[rid] = ws; depth[rid] = dsum;
      image[rid * 3] = cr; image[rid * 3 + 1] = cg; image[rid * 3 + 2] = cb;
    }
  }
  if (lane == 0 && n_eval) atomicAdd(evaluated, n_eval);
}

// ---- host-side packing into fragment order ------------------------------------------------------
// W is [n_out, n_in] row-major.  kidx(ks, q) -> input column (or -1 for a zero slot).
static inline uint16_t bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// bf16 split layout: [mt][step][hi|lo][lane][8]; kidx(step*8 + e style) is expressed through the same
// k-step functions as the fp32 layout: slot e of step s is fp32 k-step 4*(2s + (e>>2)) + (e&3).
template <class KIdx>
static void pack_section_bf16(float* dst_f, const float* W, int n_out, int n_in, int n_mt, int n_ks, KIdx kidx) {
  uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
  const int n_s = n_ks / 8;
  for (int mt = 0; mt < n_mt; ++mt)
    for (int st = 0; st < n_s; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int row = 16 * mt + (lane & 15);
          const int col = kidx(4 * (2 * st + (e >> 2)) + (e & 3), lane >> 4);
          float v = 0.f;
          if (row < n_out && col >= 0 && col < n_in) v = W[(size_t)row * n_in + col];
          const uint16_t hi = bf16_rne(v);
          const uint16_t lo = bf16_rne(v - bf16_to_f32(hi));
          dst[((((size_t)(mt * n_s + st) * 2 + 0) * 64 + lane) * 8) + e] = hi;
          dst[((((size_t)(mt * n_s + st) * 2 + 1) * 64 + lane) * 8) + e] = lo;
        }
}

// fp16 variant of the same layout for the opt-in single-pass MLP: the head slot holds the weight rounded to fp16, the
// remainder slot is unused (zero).
template <class KIdx>
static void pack_section_f16(float* dst_f, const float* W, int n_out, int n_in, int n_mt, int n_ks, KIdx kidx) {
  uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
  const int n_s = n_ks / 8;
  for (int mt = 0; mt < n_mt; ++mt)
    for (int st = 0; st < n_s; ++st)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int row = 16 * mt + (lane & 15);
          const int col = kidx(4 * (2 * st + (e >> 2)) + (e & 3), lane >> 4);
          float v = 0.f;
          if (row < n_out && col >= 0 && col < n_in) v = W[(size_t)row * n_in + col];
          const _Float16 h = (_Float16)v;                 // round to nearest even
          uint16_t bits;
          memcpy(&bits, &h, 2);
          dst[((((size_t)(mt * n_s + st) * 2 + 0) * 64 + lane) * 8) + e] = bits;
          dst[((((size_t)(mt * n_s + st) * 2 + 1) * 64 + lane) * 8) + e] = 0;
        }
}

template <class KIdx>
static void pack_section_f32(float* dst, const float* W, int n_out, int n_in, int n_mt, int n_ks, KIdx kidx) {
  for (int mt = 0; mt < n_mt; ++mt)
    for (int ks = 0; ks < n_ks; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = 16 * mt + (lane & 15);
        const int col = kidx(ks, lane >> 4);
        float v = 0.f;
        if (row < n_out && col >= 0 && col < n_in) v = W[(size_t)row * n_in + col];
        dst[frag_pos(mt, ks, lane, n_ks)] = v;
      }
}
template <class KIdx>
static void pack_section(float* dst, const float* W, int n_out, int n_in, int n_mt, int n_ks, KIdx kidx) {
#if INR_MLP_FP32
  pack_section_f32(dst, W, n_out, n_in, n_mt, n_ks, kidx);
#else
  pack_section_bf16(dst, W, n_out, n_in, n_mt, n_ks, kidx);
#endif
}
static int kidx_enc(int ks, int q) { return 16 * (ks >> 2) + 4 * q + (ks & 3); }      // encoder features
static int kidx_hidden(int ks, int q) { return 16 * (ks >> 2) + 4 * q + (ks & 3); }   // previous D registers
static int kidx_color_in(int ks, int q) {
  if (ks < 4) return 4 * ks + q;                 // sh component
  const int row = 4 * q + (ks - 4);              // sigma-net output row feeding this slot
  return row >= 1 ? 16 + row - 1 : -1;           // row 0 is the raw density: not an input
}

// Persistent grid.  MEASURED (tools/field_probe.py, 37 M samples, MI355X): the kernel is fastest with
// 8-12 waves per CU - 4: 10.3 ms, 6: 8.3, 8: 6.8, 12: 6.8, 16: 7.1-7.2 - because every resident wave adds a
// tile's worth of table lines to the L1/L2 working set; beyond ~2 waves/SIMD the extra latency hiding is
// worth less than the cache it costs (a per-wave double-buffered pipeline at 8 waves/CU measures 6.95 ms
// for the same reason).  So: ONE 512-thread workgroup per CU, all co-resident, no tail round.
constexpr int kFieldBlocksPerCU = 1;

template <class Kern>
static int grid_for(Kern kern, size_t lds_bytes, int64_t n_tiles) {
  // occupancy of (kernel, LDS size) is a constant of the build: ask the runtime once per pair
  struct Entry { const void* kern; size_t lds; int fit; };
  static Entry cache[64];
  static int n_cached = 0;
  const int cus = cu_count();
  int fit = 0;
  for (int i = 0; i < n_cached; ++i)
    if (cache[i].kern == (const void*)kern && cache[i].lds == lds_bytes) fit = cache[i].fit;
  if (fit < 1) {
    fit = kFieldBlocksPerCU;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, kFieldThreads, lds_bytes) != hipSuccess || fit < 1) fit = 1;
    if (n_cached < 64) cache[n_cached++] = Entry{(const void*)kern, lds_bytes, fit};
  }
  // (two workgroups per CU for small launches - training batches, ~6 tiles per wave - measured in round 3: no change,
  //  92.2 vs 90.9 us per 209 k samples; those launches are bound by the L2 fills of a cold table, not by latency)
  const int per_cu = std::min(fit, kFieldBlocksPerCU);
  const int64_t want = (n_tiles + kFieldThreads / 64 - 1) / (kFieldThreads / 64);
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)cus * per_cu));
}

// Cursors of the hybrid schedule (TileWalk): a ring of kStealRing sets per device, the next one zeroed on the launch's
// stream right before the launch.  Launches on different streams (FramePipeline) never share a set: every set carries
// the event of the launch that last used it, and a launch that is handed a set whose previous user has not finished
// (more than kStealRing field kernels pending at once) makes its stream wait for that event first (round-3 advisor: the
// set used to be re-zeroed under the running kernel - tiles skipped or duplicated without an error).
constexpr int kStealRing = 16;
struct StealSet {
  unsigned long long* cursors = nullptr;
  int dev = -1, slot = -1;
};
struct StealRing {
  std::atomic<unsigned long long*> base{nullptr};
  hipEvent_t ev[kStealRing] = {};
  bool used[kStealRing] = {};
};
static StealRing g_steal[64];
static std::mutex g_steal_mu;
static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
// Never called while `s` is capturing (round-4 advisor): a graph would bake ONE ring slot and its memset into every
// replay, with no event protection - an eager launch on another stream that draws the same slot could re-zero the cursors
// under the running replay (tiles skipped or duplicated, no error).  Captured launches take the static deal instead.
static StealSet steal_cursors(hipStream_t s) {
  static std::atomic<unsigned> turn{0};
  StealSet out;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return out;
  StealRing& R = g_steal[dev];
  if (!R.base.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g_steal_mu);
    if (!R.base.load(std::memory_order_relaxed)) {
      void* p = nullptr;
      if (hipMalloc(&p, (size_t)kStealRing * kStealOwners * sizeof(unsigned long long)) != hipSuccess) {
        set_error("hybrid schedule: hipMalloc of the cursor ring failed");
        return out;
      }
      R.base.store((unsigned long long*)p, std::memory_order_release);
    }
  }
  const int slot = (int)(turn.fetch_add(1) % kStealRing);
  unsigned long long* p = R.base.load(std::memory_order_acquire) + (size_t)slot * kStealOwners;
  {
    std::lock_guard<std::mutex> lock(g_steal_mu);
    if (R.used[slot] && !stream_is_capturing(s) && hipEventQuery(R.ev[slot]) == hipErrorNotReady &&
        hipStreamWaitEvent(s, R.ev[slot], 0) != hipSuccess) {
      set_error("hybrid schedule: hipStreamWaitEvent failed");
      return out;
    }
  }
  if (hipMemsetAsync(p, 0, kStealOwners * sizeof(unsigned long long), s) != hipSuccess) {
    set_error("hybrid schedule: hipMemsetAsync failed");
    return out;
  }
  out.cursors = p;
  out.dev = dev;
  out.slot = slot;
  return out;
}
// after the launch that draws from the set: its completion event (not inside a stream capture: a captured event says
// nothing about a replay - captured steps are small launches and never take the hybrid schedule anyway)
static void steal_release(const StealSet& st, hipStream_t s) {
  if (!st.cursors || stream_is_capturing(s)) return;
  std::lock_guard<std::mutex> lock(g_steal_mu);
  StealRing& R = g_steal[st.dev];
  if (!R.ev[st.slot] && hipEventCreateWithFlags(&R.ev[st.slot], hipEventDisableTiming) != hipSuccess) return;
  R.used[st.slot] = hipEventRecord(R.ev[st.slot], s) == hipSuccess;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: once per (kernel call site, device), not once per
// process (round-3 advisor: a process that trained on a second GPU launched there without the attribute)
struct AttrOnce {
  std::atomic<bool> done[64] = {};
  template <typename K>
  bool ensure(K kern, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (done[dev].load(std::memory_order_acquire)) return true;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    done[dev].store(true, std::memory_order_release);
    return true;
  }
};

}  // namespace inr

using namespace inr;

extern "C" {

int64_t inr_nerf_packed_floats(void) { return kNerfFloats; }

int inr_nerf_pack_weights(const float* sigma_w0, const float* sigma_w1, const float* color_w0, const float* color_w1,
                          const float* color_w2, float* packed) {
  INR_REQUIRE(sigma_w0 && sigma_w1 && color_w0 && color_w1 && color_w2 && packed, "null pointer");
  pack_section(packed + kSig0, sigma_w0, 64, 32, 4, 8, kidx_enc);
  pack_section(packed + kSig1, sigma_w1, 16, 64, 1, 16, kidx_hidden);
  pack_section(packed + kCol0, color_w0, 64, 31, 4, 8, kidx_color_in);
  pack_section(packed + kCol1, color_w1, 64, 64, 4, 16, kidx_hidden);
  pack_section(packed + kCol2, color_w2, 3, 64, 1, 16, kidx_hidden);
  return INR_OK;
}

int inr_nerf_pack_weights_f16(const float* sigma_w0, const float* sigma_w1, const float* color_w0, const float* color_w1,
                              const float* color_w2, float* packed) {
  INR_REQUIRE(sigma_w0 && sigma_w1 && color_w0 && color_w1 && color_w2 && packed, "null pointer");
  pack_section_f16(packed + kSig0, sigma_w0, 64, 32, 4, 8, kidx_enc);
  pack_section_f16(packed + kSig1, sigma_w1, 16, 64, 1, 16, kidx_hidden);
  pack_section_f16(packed + kCol0, color_w0, 64, 31, 4, 8, kidx_color_in);
  pack_section_f16(packed + kCol1, color_w1, 64, 64, 4, 16, kidx_hidden);
  pack_section_f16(packed + kCol2, color_w2, 3, 64, 1, 16, kidx_hidden);
  return INR_OK;
}

int64_t inr_instance_packed_floats(int32_t K) {
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  return kIns2 + K * 64;
}

int inr_instance_pack_weights(const float* w0, const float* w1, const float* w2, int32_t K, float* packed) {
  INR_REQUIRE(w0 && w1 && w2 && packed, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  pack_section(packed + kIns0, w0, 64, 32, 4, 8, kidx_enc);
  pack_section(packed + kIns1, w1, 64, 64, 4, 16, kidx_hidden);
  pack_section(packed + kIns2, w2, K, 64, K / 16, 16, kidx_hidden);
  return INR_OK;
}

int inr_nerf_forward_fast(const float* x, const float* d, int64_t M, const int32_t* n_samples_dev, float bound,
                          const void* embeddings_half, const inr_grid_desc* desc, const float* packed_f16,
                          float density_scale, float* sigma, float* rgb, inr_stream_t s) {
#if INR_MLP_FP32
  (void)x; (void)d; (void)M; (void)n_samples_dev; (void)bound; (void)embeddings_half; (void)desc; (void)packed_f16;
  (void)density_scale; (void)sigma; (void)rgb; (void)s;
  set_error("nerf_forward_fast: not available in the exact-fp32 build");
  return INR_EINVAL;
#else
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && d && embeddings_half && packed_f16 && sigma && rgb, "null pointer");
  INR_REQUIRE(((uintptr_t)embeddings_half & 3) == 0 && ((uintptr_t)packed_f16 & 15) == 0, "embeddings/packed misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 4ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes;
  const int grid = grid_for(k_nerf_fwd<true, false, 0, true, true>, lds, (M + 15) / 16);
  k_nerf_fwd<true, false, 0, true, true><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      x, d, M, n_samples_dev, bound, reinterpret_cast<const float2*>(embeddings_half), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed_f16), density_scale, sigma, rgb, nullptr, nullptr, nullptr, NerfSave{}, nullptr);
  return check_launch("nerf_forward_fast");
#endif
}

int inr_nerf_forward(const float* x, const float* d, int64_t M, const int32_t* n_samples_dev, float bound,
                     const float* embeddings, const inr_grid_desc* desc, const float* packed, float density_scale,
                     float* sigma, float* rgb, float* geo_feat, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && packed && sigma, "null pointer");
  INR_REQUIRE(!rgb || d, "rgb requested without view directions");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0, "embeddings/packed misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  if (M == 0) return INR_OK;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const float4* p = reinterpret_cast<const float4*>(packed);
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t emb_bytes = (uint32_t)emb_bytes64;
  const int64_t n_tiles = (M + 15) / 16;
  // frames and occupancy sweeps (four rounds of eight 1024-tile chunks and more): hybrid schedule, see TileWalk
  StealSet steal_set;
  if ((n_tiles >> (kXcdChunkLog2 + 3)) >= 4 && !stream_is_capturing(as_stream(s))) {   // captured: static deal, see steal_cursors
    steal_set = steal_cursors(as_stream(s));
    if (!steal_set.cursors) return INR_ELAUNCH;
  }
  unsigned long long* steal = steal_set.cursors;
  if (rgb) {
    const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes;
    const int grid = grid_for(k_nerf_fwd<true>, lds, n_tiles);
    k_nerf_fwd<true><<<grid, kFieldThreads, lds, as_stream(s)>>>(x, d, M, n_samples_dev, bound, e, emb_bytes, G, p,
                                                      density_scale, sigma, rgb, geo_feat, nullptr, nullptr, NerfSave{}, steal);
  } else {
    const size_t lds = kCol0 * sizeof(float) + kLevelRecBytes;
    const int grid = grid_for(k_nerf_fwd<false>, lds, n_tiles);
    k_nerf_fwd<false><<<grid, kFieldThreads, lds, as_stream(s)>>>(x, d, M, n_samples_dev, bound, e, emb_bytes, G, p,
                                                       density_scale, sigma, nullptr, geo_feat, nullptr, nullptr, NerfSave{}, steal);
  }
  steal_release(steal_set, as_stream(s));
  return check_launch("nerf_forward");
}

static int nerf_forward_table_impl(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                                   const void* embeddings, bool half, bool fast, const inr_grid_desc* desc,
                                   const float* packed, float density_scale, float* sigma, float* rgb, inr_stream_t s);

int inr_nerf_forward_table(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                           const float* embeddings, const inr_grid_desc* desc, const float* packed, float density_scale,
                           float* sigma, float* rgb, inr_stream_t s) {
  return nerf_forward_table_impl(x01, ray_ids, sh_table_q, M, bound, embeddings, false, false, desc, packed, density_scale,
                                 sigma, rgb, s);
}

int inr_nerf_forward_table_half(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                                const void* embeddings_half, const inr_grid_desc* desc, const float* packed,
                                float density_scale, float* sigma, float* rgb, inr_stream_t s) {
  return nerf_forward_table_impl(x01, ray_ids, sh_table_q, M, bound, embeddings_half, true, false, desc, packed,
                                 density_scale, sigma, rgb, s);
}

int inr_nerf_forward_table_fast(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                                const void* embeddings, int32_t table_is_half, const inr_grid_desc* desc,
                                const float* packed, float density_scale, float* sigma, float* rgb, inr_stream_t s) {
#if INR_MLP_FP32
  (void)x01; (void)ray_ids; (void)sh_table_q; (void)M; (void)bound; (void)embeddings; (void)table_is_half; (void)desc;
  (void)packed; (void)density_scale; (void)sigma; (void)rgb; (void)s;
  set_error("nerf_forward_table_fast: not available in the exact-fp32 build");
  return INR_EINVAL;
#else
  return nerf_forward_table_impl(x01, ray_ids, sh_table_q, M, bound, embeddings, table_is_half != 0, true, desc, packed,
                                 density_scale, sigma, rgb, s);
#endif
}

extern "C++" {
template <bool kHalf, bool kFast>
static int launch_nerf_table(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                             const void* embeddings, uint32_t emb_bytes, const GridDesc& G, const float* packed,
                             float density_scale, float* sigma, float* rgb, size_t lds, inr_stream_t s) {
  const int grid = grid_for(k_nerf_fwd<true, true, 0, kHalf, kFast>, lds, (M + 15) / 16);
  // frames (four rounds of eight 1024-tile chunks and more) take the hybrid schedule: its cursors, zeroed on this stream
  StealSet steal_set;
  if ((((M + 15) / 16) >> (kXcdChunkLog2 + 3)) >= 4 && !stream_is_capturing(as_stream(s))) {
    steal_set = steal_cursors(as_stream(s));
    if (!steal_set.cursors) return INR_ELAUNCH;
  }
  unsigned long long* steal = steal_set.cursors;
  k_nerf_fwd<true, true, 0, kHalf, kFast><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      x01, nullptr, M, nullptr, bound, reinterpret_cast<const float2*>(embeddings), emb_bytes, G,
      reinterpret_cast<const float4*>(packed), density_scale, sigma, rgb, nullptr, ray_ids,
      reinterpret_cast<const float4*>(sh_table_q), NerfSave{}, steal);
  steal_release(steal_set, as_stream(s));
  return check_launch("nerf_forward_table");
}
}  // extern "C++"

static int nerf_forward_table_impl(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                                   const void* embeddings, bool half, bool fast, const inr_grid_desc* desc,
                                   const float* packed, float density_scale, float* sigma, float* rgb, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x01 && ray_ids && sh_table_q && embeddings && packed && sigma && rgb, "null pointer");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)sh_table_q & 15) == 0,
              "embeddings/packed/sh_table_q misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = std::max(kNerfFloats * sizeof(float) + kLevelRecBytes, (size_t)g_field_lds_min);
  const uint32_t eb = (uint32_t)(half ? emb_bytes64 / 2 : emb_bytes64);
#define INR_TABLE_LAUNCH(H, F) \
  launch_nerf_table<H, F>(x01, ray_ids, sh_table_q, M, bound, embeddings, eb, G, packed, density_scale, sigma, rgb, lds, s)
#if INR_MLP_FP32
  (void)fast;
  return half ? INR_TABLE_LAUNCH(true, false) : INR_TABLE_LAUNCH(false, false);
#else
  if (fast) return half ? INR_TABLE_LAUNCH(true, true) : INR_TABLE_LAUNCH(false, true);
  return half ? INR_TABLE_LAUNCH(true, false) : INR_TABLE_LAUNCH(false, false);
#endif
#undef INR_TABLE_LAUNCH
}

// ---- sliced frame path: level-major pre-pass over the three finest levels + fused kernel on the other thirteen ------
int64_t inr_nerf_forward_table_sliced_workspace_bytes(int64_t M) {
  return M < 0 ? -1 : kSliceLevels * ((M + 31) & ~(int64_t)31) * (int64_t)sizeof(float2);
}

int inr_nerf_forward_table_sliced(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M, float bound,
                                  const float* embeddings, const inr_grid_desc* desc, const float* packed,
                                  float density_scale, float* sigma, float* rgb, float* fine_ws, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x01 && ray_ids && sh_table_q && embeddings && packed && sigma && rgb && fine_ws, "null pointer");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)sh_table_q & 15) == 0 &&
              ((uintptr_t)fine_ws & 7) == 0, "embeddings/packed/sh_table_q/fine_ws misaligned");
  INR_REQUIRE(desc->num_levels == 16, "the sliced frame path needs the 16-level table");
  for (int l = 8; l < 16; ++l) INR_REQUIRE(desc->hashed[l], "the sliced frame path needs hashed fine levels (8..15)");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t eb = (uint32_t)emb_bytes64;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const size_t lds = std::max(kNerfFloats * sizeof(float) + kLevelRecBytes, (size_t)g_field_lds_min);
  hipStream_t st = as_stream(s);
  static int fit = 0;                      // resident pre-pass workgroups per CU (a constant of the build)
  if (fit < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_grid_fine_slices, kSliceThreads, 0) != hipSuccess || fit < 1))
    fit = 4;
  // pre-pass: a full house of persistent workgroups per level; dispatch order (x fast, level slow) is the temporal order.
  // (Measured and not kept, profiles/r05_NOTES.txt 3: the frame in chunks with the pre-pass of chunk k+1 on a side stream
  //  beside the fused kernel of chunk k - the fused kernel's traffic then walks through the L2 the pre-pass lives on:
  //  27.3-27.9 ms against 27.5 ms one after the other, 4.2-4.4 against 4.07 ms with growing steps.)
  const int64_t per_wg = 32 * kSliceTilesPerIter * (kSliceThreads / 64);
  const int gx = (int)std::max<int64_t>(1, std::min<int64_t>((M + per_wg - 1) / per_wg, (int64_t)cu_count() * fit));
  k_grid_fine_slices<<<dim3(gx, kSliceLevels), kSliceThreads, 0, st>>>(x01, M, e, eb, G, fine_ws);
  const int grid = grid_for(k_nerf_fwd<true, true, 0, false, false, true>, lds, (M + 15) / 16);
  StealSet steal_set;
  if ((((M + 15) / 16) >> (kXcdChunkLog2 + 3)) >= 4 && !stream_is_capturing(st)) {
    steal_set = steal_cursors(st);
    if (!steal_set.cursors) return INR_ELAUNCH;
  }
  k_nerf_fwd<true, true, 0, false, false, true><<<grid, kFieldThreads, lds, st>>>(
      x01, fine_ws, M, nullptr, bound, e, eb, G, reinterpret_cast<const float4*>(packed), density_scale, sigma, rgb, nullptr,
      ray_ids, reinterpret_cast<const float4*>(sh_table_q), NerfSave{}, steal_set.cursors);
  steal_release(steal_set, st);
  return check_launch("nerf_forward_table_sliced");
}

int inr_nerf_forward_dirs(const float* x, int64_t M, float bound, const float* embeddings, const inr_grid_desc* desc,
                          const float* packed, const float* sh_dirs, int32_t n_dirs, float* out, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && packed && sh_dirs && out, "null pointer");
  INR_REQUIRE(n_dirs >= 1 && n_dirs <= kMaxExtractDirs, "1..8 view directions");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0,
              "embeddings/packed/out misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes + kMaxExtractDirs * 16 * sizeof(float);
  const int grid = grid_for(k_nerf_fwd_dirs<false>, lds, (M + 15) / 16);
  k_nerf_fwd_dirs<false><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      x, M, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed), sh_dirs, n_dirs, reinterpret_cast<float4*>(out), LatticeDesc{}, -INFINITY);
  return check_launch("nerf_forward_dirs");
}

int inr_nerf_forward_lattice(const float* ax_w, const float* ax_l, const float* ax_h, int32_t W, int32_t L, int32_t H,
                             float bound, const float* embeddings, const inr_grid_desc* desc, const float* packed,
                             const float* sh_dirs, int32_t n_dirs, float logit_min, float* out, inr_stream_t s) {
  INR_REQUIRE(W >= 0 && L >= 0 && H >= 0 && desc, "bad argument");
  if ((int64_t)W * L * H == 0) return INR_OK;
  INR_REQUIRE(ax_w && ax_l && ax_h && embeddings && packed && sh_dirs && out, "null pointer");
  INR_REQUIRE(n_dirs >= 1 && n_dirs <= kMaxExtractDirs, "1..8 view directions");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)out & 15) == 0,
              "embeddings/packed/out misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  LatticeDesc lat{ax_w, ax_l, ax_h, W, L, H, (W + 15) / 16 * 16};
  const int64_t M = (int64_t)lat.Wp * L * H;           // traversal length: runs of 16 along w, padded per (il, ih)
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes + kMaxExtractDirs * 16 * sizeof(float);
  const int grid = grid_for(k_nerf_fwd_dirs<true>, lds, (M + 15) / 16);
  k_nerf_fwd_dirs<true><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      nullptr, M, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed), sh_dirs, n_dirs, reinterpret_cast<float4*>(out), lat, logit_min);
  return check_launch("nerf_forward_lattice");
}

int inr_instance_forward(const float* x, int64_t M, const int32_t* n_samples_dev, float bound, const float* embeddings,
                         const inr_grid_desc* desc, const float* packed, int32_t K, float* logits, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && packed && logits, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)logits & 15) == 0,
              "embeddings/packed/logits misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  if (M == 0) return INR_OK;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const float4* p = reinterpret_cast<const float4*>(packed);
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t eb = (uint32_t)emb_bytes64;
  const size_t lds = (size_t)(kIns2 + K * 64) * sizeof(float) + kLevelRecBytes;
  const int64_t n_tiles = (M + 15) / 16;
  hipStream_t st = as_stream(s);
  switch (K / 16) {
    case 1: k_instance_fwd<1><<<grid_for(k_instance_fwd<1>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, nullptr, nullptr, nullptr); break;
    case 2: k_instance_fwd<2><<<grid_for(k_instance_fwd<2>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, nullptr, nullptr, nullptr); break;
    case 3: k_instance_fwd<3><<<grid_for(k_instance_fwd<3>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, nullptr, nullptr, nullptr); break;
    default: k_instance_fwd<4><<<grid_for(k_instance_fwd<4>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, nullptr, nullptr, nullptr); break;
  }
  return check_launch("instance_forward");
}

int64_t inr_instance_bwd_packed_floats(void) { return kBwdFloats; }

int inr_instance_pack_weights_device(const float* w0, const float* w1, const float* w2, int32_t K, float* packed_fwd,
                                     float* packed_bwd, inr_stream_t s) {
  INR_REQUIRE(w0 && w1 && w2 && packed_fwd && packed_bwd, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(((uintptr_t)packed_fwd & 15) == 0 && ((uintptr_t)packed_bwd & 15) == 0, "packed buffers misaligned");
  PackJobs jobs;
  //           W   rows cols  n_out n_in  mt      steps T  offset
  jobs.j[0] = {w0, 64, 32, 64, 32, 4, 1, 0, kIns0, 0, 0, 0, 0};
  jobs.j[1] = {w1, 64, 64, 64, 64, 4, 2, 0, kIns1, 0, 0, 0, 0};
  jobs.j[2] = {w2, K, 64, K, 64, K / 16, 2, 0, kIns2, 0, 0, 0, 0};
  jobs.j[3] = {w2, K, 64, 64, K, 4, 2, 1, kBwd2, 0, 0, 0, 1};
  jobs.j[4] = {w1, 64, 64, 64, 64, 4, 2, 1, kBwd1, 0, 0, 0, 1};
  jobs.j[5] = {w0, 64, 32, 32, 64, 2, 2, 1, kBwd0, 0, 0, 0, 1};
  k_pack_weights<<<dim3(8, 6), 256, 0, as_stream(s)>>>(jobs, packed_fwd, packed_bwd);
  return check_launch("instance_pack_weights_device");
}

int inr_instance_forward_train(const float* x, int64_t M, float bound, const float* embeddings, const inr_grid_desc* desc,
                               const float* packed, int32_t K, float* logits, float* enc, float* h1, float* h2,
                               inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && packed && logits && enc && h1 && h2, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 &&
                  (((uintptr_t)packed | (uintptr_t)logits | (uintptr_t)enc | (uintptr_t)h1 | (uintptr_t)h2) & 15) == 0,
              "embeddings/packed/outputs misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const float4* p = reinterpret_cast<const float4*>(packed);
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t eb = (uint32_t)emb_bytes64;
  const size_t lds = (size_t)(kIns2 + K * 64) * sizeof(float) + kLevelRecBytes;
  const int64_t n_tiles = (M + 15) / 16;
  hipStream_t st = as_stream(s);
  switch (K / 16) {
    case 1: k_instance_fwd<1, 1><<<grid_for(k_instance_fwd<1, 1>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, nullptr, bound, e, eb, G, p, logits, enc, h1, h2); break;
    case 2: k_instance_fwd<2, 1><<<grid_for(k_instance_fwd<2, 1>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, nullptr, bound, e, eb, G, p, logits, enc, h1, h2); break;
    case 3: k_instance_fwd<3, 1><<<grid_for(k_instance_fwd<3, 1>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, nullptr, bound, e, eb, G, p, logits, enc, h1, h2); break;
    default: k_instance_fwd<4, 1><<<grid_for(k_instance_fwd<4, 1>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, nullptr, bound, e, eb, G, p, logits, enc, h1, h2); break;
  }
  return check_launch("instance_forward_train");
}

int inr_instance_forward_enc(const float* x, int64_t M, const int32_t* n_samples_dev, float bound, const float* embeddings,
                             const inr_grid_desc* desc, const float* packed, int32_t K, float* logits, float* enc,
                             inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && embeddings && packed && logits && enc, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && (((uintptr_t)packed | (uintptr_t)logits | (uintptr_t)enc) & 15) == 0,
              "embeddings/packed/outputs misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const float4* p = reinterpret_cast<const float4*>(packed);
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t eb = (uint32_t)emb_bytes64;
  const size_t lds = (size_t)(kIns2 + K * 64) * sizeof(float) + kLevelRecBytes;
  const int64_t n_tiles = (M + 15) / 16;
  hipStream_t st = as_stream(s);
  switch (K / 16) {
    case 1: k_instance_fwd<1, 2><<<grid_for(k_instance_fwd<1, 2>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, enc, nullptr, nullptr); break;
    case 2: k_instance_fwd<2, 2><<<grid_for(k_instance_fwd<2, 2>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, enc, nullptr, nullptr); break;
    case 3: k_instance_fwd<3, 2><<<grid_for(k_instance_fwd<3, 2>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, enc, nullptr, nullptr); break;
    default: k_instance_fwd<4, 2><<<grid_for(k_instance_fwd<4, 2>, lds, n_tiles), kFieldThreads, lds, st>>>(x, M, n_samples_dev, bound, e, eb, G, p, logits, enc, nullptr, nullptr); break;
  }
  return check_launch("instance_forward_enc");
}

static size_t head_bwd_lds() { return (size_t)(kHbFwdFloats + kBwdFloats + kHbStageFloats) * sizeof(float); }

int64_t inr_instance_head_workspace_bytes(void) {
  return (int64_t)cu_count() * kFieldBlocksPerCU * kHbAccTiles * 64 * (int64_t)sizeof(float4);
}

int inr_instance_head_backward(const float* enc, const float* weights, const int32_t* sample_ray, const float* grad_pix,
                               int32_t K, int64_t N, int64_t M, const int32_t* n_samples_dev, const float* scale_a,
                               const float* scale_b, const float* packed_fwd, const float* packed_bwd, float* grad_enc,
                               void* workspace, float* grad_w0, float* grad_w1, float* grad_w2, float* zero_buf,
                               int64_t zero_floats, inr_stream_t s) {
  INR_REQUIRE(zero_floats >= 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) == 0), "zero_buf must be 16-byte aligned");
  INR_REQUIRE(M >= 0 && N >= 0, "negative size");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(grad_w0 && grad_w1 && grad_w2 && workspace && packed_fwd && packed_bwd, "null pointer");
  INR_REQUIRE(M == 0 || (enc && weights && sample_ray && grad_pix && grad_enc), "null sample arrays");
  INR_REQUIRE((((uintptr_t)enc | (uintptr_t)grad_pix | (uintptr_t)packed_fwd | (uintptr_t)packed_bwd | (uintptr_t)grad_enc |
                (uintptr_t)workspace) & 15) == 0, "arrays must be 16-byte aligned");
  hipStream_t st = as_stream(s);
  const size_t lds = head_bwd_lds();
  static AttrOnce attr;     // > 64 KB of dynamic LDS must be allowed explicitly, on every device
  if (!attr.ensure(k_instance_head_bwd, lds)) {
    set_error("instance_head_backward: %zu bytes of LDS refused", lds);
    return INR_ELAUNCH;
  }
  const int64_t n_tiles = (M + 15) / 16;
  const int grid = M == 0 ? 1 : grid_for(k_instance_head_bwd, lds, n_tiles);
  k_instance_head_bwd<<<grid, kFieldThreads, lds, st>>>(enc, weights, sample_ray, grad_pix, K, M, n_samples_dev,
                                                        scale_a, scale_b, reinterpret_cast<const float4*>(packed_fwd),
                                                        reinterpret_cast<const float4*>(packed_bwd), grad_enc,
                                                        reinterpret_cast<float4*>(workspace), zero_buf, zero_floats);
  k_head_wgrad_reduce<<<kHbAccTiles * 256 / 64, 1024, 0, st>>>(reinterpret_cast<const float*>(workspace), grid, K, grad_w0,
                                                               grad_w1, grad_w2);
  return check_launch("instance_head_backward");
}

int inr_instance_backward(const float* grad_logits, int32_t K, const float* h1, const float* h2, int64_t M,
                          const float* packed_bwd, float* grad_z2, float* grad_z1, float* grad_enc, inr_stream_t s) {
  INR_REQUIRE(M >= 0, "negative M");
  if (M == 0) return INR_OK;
  INR_REQUIRE(grad_logits && h1 && h2 && packed_bwd && grad_z2 && grad_z1 && grad_enc, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE((((uintptr_t)grad_logits | (uintptr_t)h1 | (uintptr_t)h2 | (uintptr_t)packed_bwd | (uintptr_t)grad_z2 |
                (uintptr_t)grad_z1 | (uintptr_t)grad_enc) & 15) == 0, "arrays must be 16-byte aligned");
  const size_t lds = (size_t)kBwdFloats * sizeof(float);
  const int64_t n_tiles = (M + 15) / 16;
  k_instance_bwd<<<grid_for(k_instance_bwd, lds, n_tiles), kFieldThreads, lds, as_stream(s)>>>(
      grad_logits, K, h1, h2, M, reinterpret_cast<const float4*>(packed_bwd), grad_z2, grad_z1, grad_enc);
  return check_launch("instance_backward");
}

int64_t inr_nerf_bwd_packed_floats(void) { return kNerfBwdFloats; }

int inr_nerf_pack_weights_device(const float* sigma_w0, const float* sigma_w1, const float* color_w0,
                                 const float* color_w1, const float* color_w2, float* packed_fwd, float* packed_bwd,
                                 inr_stream_t s) {
  INR_REQUIRE(sigma_w0 && sigma_w1 && color_w0 && color_w1 && color_w2 && packed_fwd && packed_bwd, "null pointer");
  INR_REQUIRE(((uintptr_t)packed_fwd & 15) == 0 && ((uintptr_t)packed_bwd & 15) == 0, "packed buffers misaligned");
  PackJobs jobs;
  //           W         rows cols n_out n_in mt steps T  offset kmode row_min shift bwd
  jobs.j[0] = {sigma_w0, 64, 32, 64, 32, 4, 1, 0, kSig0, 0, 0, 0, 0};
  jobs.j[1] = {sigma_w1, 16, 64, 16, 64, 1, 2, 0, kSig1, 0, 0, 0, 0};
  jobs.j[2] = {color_w0, 64, 31, 64, 31, 4, 1, 0, kCol0, 1, 0, 0, 0};
  jobs.j[3] = {color_w1, 64, 64, 64, 64, 4, 2, 0, kCol1, 0, 0, 0, 0};
  jobs.j[4] = {color_w2, 3, 64, 3, 64, 1, 2, 0, kCol2, 0, 0, 0, 0};
  jobs.j[5] = {color_w2, 3, 64, 64, 3, 4, 1, 1, kNbC2, 0, 0, 0, 1};
  jobs.j[6] = {color_w1, 64, 64, 64, 64, 4, 2, 1, kNbC1, 0, 0, 0, 1};
  jobs.j[7] = {color_w0, 64, 31, 16, 64, 1, 2, 1, kNbC0, 0, 1, 15, 1};      // row i >= 1 <- input column 15 + i
  jobs.j[8] = {sigma_w1, 16, 64, 64, 16, 4, 1, 1, kNbS1, 0, 0, 0, 1};
  jobs.j[9] = {sigma_w0, 64, 32, 32, 64, 2, 2, 1, kNbS0, 0, 0, 0, 1};
  k_pack_weights<<<dim3(8, 10), 256, 0, as_stream(s)>>>(jobs, packed_fwd, packed_bwd);
  return check_launch("nerf_pack_weights_device");
}

int inr_nerf_forward_train(const float* x, const float* d, int64_t M, float bound, const float* embeddings,
                           const inr_grid_desc* desc, const float* packed, float* sigma, float* rgb, float* enc,
                           float* h1, float* so, float* cin, float* c1, float* c2, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && d && embeddings && packed && sigma && rgb && enc && h1 && so && cin && c1 && c2, "null pointer");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 &&
                  (((uintptr_t)packed | (uintptr_t)enc | (uintptr_t)h1 | (uintptr_t)so | (uintptr_t)cin | (uintptr_t)c1 |
                    (uintptr_t)c2) & 15) == 0, "embeddings/packed/activation arrays misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes;
  const int grid = grid_for(k_nerf_fwd<true, false, 1>, lds, (M + 15) / 16);
  k_nerf_fwd<true, false, 1><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      x, d, M, nullptr, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed), 1.0f, sigma, rgb, nullptr, nullptr, nullptr,
      NerfSave{enc, h1, so, cin, c1, c2}, nullptr);
  return check_launch("nerf_forward_train");
}

int inr_nerf_forward_enc(const float* x, const float* d, int64_t M, float bound, const float* embeddings,
                         const inr_grid_desc* desc, const float* packed, float* sigma, float* rgb, float* enc, inr_stream_t s) {
  INR_REQUIRE(M >= 0 && desc, "bad argument");
  if (M == 0) return INR_OK;
  INR_REQUIRE(x && d && embeddings && packed && sigma && rgb && enc, "null pointer");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && (((uintptr_t)packed | (uintptr_t)enc) & 15) == 0,
              "embeddings/packed/enc misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes;
  const int grid = grid_for(k_nerf_fwd<true, false, 2>, lds, (M + 15) / 16);
  NerfSave sv{};
  sv.enc = enc;
  k_nerf_fwd<true, false, 2><<<grid, kFieldThreads, lds, as_stream(s)>>>(
      x, d, M, nullptr, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed), 1.0f, sigma, rgb, nullptr, nullptr, nullptr, sv, nullptr);
  return check_launch("nerf_forward_enc");
}

int inr_nerf_head_backward(const float* enc, const float* d, const float* grad_sigma, const float* grad_rgb, int64_t M,
                           float density_scale, const float* packed_fwd, const float* packed_bwd, float* grad_enc,
                           void* workspace, float* grad_ws0, float* grad_ws1, float* grad_wc0, float* grad_wc1,
                           float* grad_wc2, float* zero_buf, int64_t zero_floats, inr_stream_t s) {
  INR_REQUIRE(M >= 0, "negative M");
  INR_REQUIRE(zero_floats >= 0 && (!zero_buf || ((uintptr_t)zero_buf & 15) == 0), "zero_buf must be 16-byte aligned");
  INR_REQUIRE(packed_fwd && packed_bwd && workspace && grad_ws0 && grad_ws1 && grad_wc0 && grad_wc1 && grad_wc2, "null pointer");
  INR_REQUIRE(M == 0 || (enc && d && grad_sigma && grad_rgb && grad_enc), "null sample arrays");
  INR_REQUIRE((((uintptr_t)enc | (uintptr_t)packed_fwd | (uintptr_t)packed_bwd | (uintptr_t)grad_enc | (uintptr_t)workspace) & 15) == 0,
              "arrays must be 16-byte aligned");
  hipStream_t st = as_stream(s);
  const size_t lds = (size_t)(kNhFwdFloats + kNerfBwdFloats + kHbStageFloats) * sizeof(float);
  static AttrOnce attr;
  if (!attr.ensure(k_nerf_head_bwd, lds)) {
    set_error("nerf_head_backward: %zu bytes of LDS refused", lds);
    return INR_ELAUNCH;
  }
  const int grid = M == 0 ? 1 : grid_for(k_nerf_head_bwd, lds, (M + 15) / 16);
  k_nerf_head_bwd<<<grid, kFieldThreads, lds, st>>>(enc, d, grad_sigma, grad_rgb, M, density_scale,
                                                    reinterpret_cast<const float4*>(packed_fwd),
                                                    reinterpret_cast<const float4*>(packed_bwd), grad_enc,
                                                    reinterpret_cast<float4*>(workspace), zero_buf, zero_floats);
  k_nerf_head_wgrad_reduce<<<kHbAccTiles * 256 / 64, 1024, 0, st>>>(reinterpret_cast<const float*>(workspace), grid, grad_wc2,
                                                                    grad_wc1, grad_wc0, grad_ws1, grad_ws0);
  return check_launch("nerf_head_backward");
}

int inr_nerf_backward(const float* grad_sigma, const float* grad_rgb, const float* rgb, const float* so, const float* h1,
                      const float* c1, const float* c2, int64_t M, float density_scale, const float* packed_bwd,
                      float* grad_o, float* grad_zc2, float* grad_zc1, float* grad_so, float* grad_zh1, float* grad_enc,
                      inr_stream_t s) {
  INR_REQUIRE(M >= 0, "negative M");
  if (M == 0) return INR_OK;
  INR_REQUIRE(grad_sigma && grad_rgb && rgb && so && h1 && c1 && c2 && packed_bwd && grad_o && grad_zc2 && grad_zc1 &&
                  grad_so && grad_zh1 && grad_enc, "null pointer");
  INR_REQUIRE((((uintptr_t)so | (uintptr_t)h1 | (uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)packed_bwd | (uintptr_t)grad_o |
                (uintptr_t)grad_zc2 | (uintptr_t)grad_zc1 | (uintptr_t)grad_so | (uintptr_t)grad_zh1 |
                (uintptr_t)grad_enc) & 15) == 0, "arrays must be 16-byte aligned");
  const size_t lds = (size_t)kNerfBwdFloats * sizeof(float);
  NerfBwdIO io{grad_sigma, grad_rgb, rgb, so, h1, c1, c2, grad_o, grad_zc2, grad_zc1, grad_so, grad_zh1, grad_enc};
  k_nerf_bwd<<<grid_for(k_nerf_bwd, lds, (M + 15) / 16), kFieldThreads, lds, as_stream(s)>>>(
      io, M, density_scale, reinterpret_cast<const float4*>(packed_bwd));
  return check_launch("nerf_backward");
}

static int instance_render_impl(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M,
                                float bound, const void* embeddings, bool fast, const inr_grid_desc* desc,
                                const float* packed, int32_t K, float* extra_out, int32_t x_is_01, uint64_t* cursors,
                                inr_stream_t s);

int inr_instance_render(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M, float bound,
                        const float* embeddings, const inr_grid_desc* desc, const float* packed, int32_t K,
                        float* extra_out, int32_t x_is_01, uint64_t* cursors, inr_stream_t s) {
  return instance_render_impl(xyzs, rays, weights, N, M, bound, embeddings, false, desc, packed, K, extra_out, x_is_01,
                              cursors, s);
}

int inr_instance_render_fast(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M, float bound,
                             const void* embeddings_half, const inr_grid_desc* desc, const float* packed_f16, int32_t K,
                             float* extra_out, int32_t x_is_01, uint64_t* cursors, inr_stream_t s) {
#if INR_MLP_FP32
  (void)xyzs; (void)rays; (void)weights; (void)N; (void)M; (void)bound; (void)embeddings_half; (void)desc;
  (void)packed_f16; (void)K; (void)extra_out; (void)x_is_01; (void)cursors; (void)s;
  set_error("instance_render_fast: not available in the exact-fp32 build");
  return INR_EINVAL;
#else
  return instance_render_impl(xyzs, rays, weights, N, M, bound, embeddings_half, true, desc, packed_f16, K, extra_out,
                              x_is_01, cursors, s);
#endif
}

int inr_instance_pack_weights_f16(const float* w0, const float* w1, const float* w2, int32_t K, float* packed) {
  INR_REQUIRE(w0 && w1 && w2 && packed, "null pointer");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  pack_section_f16(packed + kIns0, w0, 64, 32, 4, 8, kidx_enc);
  pack_section_f16(packed + kIns1, w1, 64, 64, 4, 16, kidx_hidden);
  pack_section_f16(packed + kIns2, w2, K, 64, K / 16, 16, kidx_hidden);
  return INR_OK;
}

static int instance_render_impl(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M,
                                float bound, const void* embeddings, bool fast, const inr_grid_desc* desc,
                                const float* packed, int32_t K, float* extra_out, int32_t x_is_01, uint64_t* cursors,
                                inr_stream_t s) {
  INR_REQUIRE(N >= 0 && M >= 0 && desc, "bad argument");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays && embeddings && packed && extra_out && cursors && ((uintptr_t)cursors & 7) == 0, "null pointer");
  INR_REQUIRE(M == 0 || (xyzs && weights), "null sample arrays");
  INR_REQUIRE(K > 0 && K <= 64 && K % 16 == 0, "K must be 16, 32, 48 or 64");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)extra_out & 15) == 0,
              "embeddings/packed/extra_out misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const float2* e = reinterpret_cast<const float2*>(embeddings);
  const float4* p = reinterpret_cast<const float4*>(packed);
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const uint32_t eb = (uint32_t)(fast ? emb_bytes64 / 2 : emb_bytes64);
  // weights + level records + two buffers of eight partial [16 rays x K] sums
  const size_t lds = (size_t)(kIns2 + K * 64) * sizeof(float) + kLevelRecBytes + 2 * (kFieldThreads / 64) * (size_t)K * 64;
  const int64_t n_groups = (N + 15) / 16;
  const int64_t as_tiles = n_groups * (kFieldThreads / 64);        // one workgroup per group
  INR_REQUIRE(n_groups < ((int64_t)1 << 31) - 65536, "too many rays for the 32-bit group cursor");
  unsigned long long* cur = reinterpret_cast<unsigned long long*>(cursors);
  hipStream_t st = as_stream(s);
#define INR_IR_LAUNCH(KMT, O)                                                                                         \
  k_instance_render<KMT, O><<<grid_for(k_instance_render<KMT, O>, lds, as_tiles), kFieldThreads, lds, st>>>(              \
      xyzs, rays, weights, N, M, bound, e, eb, G, p, extra_out, x_is_01, cur)
#if INR_MLP_FP32
  (void)fast;
  switch (K / 16) {
    case 1: INR_IR_LAUNCH(1, false); break;
    case 2: INR_IR_LAUNCH(2, false); break;
    case 3: INR_IR_LAUNCH(3, false); break;
    default: INR_IR_LAUNCH(4, false); break;
  }
#else
  if (fast) {
    switch (K / 16) {
      case 1: INR_IR_LAUNCH(1, true); break;
      case 2: INR_IR_LAUNCH(2, true); break;
      case 3: INR_IR_LAUNCH(3, true); break;
      default: INR_IR_LAUNCH(4, true); break;
    }
  } else {
    switch (K / 16) {
      case 1: INR_IR_LAUNCH(1, false); break;
      case 2: INR_IR_LAUNCH(2, false); break;
      case 3: INR_IR_LAUNCH(3, false); break;
      default: INR_IR_LAUNCH(4, false); break;
    }
  }
#endif
#undef INR_IR_LAUNCH
  return check_launch("instance_render");
}

static int nerf_render_impl(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d, int64_t N,
                            int64_t M, float bound, const void* embeddings, bool fast, const inr_grid_desc* desc,
                            const float* packed, float density_scale, float T_thresh, float* weights_sum, float* depth,
                            float* image, float* weights, uint64_t* evaluated, int32_t x_is_01, inr_stream_t s);

int inr_nerf_render(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d, int64_t N,
                    int64_t M, float bound, const float* embeddings, const inr_grid_desc* desc, const float* packed,
                    float density_scale, float T_thresh, float* weights_sum, float* depth, float* image, float* weights,
                    uint64_t* evaluated, int32_t x_is_01, inr_stream_t s) {
  return nerf_render_impl(xyzs, deltas, rays, rays_d, N, M, bound, embeddings, false, desc, packed, density_scale, T_thresh,
                          weights_sum, depth, image, weights, evaluated, x_is_01, s);
}

int inr_nerf_render_fast(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d, int64_t N,
                         int64_t M, float bound, const void* embeddings_half, const inr_grid_desc* desc,
                         const float* packed_f16, float density_scale, float T_thresh, float* weights_sum, float* depth,
                         float* image, float* weights, uint64_t* evaluated, int32_t x_is_01, inr_stream_t s) {
#if INR_MLP_FP32
  (void)xyzs; (void)deltas; (void)rays; (void)rays_d; (void)N; (void)M; (void)bound; (void)embeddings_half; (void)desc;
  (void)packed_f16; (void)density_scale; (void)T_thresh; (void)weights_sum; (void)depth; (void)image; (void)weights;
  (void)evaluated; (void)x_is_01; (void)s;
  set_error("nerf_render_fast: not available in the exact-fp32 build");
  return INR_EINVAL;
#else
  return nerf_render_impl(xyzs, deltas, rays, rays_d, N, M, bound, embeddings_half, true, desc, packed_f16, density_scale,
                          T_thresh, weights_sum, depth, image, weights, evaluated, x_is_01, s);
#endif
}

static int nerf_render_impl(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d, int64_t N,
                            int64_t M, float bound, const void* embeddings, bool fast, const inr_grid_desc* desc,
                            const float* packed, float density_scale, float T_thresh, float* weights_sum, float* depth,
                            float* image, float* weights, uint64_t* evaluated, int32_t x_is_01, inr_stream_t s) {
  INR_REQUIRE(N >= 0 && M >= 0 && desc, "bad argument");
  if (N == 0) return INR_OK;
  INR_REQUIRE(rays && rays_d && embeddings && packed && weights_sum && depth && image && evaluated, "null pointer");
  INR_REQUIRE(M == 0 || (xyzs && deltas), "null sample arrays");
  INR_REQUIRE(((uintptr_t)embeddings & 7) == 0 && ((uintptr_t)packed & 15) == 0 && ((uintptr_t)deltas & 7) == 0 &&
                  ((uintptr_t)evaluated & 7) == 0, "embeddings/packed/deltas/evaluated misaligned");
  GridDesc G;
  int rc = make_grid_desc(desc, G);
  if (rc) return rc;
  const uint64_t emb_bytes64 = (uint64_t)desc->offsets[desc->num_levels] * 8ull;
  INR_REQUIRE(emb_bytes64 < (1ull << 31), "table larger than 2 GiB is not addressable by the 32-bit gather offsets");
  const size_t lds = kNerfFloats * sizeof(float) + kLevelRecBytes;
  const int64_t n_groups = (N + 15) / 16;
  INR_REQUIRE(n_groups < ((int64_t)1 << 31) - 65536, "too many rays for the 32-bit group cursor");
#if !INR_MLP_FP32
  if (fast) {
    k_nerf_render<true><<<grid_for(k_nerf_render<true>, lds, n_groups), kFieldThreads, lds, as_stream(s)>>>(
        xyzs, deltas, rays, rays_d, N, M, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)(emb_bytes64 / 2), G,
        reinterpret_cast<const float4*>(packed), density_scale, T_thresh, weights_sum, depth, image, weights,
        reinterpret_cast<unsigned long long*>(evaluated), x_is_01);
    return check_launch("nerf_render_fast");
  }
#endif
  (void)fast;
  k_nerf_render<false><<<grid_for(k_nerf_render<false>, lds, n_groups), kFieldThreads, lds, as_stream(s)>>>(
      xyzs, deltas, rays, rays_d, N, M, bound, reinterpret_cast<const float2*>(embeddings), (uint32_t)emb_bytes64, G,
      reinterpret_cast<const float4*>(packed), density_scale, T_thresh, weights_sum, depth, image, weights,
      reinterpret_cast<unsigned long long*>(evaluated), x_is_01);
  return check_launch("nerf_render");
}

}  // extern "C"
