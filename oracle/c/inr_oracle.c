/* CPU oracle in plain C for the instance-field NeRF render hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this library (through oracle/c_port.py); nothing under instance_nerf_amd/ does.
 *
 * PARITY UNPINNED: the reference's implementation of this path is an un-vendored git submodule
 * (/root/reference/.gitmodules:4-6 -> zymk9/torch-ngp, pinned in prose at /root/reference/README.md:27,59);
 * /root/reference/instance_nerf is empty and the reference ships no tests or golden vectors.  This file is a
 * second, independent restatement (scalar C, one ray / one sample at a time) of the algorithm spec recorded in
 * SURVEY.md Appendix A - the first is the vectorised numpy/torch one in the .py modules of oracle/ - and the two are checked
 * against each other and against the tests/golden vectors by tests/test_oracle_c.py.
 *
 * Arithmetic contract: every marching operation is one IEEE binary32 operation in the order written (build
 * with -ffp-contract=off, no -ffast-math; x86-64 SSE has no excess precision), so counts, offsets and sample
 * positions are bit-identical to oracle/march.py and to the HIP kernels.  Field arithmetic is fp32 with
 * sequential dot products (differs from BLAS summation order in the last bits).
 *
 * Upstream symbols restated: a2 near_far_from_aabb, a3 morton3D / packbits,
 * a4 march_rays_train, a7 gridencoder forward, a9 NeRFNetwork.forward, a10 SH-4, a12 composite_rays_train
 * forward, a13 instance logits, and orc_render = the a2 -> a4 -> a7/a9/a10 -> a12 chain per ray (OpenMP over
 * rays) that bench.py times as the CPU baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LEVELS 16

typedef struct {
  int32_t num_levels;
  uint32_t offsets[ORC_MAX_LEVELS + 1];
  float scales[ORC_MAX_LEVELS];
  uint32_t resolutions[ORC_MAX_LEVELS];
  uint32_t hashed[ORC_MAX_LEVELS];
} orc_grid;

typedef struct {
  const float* embeddings; /* [T,2] */
  const float* sigma_w0;   /* [64,32] */
  const float* sigma_w1;   /* [16,64] */
  const float* color_w0;   /* [64,31] */
  const float* color_w1;   /* [64,64] */
  const float* color_w2;   /* [3,64]  */
} orc_nerf;

typedef struct {
  const float* embeddings; /* [T,2] */
  const float* w0;         /* [64,32] */
  const float* w1;         /* [64,64] */
  const float* w2;         /* [K,64]  */
  int32_t K;
} orc_inst;

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ---- a3: Morton codes, bit packing (SURVEY Appendix A.1 "Morton", "packbits") ------------------------- */
static inline uint32_t expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
static inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static inline uint32_t compact_bits(uint32_t x) {
  x &= 0x49249249u;
  x = (x | (x >> 2)) & 0xC30C30C3u;
  x = (x | (x >> 4)) & 0x0F00F00Fu;
  x = (x | (x >> 8)) & 0xFF0000FFu;
  x = (x | (x >> 16)) & 0x0000FFFFu;
  return x;
}
void orc_morton3D(const int32_t* coords, int64_t n, uint32_t* out) {
  for (int64_t i = 0; i < n; ++i)
    out[i] = morton3((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}
void orc_morton3D_invert(const uint32_t* idx, int64_t n, int32_t* coords) {
  for (int64_t i = 0; i < n; ++i) {
    coords[3 * i] = (int32_t)compact_bits(idx[i]);
    coords[3 * i + 1] = (int32_t)compact_bits(idx[i] >> 1);
    coords[3 * i + 2] = (int32_t)compact_bits(idx[i] >> 2);
  }
}
void orc_packbits(const float* grid, int64_t n, float thresh, uint8_t* bits) {
  for (int64_t k = 0; k < n / 8; ++k) {
    uint8_t b = 0;
    for (int i = 0; i < 8; ++i)
      if (grid[8 * k + i] > thresh) b |= (uint8_t)(1u << i);
    bits[k] = b;
  }
}

/* ---- a2: slab test (SURVEY Appendix A.1 "near_far_from_aabb") ------------------------------------------- */
void orc_near_far(const float* o, const float* d, const float* aabb, float min_near, int64_t N, float* nears,
                  float* fars) {
  for (int64_t i = 0; i < N; ++i) {
    float near = -INFINITY, far = INFINITY;
    int miss = 0;
    for (int a = 0; a < 3; ++a) {
      const float rd = 1.0f / d[3 * i + a];
      const float t0 = (aabb[a] - o[3 * i + a]) * rd;
      const float t1 = (aabb[a + 3] - o[3 * i + a]) * rd;
      const float lo = t0 > t1 ? t1 : t0;
      const float hi = t0 > t1 ? t0 : t1;
      if (a > 0 && (near > hi || lo > far)) miss = 1;
      if (lo > near) near = lo;
      if (hi < far) far = hi;
    }
    if (near < min_near) near = min_near;
    nears[i] = miss ? 3.402823466e+38f : near;
    fars[i] = miss ? 3.402823466e+38f : far;
  }
}

/* ---- a4: occupancy-grid march of one ray (SURVEY Appendix A.1 "march_rays_train") ----------------------- */
typedef struct {
  float bound, dt_gamma, dt_min, dt_max;
  int32_t cascade, H, max_steps;
  const uint8_t* bits;
} march_cfg;

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float dt_of(const march_cfg* c, float t) { return clampf(t * c->dt_gamma, c->dt_min, c->dt_max); }
static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void make_cfg(march_cfg* c, const uint8_t* bits, float bound, int32_t cascade, int32_t H, float dt_gamma,
                     int32_t max_steps) {
  const float s32 = 3.4641016151377544f; /* float32(2*sqrt(3)) */
  c->bits = bits;
  c->bound = bound;
  c->cascade = cascade;
  c->H = H;
  c->dt_gamma = dt_gamma;
  c->max_steps = max_steps;
  c->dt_min = s32 / (float)max_steps;
  c->dt_max = s32 * (float)(1 << (cascade - 1)) / (float)H;
}

/* Emits up to max_steps samples of one ray into xyz[3k..], dts[k], dls[k] (any may be NULL); returns the count. */
static int march_one(const march_cfg* c, const float* o, const float* d, float near, float far, float noise,
                     float* xyz, float* dts, float* dls) {
  const int H = c->H;
  const float invH = (float)(1.0 / (double)H);
  float rd[3], sg[3];
  for (int a = 0; a < 3; ++a) {
    rd[a] = 1.0f / d[a];
    sg[a] = copysignf(1.0f, d[a]);
  }
  float t = near + dt_of(c, near) * noise;
  float last_t = t;
  int count = 0;
  while (t < far && count < c->max_steps) {
    float p[3];
    for (int a = 0; a < 3; ++a) {
      const float m = t * d[a];
      p[a] = clampf(o[a] + m, -c->bound, c->bound);
    }
    const float dt = dt_of(c, t);
    int e1, e2;
    (void)frexpf(fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2]))), &e1);
    (void)frexpf(dt * (float)H * 0.5f, &e2);
    const int level = clampi(e1, 0, c->cascade - 1) > clampi(e2, 0, c->cascade - 1) ? clampi(e1, 0, c->cascade - 1)
                                                                                     : clampi(e2, 0, c->cascade - 1);
    const float mb = fminf(ldexpf(1.0f, level), c->bound);
    const float rmb = 1.0f / mb;
    int n[3];
    for (int a = 0; a < 3; ++a) {
      const float f = ((p[a] * rmb + 1.0f) * 0.5f) * (float)H;
      n[a] = clampi((int)f, 0, H - 1);
    }
    const int64_t bit = (int64_t)level * H * H * H + (int64_t)morton3((uint32_t)n[0], (uint32_t)n[1], (uint32_t)n[2]);
    const int occ = (c->bits[bit >> 3] >> (bit & 7)) & 1;
    if (occ) {
      const float tn = t + dt;
      if (xyz) {
        xyz[3 * count] = p[0]; xyz[3 * count + 1] = p[1]; xyz[3 * count + 2] = p[2];
      }
      if (dts) dts[count] = dt;
      if (dls) dls[count] = tn - last_t;
      t = tn;
      last_t = tn;
      ++count;
    } else {
      float tc[3];
      for (int a = 0; a < 3; ++a) {
        const float aa = ((float)n[a] + 0.5f) + 0.5f * sg[a];
        tc[a] = ((aa * invH * 2.0f - 1.0f) * mb - p[a]) * rd[a];
      }
      const float tm = fminf(tc[0], fminf(tc[1], tc[2]));
      const float tt = t + fmaxf(0.0f, tm);
      do {
        t = t + dt_of(c, t);
      } while (t < tt);
    }
  }
  return count;
}

/* counts[N]; rays[N,3] = (n, exclusive-scan offset, count); returns the total. */
int64_t orc_march_count(const float* o, const float* d, const uint8_t* bits, float bound, int32_t cascade, int32_t H,
                        const float* nears, const float* fars, const float* noises, float dt_gamma, int32_t max_steps,
                        int64_t N, int32_t* rays) {
  march_cfg c;
  make_cfg(&c, bits, bound, cascade, H, dt_gamma, max_steps);
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t i = 0; i < N; ++i)
    rays[3 * i + 2] = march_one(&c, o + 3 * i, d + 3 * i, nears[i], fars[i], noises ? noises[i] : 0.0f, NULL, NULL, NULL);
  int64_t total = 0;
  for (int64_t i = 0; i < N; ++i) {
    rays[3 * i] = (int32_t)i;
    rays[3 * i + 1] = (int32_t)total;
    total += rays[3 * i + 2];
  }
  return total;
}

/* Second pass: samples of ray n go to slots rays[n,1] .. +rays[n,2]; a ray with offset + count > M is dropped
 * (upstream drops overflowing rays the same way).  Arrays must be zero-filled by the caller. */
void orc_march_write(const float* o, const float* d, const uint8_t* bits, float bound, int32_t cascade, int32_t H,
                     const float* nears, const float* fars, const float* noises, float dt_gamma, int32_t max_steps,
                     int64_t N, const int32_t* rays, int64_t M, float* xyzs, float* dirs, float* deltas) {
  march_cfg c;
  make_cfg(&c, bits, bound, cascade, H, dt_gamma, max_steps);
#pragma omp parallel
  {
    float* xyz = (float*)malloc(sizeof(float) * 3 * (size_t)max_steps);
    float* dts = (float*)malloc(sizeof(float) * (size_t)max_steps);
    float* dls = (float*)malloc(sizeof(float) * (size_t)max_steps);
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < N; ++i) {
      const int64_t off = rays[3 * i + 1], cnt = rays[3 * i + 2];
      if (off + cnt > M) continue;
      const int got = march_one(&c, o + 3 * i, d + 3 * i, nears[i], fars[i], noises ? noises[i] : 0.0f, xyz, dts, dls);
      for (int k = 0; k < got; ++k) {
        for (int a = 0; a < 3; ++a) {
          xyzs[3 * (off + k) + a] = xyz[3 * k + a];
          dirs[3 * (off + k) + a] = d[3 * i + a];
        }
        deltas[2 * (off + k)] = dts[k];
        deltas[2 * (off + k) + 1] = dls[k];
      }
    }
    free(xyz); free(dts); free(dls);
  }
}

/* ---- a7: multiresolution hash-grid encoding of one point (SURVEY Appendix A.1 "Hash grid") --------------- */
static void encode_one(const orc_grid* G, const float* emb, float bound, const float* x, float* out /* [2L] */) {
  const float rb = 2.0f * bound;
  float x01[3];
  int oob = 0;
  for (int a = 0; a < 3; ++a) {
    x01[a] = (x[a] + bound) / rb;
    if (!(x01[a] >= 0.0f && x01[a] <= 1.0f)) oob = 1;
  }
  if (oob) { /* upstream's flag_oob: zero features */
    for (int i = 0; i < 2 * G->num_levels; ++i) out[i] = 0.0f;
    return;
  }
  for (int l = 0; l < G->num_levels; ++l) {
    const float s = G->scales[l];
    const uint32_t rows = G->offsets[l + 1] - G->offsets[l];
    const uint32_t res1 = G->resolutions[l] + 1;
    float fr[3];
    uint32_t g[3];
    for (int a = 0; a < 3; ++a) {
      const float pos = x01[a] * s + 0.5f;
      const float fl = floorf(pos);
      fr[a] = pos - fl;
      g[a] = (uint32_t)fl;
    }
    float ax = 0.0f, ay = 0.0f;
    for (int k = 0; k < 8; ++k) {
      const uint32_t cx = g[0] + (k & 1), cy = g[1] + ((k >> 1) & 1), cz = g[2] + ((k >> 2) & 1);
      const float wx = (k & 1) ? fr[0] : 1.0f - fr[0];
      const float wy = (k & 2) ? fr[1] : 1.0f - fr[1];
      const float wz = (k & 4) ? fr[2] : 1.0f - fr[2];
      const float w = (wx * wy) * wz;
      uint32_t idx;
      if (G->hashed[l]) idx = (cx ^ (cy * 2654435761u) ^ (cz * 805459861u)) % rows;
      else idx = (cx + cy * res1 + cz * res1 * res1) % rows;
      const float* row = emb + 2 * ((size_t)G->offsets[l] + idx);
      const float px = w * row[0], py = w * row[1];
      ax = ax + px;
      ay = ay + py;
    }
    out[2 * l] = ax;
    out[2 * l + 1] = ay;
  }
}

void orc_grid_encode(const float* x, const float* emb, const orc_grid* G, float bound, int64_t M, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) encode_one(G, emb, bound, x + 3 * m, out + (size_t)m * 2 * G->num_levels);
}

/* ---- a10: degree-4 real spherical harmonics (SURVEY Appendix A.1 "SH degree 4") --------------------------- */
static void sh4(const float* d, float* v) {
  const float x = d[0], y = d[1], z = d[2];
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
void orc_sh4(const float* d, int64_t M, float* out) {
  for (int64_t m = 0; m < M; ++m) sh4(d + 3 * m, out + 16 * m);
}

/* ---- a9 / a13: the MLPs (bias-free Linear + ReLU; SURVEY Appendix A.1 "Network") --------------------------- */
static void linear(const float* W, int n_out, int n_in, const float* x, float* y, int relu) {
  for (int o = 0; o < n_out; ++o) {
    float acc = 0.0f;
    const float* w = W + (size_t)o * n_in;
    for (int i = 0; i < n_in; ++i) acc += w[i] * x[i];
    y[o] = (relu && acc < 0.0f) ? 0.0f : acc;
  }
}

/* sigma (= exp(h0), no density_scale), rgb (nullable), geo (nullable, 15), raw (nullable: h0) of one sample */
static void nerf_one(const orc_grid* G, const orc_nerf* P, float bound, const float* x, const float* sh16, float* sigma,
                     float* rgb, float* geo, float* raw) {
  float enc[2 * ORC_MAX_LEVELS], h1[64], h[16];
  encode_one(G, P->embeddings, bound, x, enc);
  linear(P->sigma_w0, 64, 2 * G->num_levels, enc, h1, 1);
  linear(P->sigma_w1, 16, 64, h1, h, 0);
  *sigma = expf(h[0]);
  if (raw) *raw = h[0];
  if (geo) memcpy(geo, h + 1, 15 * sizeof(float));
  if (rgb) {
    float cin[31], c1[64], c2[64], o[3];
    memcpy(cin, sh16, 16 * sizeof(float));
    memcpy(cin + 16, h + 1, 15 * sizeof(float));
    linear(P->color_w0, 64, 31, cin, c1, 1);
    linear(P->color_w1, 64, 64, c1, c2, 1);
    linear(P->color_w2, 3, 64, c2, o, 0);
    for (int c = 0; c < 3; ++c) rgb[c] = 1.0f / (1.0f + expf(-o[c]));
  }
}

void orc_nerf_forward(const float* x, const float* d, int64_t M, float bound, const orc_grid* G, const orc_nerf* P,
                      float* sigma, float* rgb, float* geo) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    float sh[16];
    if (rgb) sh4(d + 3 * m, sh);
    nerf_one(G, P, bound, x + 3 * m, sh, sigma + m, rgb ? rgb + 3 * m : NULL, geo ? geo + 15 * m : NULL, NULL);
  }
}

static void inst_one(const orc_grid* G, const orc_inst* P, float bound, const float* x, float* logits) {
  float enc[2 * ORC_MAX_LEVELS], h1[64], h2[64];
  encode_one(G, P->embeddings, bound, x, enc);
  linear(P->w0, 64, 2 * G->num_levels, enc, h1, 1);
  linear(P->w1, 64, 64, h1, h2, 1);
  linear(P->w2, P->K, 64, h2, logits, 0);
}

void orc_instance_forward(const float* x, int64_t M, float bound, const orc_grid* G, const orc_inst* P, float* logits) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) inst_one(G, P, bound, x + 3 * m, logits + (size_t)m * P->K);
}

/* ---- a12 / a13: alpha compositing forward (SURVEY Appendix A.1 "Composite") ------------------------------- */
void orc_composite_train(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, int64_t N,
                         int64_t M, float T_thresh, const float* extra, int32_t K, float* weights_sum, float* depth,
                         float* image, float* extra_out, float* weights) {
  for (int64_t r = 0; r < N; ++r) {
    const int64_t rid = rays[3 * r], off = rays[3 * r + 1], cnt = rays[3 * r + 2];
    float T = 1.0f, t = 0.0f, ws = 0.0f, dp = 0.0f, c[3] = {0.0f, 0.0f, 0.0f};
    if (extra_out) memset(extra_out + (size_t)rid * K, 0, sizeof(float) * (size_t)K);
    if (off + cnt <= M) {
      for (int64_t i = off; i < off + cnt; ++i) {
        if (T < T_thresh) break;
        const float alpha = 1.0f - expf(-sigmas[i] * deltas[2 * i]);
        const float w = alpha * T;
        ws += w;
        t += deltas[2 * i + 1];
        dp += w * t;
        for (int a = 0; a < 3; ++a) c[a] += w * rgbs[3 * i + a];
        if (extra_out)
          for (int k = 0; k < K; ++k) extra_out[(size_t)rid * K + k] += w * extra[(size_t)i * K + k];
        if (weights) weights[i] = w;
        T *= 1.0f - alpha;
      }
    }
    weights_sum[rid] = ws;
    depth[rid] = dp;
    image[3 * rid] = c[0]; image[3 * rid + 1] = c[1]; image[3 * rid + 2] = c[2];
  }
}

/* ---- the whole path per ray: a2 -> a4 -> a7/a9/a10 (-> a13) -> a12 -> background / depth normalisation ------
 * One ray at a time, OpenMP over rays; nothing of size M is materialised.  image[N,3], depth[N] (normalised),
 * weights_sum[N], instance[N,K] (nullable, with `I`), counts[N] (nullable).  Returns the number of live samples
 * the march produced (every one of them is evaluated, as on the training path). */
int64_t orc_render(const float* o, const float* d, int64_t N, const uint8_t* bits, float bound, int32_t cascade, int32_t H,
                   float min_near, float dt_gamma, int32_t max_steps, float T_thresh, float bg, float density_scale,
                   int32_t absolute_depth, const orc_grid* G, const orc_nerf* P, const orc_inst* I, float* image,
                   float* depth, float* weights_sum, float* instance, int32_t* counts) {
  march_cfg c;
  make_cfg(&c, bits, bound, cascade, H, dt_gamma, max_steps);
  const float aabb[6] = {-bound, -bound, -bound, bound, bound, bound};
  int64_t total = 0;
#pragma omp parallel reduction(+ : total)
  {
    float* xyz = (float*)malloc(sizeof(float) * 3 * (size_t)max_steps);
    float* dts = (float*)malloc(sizeof(float) * (size_t)max_steps);
    float* dls = (float*)malloc(sizeof(float) * (size_t)max_steps);
    float logits[64];
#pragma omp for schedule(dynamic, 16)
    for (int64_t r = 0; r < N; ++r) {
      float near, far, sh[16];
      orc_near_far(o + 3 * r, d + 3 * r, aabb, min_near, 1, &near, &far);
      const int cnt = march_one(&c, o + 3 * r, d + 3 * r, near, far, 0.0f, xyz, dts, dls);
      total += cnt;
      if (counts) counts[r] = cnt;
      sh4(d + 3 * r, sh);
      /* depth: upstream's training compositing counts t from the first step (composite_rays_train: t = 0), its
       * inference compositing from the ray's current parameter (composite_rays: t = rays_t = near) */
      float T = 1.0f, t = absolute_depth ? near : 0.0f, ws = 0.0f, dp = 0.0f, col[3] = {0.0f, 0.0f, 0.0f};
      if (instance) memset(instance + (size_t)r * I->K, 0, sizeof(float) * (size_t)I->K);
      for (int k = 0; k < cnt; ++k) {
        /* every marched sample is evaluated (training semantics); only its weight stops counting */
        float sigma, rgb[3];
        nerf_one(G, P, bound, xyz + 3 * k, sh, &sigma, rgb, NULL, NULL);
        if (instance) inst_one(G, I, bound, xyz + 3 * k, logits);
        if (T < T_thresh) continue;
        const float alpha = 1.0f - expf(-(sigma * density_scale) * dts[k]);
        const float w = alpha * T;
        ws += w;
        t += dls[k];
        dp += w * t;
        for (int a = 0; a < 3; ++a) col[a] += w * rgb[a];
        if (instance)
          for (int q = 0; q < I->K; ++q) instance[(size_t)r * I->K + q] += w * logits[q];
        T *= 1.0f - alpha;
      }
      for (int a = 0; a < 3; ++a) image[3 * r + a] = col[a] + (1.0f - ws) * bg;
      depth[r] = fmaxf(dp - near, 0.0f) / (far - near);
      weights_sum[r] = ws;
    }
    free(xyz); free(dts); free(dls);
  }
  return total;
}
