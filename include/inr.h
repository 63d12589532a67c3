/*
 * inr.h - C ABI of libinr_hip.so: the MI355X (gfx950) implementation of the
 * instance-field NeRF render/train hot path.
 *
 * This is the drop-in boundary.  Every entry point
 * replaces one function of the reference's torch-ngp extension modules.  Those
 * modules are an UN-VENDORED SUBMODULE of the reference
 * (/root/reference/.gitmodules:4-6, pinned in prose at
 * /root/reference/README.md:27,59 to zymk9/torch-ngp @ 6be6af19), so the
 * "replaces" notes below name the upstream extension symbol and the survey
 * row instead of a file:line
 * inside /root/reference.  The one FFI call site that IS in the reference tree,
 * roi_align_3d (/root/reference/nerf_rcnn/model/utils.py:608), is a "next" row
 * (section 8f, f2) and is declared at the end of this header.
 *
 * Conventions (chosen to fix the flaws of the reference's only in-tree
 * extension, /root/reference/nerf_rcnn/model/rotated_iou/cuda_op/):
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *     marked "host"; buffers are owned and sized by the caller; nothing is
 *     allocated or freed inside the library;
 *   - every launch goes onto the stream passed in (cf. the default-stream bug at
 *     sort_vert_kernel.cu:137-138); no call synchronises the device;
 *   - return value 0 = success, negative = INR_E*; never exit()/abort()
 *     (cf. cuda_utils.h:26-35); inr_last_error() gives a message for the
 *     calling thread;
 *   - arrays are dense row-major, fp32 unless stated.
 */
#ifndef INR_H
#define INR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an existing prototype changes or an entry point is removed (round 2 changed four argument lists
 * without a bump: a stale library or an external caller built against the old header was only rejected by accident).
 * instance_nerf_amd/_lib.py refuses a library whose version differs from the one it was written against. */
#define INR_ABI_VERSION 9
#define INR_MAX_LEVELS 16

enum {
  INR_OK = 0,
  INR_EINVAL = -1,   /* bad argument (null pointer, size, unsupported shape) */
  INR_ELAUNCH = -2,  /* hipLaunch / runtime error, see inr_last_error()       */
  INR_ENODEV = -3    /* no usable gfx950 device                                */
};

typedef void* inr_stream_t; /* a hipStream_t; NULL = the null stream */

/* Host-side description of a multiresolution hash grid.  Filled by
 * the caller (instance_nerf_amd.gridencoder.level_table) so host and device
 * index from the same numbers. */
typedef struct inr_grid_desc {
  int32_t num_levels;                    /* L <= INR_MAX_LEVELS                     */
  int32_t level_dim;                     /* F, features per row; 2 supported        */
  uint32_t offsets[INR_MAX_LEVELS + 1];  /* first row of each level; [L] = T        */
  float scales[INR_MAX_LEVELS];          /* pos = x01 * scale + 0.5                 */
  uint32_t resolutions[INR_MAX_LEVELS];  /* ceil(scale) + 1; dense stride = res + 1 */
  uint32_t hashed[INR_MAX_LEVELS];       /* 1: spatial hash, 0: dense index         */
} inr_grid_desc;

int inr_abi_version(void);
const char* inr_last_error(void);
/* Fills props[0..3] = {CU count, wavefront size, LDS bytes per CU, gcn arch number}. */
int inr_device_info(int32_t device, int64_t* props);
/* No upstream counterpart.  For a caller that renders view after view with the march of view i+1 on one stream and
 * the field kernel of view i on another (NeRFRenderer.run_cuda(shade_stream=...)): on != 0 makes the eval field
 * launch (inr_nerf_forward_table) and the march launches (inr_march_rays_train_count, inr_march_rays_patch_write)
 * request extra LDS per workgroup, which is how a launch tells the dispatcher to keep one field workgroup and a
 * bounded number of march workgroups per CU.  Results do not change; process-wide; off by default.  on >= 1024: the
 * field workgroup's request in BYTES instead of the default 84 KB (not clamped: a request no CU can satisfy makes the
 * next field launch return INR_ELAUNCH); negative values and 2..1023 are INR_EINVAL.                               */
int inr_set_overlap_placement(int32_t on);
/* THREADING / MULTI-DEVICE NOTE for the three mode switches of this header - inr_set_overlap_placement,
 * inr_set_march_mode, inr_roi_align_3d_set_mode: each writes one plain process-global variable that every later launch
 * of the library reads, on every device and every stream.  They are meant to be set once, before the work starts, by
 * the one process that drives one GPU (the deployment this library is written for).  They are NOT thread-safe and NOT
 * per-device: a host that drives several devices from one process, or flips a mode from one thread while another thread
 * launches, gets whichever value the launch happens to read (every value selects a correct kernel - results do not
 * depend on them beyond fp32 summation order in RoIAlign - only the speed and the LDS footprint change).  Everything
 * else in the library is stateless apart from inr_last_error()'s thread-local message.                              */
/* No upstream counterpart.  Which marcher inr_march_rays_train_count / _write use: -1 (default) by batch size
 * (wave per ray up to 32768 rays, lane per ray above), 0 lane per ray, 1 wave per ray.  Both produce the same
 * bits; the parity tests run both.  Process-wide.  (The library reads no environment variables.)                  */
int inr_set_march_mode(int32_t mode);

/* ---- rays: generation (replaces nerf/utils.py::get_rays) and ray/AABB (a2) -------------
 * poses [B,4,4] camera-to-world row-major; pixel `inds[k]` (flat j*W+i; NULL = 0..n-1) -> rays_o/rays_d [B,n,3]:
 * dir = ((i+.5-cx)/fx, (j+.5-cy)/fy, 1) normalised and rotated by the pose, origin = translation.     */
int inr_get_rays(const float* poses, int64_t B, float fx, float fy, float cx, float cy, int32_t W,
                 const int64_t* inds /*[n] nullable*/, int64_t n, float* rays_o, float* rays_d, inr_stream_t s);

/* One training batch of a loader whose images are resident on the device, in ONE launch (round 6; replaces the body of
 * upstream's nerf/provider.py::NeRFDataset.collate [U] - torch.randint pixel draw, get_rays, image gather, mask gather -
 * for the loader of the reference's two training stages, /root/reference/README.md:58-66; mask format
 * /root/reference/Mask2Former_sample/match_seg.py:131-140).  Sample k of step `step` (0 <= step < 2^31) draws the pixel
 *   inds[k] = ((mix64(seed + 0x9E3779B97F4A7C15 * (step * 2^32 + k)) >> 32) * (H*W)) >> 32      (SplitMix64 finaliser)
 * of ONE H x W image (H*W < 2^31): a counter-based draw with replacement, reproducible from (seed, step) alone;
 * rays_o / rays_d [n,3] as inr_get_rays for those pixels of `pose` [4,4] (bit-identical); rgb [n,channels] =
 * image[inds] (image float [H*W, channels], channels 3 or 4; nullable -> not gathered); labels int64 [n] = mask[inds]
 * (mask int32 [H*W], nullable; ids >= num_instances have no logit and become -1, -1 stays the ignore label).           */
int inr_sample_training_batch(const float* pose /*[4,4]*/, float fx, float fy, float cx, float cy, int32_t H, int32_t W,
                              const float* image /*nullable*/, int32_t channels, const int32_t* mask /*nullable*/,
                              int32_t num_instances, int64_t seed, int64_t step, int64_t n, int64_t* inds /*[n]*/,
                              float* rays_o, float* rays_d, float* rgb /*[n,channels]*/, int64_t* labels /*[n]*/,
                              inr_stream_t s);

int inr_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb /*[6]*/,
                           int64_t N, float min_near, float* nears, float* fars, inr_stream_t s);
/* The same with per-ray labels (int64 [N]): a ray whose label equals `ignore_index` is reported as a miss (near = far =
 * FLT_MAX), so the training march gives it no samples.  No upstream counterpart: upstream marches and evaluates the
 * rays its mask loss then ignores (cross entropy with ignore_index -1 over matched masks,
 * /root/reference/Mask2Former_sample/match_seg.py:111-138 writes the -1s).  Used by Trainer(stage="instance").          */
int inr_near_far_from_aabb_skip(const float* rays_o, const float* rays_d, const float* aabb /*[6]*/, int64_t N,
                                float min_near, const int64_t* labels, int64_t ignore_index, float* nears, float* fars,
                                inr_stream_t s);

/* ---- occupancy helpers (replace raymarching.morton3D / morton3D_invert / packbits, a3) */
int inr_morton3D(const int32_t* coords /*[N,3]*/, int64_t N, int32_t* indices, inr_stream_t s);
int inr_morton3D_invert(const int32_t* indices, int64_t N, int32_t* coords /*[N,3]*/, inr_stream_t s);
/* bitfield[k] bit i = grid[8k+i] > thresh; n_bytes = cells / 8 */
int inr_packbits(const float* grid, int64_t n_bytes, float thresh, uint8_t* bitfield, inr_stream_t s);

/* ---- occupancy-grid update (replaces the tensor-op body of NeRFRenderer.update_extra_state, a3) -------------------
 * One cascade at a time; the grid is in Morton order.  inr_occ_cell_positions: query position of cell morton_idx[i]
 * (NULL = cell i) - centre (2c/(H-1) - 1) * (b - b/H) plus (2 noise - 1) * b/H, noise [n,3] in [0,1) or NULL.
 * inr_occ_update: grid = max(grid * decay, sigma * density_scale) where both >= 0 (cells at -1 stay out), for all
 * cells (morton_idx NULL, m == n_cells, sigma in Morton order) or the listed ones (tmp = scratch [n_cells]);
 * *mean_sum (double, caller zeroes it once per update) accumulates sum(max(grid, 0)).
 * inr_packbits_mean: packbits with thresh = min(*mean_sum / n_cells, density_thresh) formed on the device (no host
 * round trip between the update and the bitfield); mean_out (nullable) receives the mean; with n_counters > 0
 * stats_out = {mean, sum_k counters[k * counter_stride]} - the one 16-byte read-back of an update (mean density and
 * the sample totals of the last steps, which size the next steps' buffers).                                         */
/* mark_untrained_grid (a3): grid[cas][cell] = -1 for every cell (Morton order) that none of the B training cameras
 * sees - poses [B,4,4] camera-to-world row-major, pinhole (fx, fy, cx, cy): the cell centre x in camera frame
 * cam = R^T (x - t) is seen if z > 0, |x| < cx/fx z + 2 half, |y| < cy/fy z + 2 half (half = half a cell).        */
int inr_mark_untrained_grid(const float* poses, int32_t B, float fx, float fy, float cx, float cy, int32_t H,
                            int32_t cascade, float bound, float* grid /*[cascade, H^3]*/, inr_stream_t s);
int inr_occ_cell_positions(const int32_t* morton_idx, const float* noise, int64_t n, int32_t H, float cascade_bound,
                           float* xyz /*[n,3]*/, inr_stream_t s);
int inr_occ_update(float* grid /*[n_cells]*/, const float* sigma /*[m]*/, const int32_t* morton_idx /*[m] or NULL*/,
                   int64_t n_cells, int64_t m, float decay, float density_scale, float* tmp, double* mean_sum,
                   inr_stream_t s);
int inr_packbits_mean(const float* grid, int64_t n_cells /*all cascades*/, const double* mean_sum, float density_thresh,
                      uint8_t* bitfield, float* mean_out /*nullable*/, const int32_t* counters /*nullable*/,
                      int32_t n_counters, int32_t counter_stride /*in int32*/, double* stats_out /*[2]*/,
                      inr_stream_t s);
/* The steady-state cell choice of update_extra_state (after the first 16 updates; one cascade): n cells uniformly at
 * random + n picks, with replacement, among the cells with grid > 0 - upstream's randint coordinates and
 * nonzero(grid > 0)[randint] - in three launches.  u [4n] uniform in [0,1): slice draws [2n] then in-slice draws [2n]
 * (uniform half first).  morton_idx [2n] receives the cells (Morton indices), uniform half first; each half comes out
 * grouped by 4096 slices of the Morton range / of the occupied ranks (an i.i.d. sample drawn as multinomial slice
 * counts + uniform positions inside the slices), which is what lets the field kernel evaluate them at its coherent
 * rate.  No cell occupied: the occupied half is cell 0.  Exact restatement: oracle/occupancy.py::sample_cells.   */
int64_t inr_occ_sample_workspace_bytes(int64_t n_cells);
int inr_occ_sample_cells(const float* grid /*[n_cells] Morton order*/, int64_t n_cells, const float* u /*[4n]*/,
                         int64_t n, int32_t* morton_idx /*[2n]*/, void* workspace, inr_stream_t s);

/* ---- training march (replaces raymarching.march_rays_train, a4) ----------------------
 * Deterministic: sample slots are an exclusive scan of the per-ray counts in ray
 * order.  Call inr_march_rays_train_count first (fills counts/offsets and
 * counter[0] = total samples, counter[1] = N), size the outputs (M rows), then
 * inr_march_rays_train_write.  A ray whose offset + count > M is dropped (its
 * rays row is still written).  workspace: inr_march_workspace_bytes(N, sample_cap), 8-byte aligned.
 * sample_cap > 0: the count pass records, one bit per step candidate (the sequence t_{i+1} = t_i + dt(t_i) is
 * independent of the grid), which of each ray's first sample_cap candidates were emitted; the write pass
 * regenerates the sequence and emits at the set bits instead of walking the occupancy grid again (rays that
 * need more candidates are re-marched); pass the SAME workspace and sample_cap to the write call.  Results
 * are identical.                                                                                        */
/* 1 when inr_march_rays_train_write zero-fills the rows of xyzs / dirs / deltas that no ray owns by itself (the staged
 * wave-per-ray marcher does; the caller may then pass uninitialised buffers), 0 when the caller must zero them first. */
int inr_march_write_fills_unowned_rows(int64_t N, int32_t sample_cap, int32_t max_steps);
int64_t inr_march_workspace_bytes(int64_t N, int32_t sample_cap);
int inr_march_rays_train_count(const float* rays_o, const float* rays_d, const uint8_t* bitfield,
                               float bound, float dt_gamma, int32_t max_steps, int64_t N,
                               int32_t cascade, int32_t H, const float* nears, const float* fars,
                               const float* noises /*nullable*/, int32_t* rays /*[N,3]*/,
                               int32_t* counter /*[2]*/, void* workspace, int32_t sample_cap, inr_stream_t s);
int inr_march_rays_train_write(const float* rays_o, const float* rays_d, const uint8_t* bitfield,
                               float bound, float dt_gamma, int32_t max_steps, int64_t N,
                               int32_t cascade, int32_t H, int64_t M, const float* nears,
                               const float* fars, const float* noises /*nullable*/,
                               const int32_t* rays /*[N,3] from _count*/, float* xyzs /*[M,3]*/,
                               float* dirs /*[M,3]*/, float* deltas /*[M,2]*/, const void* workspace,
                               int32_t sample_cap, inr_stream_t s);

/* ---- fused full-frame inference path (replaces the alive-ray loop of NeRFRenderer.run_cuda, a5) ----
 * Same counting pass as training (inr_march_rays_train_count), then samples are written in the
 * PATCH-INTERLEAVED layout: rays in groups of 16 consecutive rays; inside a group
 *   slot(r, k) = rays[g0,1] + sum_i min(c_i, k) + #{ i < r : c_i > k }
 * (all k-th samples of a group adjacent).  A group whose slots do not fit in M is dropped whole.
 * inr_composite_rays_patch_forward composites that layout (per ray, in k order, stop at T < T_thresh;
 * same arithmetic as the sequential training compositing); weights [M] (nullable unless extra is given)
 * receives the per-sample weight, which the K-channel pass (one wave per ray) then uses.              */
int inr_march_rays_patch_write(const float* rays_o, const float* rays_d, const uint8_t* bitfield,
                               float bound, float dt_gamma, int32_t max_steps, int64_t N,
                               int32_t cascade, int32_t H, int64_t M, const float* nears,
                               const float* fars, const float* noises /*nullable*/,
                               const int32_t* rays /*[N,3] from inr_march_rays_train_count*/,
                               float* xyzs, float* dirs /*nullable if ray_ids*/, float* deltas,
                               const void* workspace, int32_t sample_cap, int32_t* ray_ids /*[M] nullable*/,
                               int32_t normalise /*1: xyzs = (p+bound)/(2 bound)*/, inr_stream_t s);
int inr_composite_rays_patch_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, int64_t N, int64_t M, float T_thresh,
                                     const float* extra /*[M,K] nullable*/, int32_t K,
                                     float* weights_sum, float* depth, float* image,
                                     float* extra_out /*[N,K]*/, float* weights /*[M]*/,
                                     uint64_t* skippable /*device, nullable: += samples of steps at which a whole
                                     16-ray group is already below T_thresh - what inr_nerf_render would skip*/,
                                     inr_stream_t s);
/* No upstream counterpart in the submodule's extension; the counterpart of the reference pipeline's project_3d_masks step
 * (voxel masks of nerf_rcnn/run_rcnn.py:652-666 projected into the training images that
 * Mask2Former_sample/match_seg.py:99-102 reads).  For the samples of a patch-interleaved frame (inr_march_rays_patch_write)
 * and their compositing weights (inr_composite_rays_patch_forward, weights != NULL): out[ray][k_base + i] =
 * sum over the ray's samples of w * bit_i(mask_words[cell(x)]), i < k_count <= 32; mask_words uint32 [W,L,H], bit i = mask
 * k_base + i contains the voxel; cell = floor((x - lo) / (hi - lo) * (W,L,H)), samples outside the box contribute nothing;
 * bbox: HOST array lo[3], hi[3]; out float [N, k_total].                                                               */
int inr_project_masks_patch(const float* xyzs, const float* weights, const int32_t* rays, int64_t N, int64_t M,
                            const uint32_t* mask_words, int32_t W, int32_t L, int32_t H, const float* bbox /*host*/,
                            int32_t k_total, int32_t k_base, int32_t k_count, float* out, inr_stream_t s);

/* ---- inference march/composite (replace raymarching.march_rays / composite_rays, a5) */
int inr_march_rays(int64_t n_alive, int32_t n_step, const int32_t* rays_alive, const float* rays_t,
                   const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                   int32_t max_steps, int32_t cascade, int32_t H, const uint8_t* bitfield,
                   const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                   inr_stream_t s);
int inr_composite_rays(int64_t n_alive, int32_t n_step, int32_t* rays_alive, float* rays_t,
                       const float* sigmas, const float* rgbs, const float* deltas,
                       float* weights_sum, float* depth, float* image, float T_thresh,
                       const float* extra /*[n_alive*n_step,K] nullable*/, float* extra_acc /*[N,K]*/,
                       int32_t K, inr_stream_t s);
/* order-preserving compaction of rays_alive >= 0; n_out (device int32[2]) = {survivors, n_alive};
 * `out` needs room for n_alive + inr_march_workspace_bytes(n_alive, 0)/4 int32 (list first, scratch after) */
int inr_compact_alive(const int32_t* rays_alive, int64_t n_alive, int32_t* out, int32_t* n_out,
                      inr_stream_t s);

/* ---- compositing for training (replaces raymarching.composite_rays_train fwd/bwd, a12/a13) ----
 * M = rows of the sample arrays: a ray with offset + count > M was dropped by the march writer and
 * composites to zero (its gradient rows are left untouched - the caller zero-initialises them - unless the backward
 * is given total_dev, the device int32 with the marcher's sample total: the rays' rows then tile [0, total) and the
 * backward writes every row of grad_sigmas / grad_rgbs in [0, M) itself, zeros where no ray owns the row).
 * weights [M] (nullable unless extra is given): receives the per-sample compositing weight
 * w = alpha * T (0 behind the termination point); the K-channel forward/backward use it.
 * sample_ray [M] (nullable; needs weights): receives, for every sample a ray owns, the row rays[n][0] of that ray's
 * outputs - what inr_instance_head_backward looks dL/d(rendered logits) up by.                       */
int inr_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                     const int32_t* rays, int64_t N, int64_t M, float T_thresh,
                                     const float* extra /*[M,K] nullable*/, int32_t K,
                                     float* weights_sum, float* depth, float* image,
                                     float* extra_out /*[N,K]*/, float* weights /*[M]*/,
                                     int32_t* sample_ray /*[M]*/, inr_stream_t s);
/* The K-channel half of the call above on its own: extra_out[rays[n][0]][ch] = sum over the ray's samples of
 * weights[i] * extra[i][ch] (weights from an earlier inr_composite_rays_train_forward; rows of dropped rays are zero). */
/* Optional cross-entropy epilogue (labels non-null): the wave that has just rendered a ray's K logits also forms that
 * ray's cross-entropy row against labels[row] (int64 [N]; ignore_index rows skipped; classes 0 .. n_classes-1, K may be
 * padded beyond) - grad_pix [N,K] = softmax - onehot for the kept rows (NOT yet divided by their number), and a second
 * small launch leaves loss_out[0..3] = {mean loss over the kept rows, 1 / kept, kept, labels out of range}; a label that
 * is neither ignore_index nor a class makes the loss NaN.  workspace: N * 16 bytes.  inr_instance_head_backward takes
 * grad_pix with loss_out + 1 as one of its scale pointers.                                                         */
int inr_composite_rays_extra_forward(const float* weights /*[M]*/, const float* extra /*[M,K]*/, const int32_t* rays,
                                     int64_t N, int64_t M, int32_t K, float* extra_out /*[N,K]*/,
                                     const int64_t* labels /*nullable*/, int32_t n_classes, int64_t ignore_index,
                                     float* grad_pix, void* workspace, float* loss_out, inr_stream_t s);
int inr_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                      const float* grad_extra_out /*nullable*/, const float* sigmas,
                                      const float* rgbs, const float* extra, const float* deltas,
                                      const int32_t* rays, const float* weights_sum,
                                      const float* image, const float* weights /*[M] from forward*/,
                                      int64_t N, int64_t M, float T_thresh, int32_t K,
                                      float* grad_sigmas, float* grad_rgbs /*both nullable: frozen NeRF*/,
                                      float* grad_extra /*[M,K] nullable*/, const int32_t* total_dev /*nullable*/,
                                      inr_stream_t s);

/* ---- hash grid (replaces gridencoder grid_encode_forward / _backward, a7/a8) -------- */
int inr_grid_encode_forward(const float* x /*[M,3]*/, const float* embeddings /*[T,F]*/,
                            const inr_grid_desc* desc /*host*/, int64_t M, float bound,
                            float* out /*[M,L*F]*/, inr_stream_t s);
/* grad_embeddings is ACCUMULATED into (caller zeroes it) */
int inr_grid_encode_backward(const float* x, const float* grad_out /*[M,L*F]*/,
                             const inr_grid_desc* desc /*host*/, int64_t M, float bound,
                             float* grad_embeddings /*[T,F]*/, inr_stream_t s);

/* same, processing the samples in the order given (int32 permutation of 0..M-1, nullable): the sum is
 * order-independent up to fp32 rounding; a spatial (Morton) order makes the atomics address-adjacent */
int inr_grid_encode_backward_ordered(const float* x, const float* grad_out, const int32_t* order,
                                     const inr_grid_desc* desc /*host*/, int64_t M, float bound,
                                     float* grad_embeddings /*[T,F]*/, inr_stream_t s);
/* Gradient with respect to the input coordinates (upstream's dy_dx path of gridencoder, taken when the positions
 * require grad): grad_x [M,3] = sum over levels and corners of grad_out . row * d(weight)/dx * scale_l / (2 bound);
 * zero for out-of-range points.  A.e. derivative of the trilinear interpolation (cells are piecewise linear).     */
int inr_grid_encode_backward_input(const float* x, const float* grad_out /*[M,L*F]*/, const float* embeddings,
                                   const inr_grid_desc* desc /*host*/, int64_t M, float bound, float* grad_x,
                                   inr_stream_t s);
/* The same scatter restricted to levels [level_lo, level_hi): a caller that all-reduces the table gradient over
 * several GPUs launches it in two or three level ranges and starts the collective on each row range
 * [offsets[level_lo], offsets[level_hi]) as soon as its launch is queued (nerf/network.py::_table_backward).      */
int inr_grid_encode_backward_levels(const float* x, const float* grad_out, const int32_t* order,
                                    const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                    int32_t level_lo, int32_t level_hi, inr_stream_t s);

/* Fixed-point form of the table-gradient scatter (round 6; OPT-IN - the product's default stays the fp32 atomics of
 * upstream's grid_encode_backward, see the end of this comment).  Why: on MI355X every atomic is executed by the memory-side unit, which takes int32 adds at 26.9 G
 * requests/s against 21.0 for fp32 (tools/micro/atomic_type_bench.hip) - the scatter is the largest kernel of a
 * training step - and integer sums do not depend on the order of arrival: the gradient becomes bit-reproducible.
 * fx_state: INR_GRID_FX_STATE_FLOATS device floats per table, zero-initialised by the caller once, then owned by these
 * three calls: [0,16) the scale of each level for the current step (a power of two; 0 = this level uses fp32 atomics),
 * [16,32) reference magnitudes, [32,48) the last step's largest |row gradient| per level, [48] fixed-point steps so far,
 * [49] near misses so far, [80,96) the largest fraction of the int32 range a row sum of each level has used so far, the
 * rest scratch.  A non-finite contribution to a fixed-point level (it has no int32
 * image) turns that level's WHOLE gradient into NaN in the finishing pass - as loud as the NaN rows fp32 atomics leave.
 * Per training step, on one stream:
 *   inr_grid_encode_backward_levels_fx   scatter of a level range; a level with a scale accumulates round(w g scale) as
 *                                        int32 bit patterns in grad_embeddings (zeroed by the caller), others fp32;
 *                                        fx_state NULL = inr_grid_encode_backward_levels
 *   inr_grid_grad_finish_fx              in place: int32 sums -> fp32 gradients (exact: the scale is a power of two) and
 *                                        the level's largest |gradient|; afterwards grad_embeddings is an ordinary fp32
 *                                        gradient (call it for every level range that was scattered, fp32 levels too)
 *   inr_grid_fx_update                   once per step after all ranges (sum_bits = 32): next step's scales = 2^floor(log2(2^30 /
 *                                        (headroom x reference))), reference = max(this step's max, 0.97 reference).
 * A level runs on fp32 atomics only while it has no reference: before its first step (the Python host primes the scales
 * with one extra scatter into a scratch buffer, so that no training step ever depends on the order of arrival) and
 * after an all-zero or non-finite gradient.  A near miss (a step that used more than 1/8 of the int32 range) is
 * counted; a row's FINAL sum that grows more than `headroom` times (128 in the product) against the reference - the
 * largest of the last ~50 steps - would wrap (intermediate overflow is harmless: int32 addition is modular); measured
 * peak use of the range over 3000 training steps: 0.06.  Quantisation: a row gradient is a multiple of headroom x
 * reference x 2^-30 (1.2e-7 of the level's recent largest at 128, ~5e-7 of a typical step's).
 * Why opt-in: a row whose gradient is below half a quantum gets none, and Adam with eps = 1e-15 (upstream's optimiser)
 * moves a row by a full lr step whatever its gradient's SIZE - rows fed only by weak samples learn at full speed under
 * fp32 atomics and not at all here.  Measured: identical PSNR / mIoU after 1500 + 1500 steps on the bench scene, but
 * held-out mIoU 0.75-0.82 against 0.80-0.86 on the shorter schedule of tests/test_pipeline_e2e.py
 * (profiles/r06_NOTES.txt 5).  Host switch: INR_FX_GRAD=1 / Trainer(fixed_point_grad=True).                          */
#define INR_GRID_FX_STATE_FLOATS 4192
int inr_grid_encode_backward_levels_fx(const float* x, const float* grad_out, const int32_t* order,
                                       const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                       int32_t level_lo, int32_t level_hi, float* fx_state /*nullable*/,
                                       inr_stream_t s);
int inr_grid_grad_finish_fx(float* grad_embeddings, const inr_grid_desc* desc /*host*/, int32_t level_lo,
                            int32_t level_hi, float* fx_state, inr_stream_t s);
int inr_grid_fx_update(float* fx_state, int32_t num_levels, float headroom, int32_t sum_bits /*32 | 64*/, inr_stream_t s);
/* The 64-bit form of the same (opt-in too): the sums live in a caller-owned int64 accumulator acc64 [T,2] (zero-initialised
 * once; the finishing pass zeroes it again), the scale is 2^32 larger (sum_bits = 64 in inr_grid_fx_update; headroom 1024 in
 * the product) - a quantum of ~2e-16 of the level's recent maximum, below anything Adam with eps 1e-15 can see: the
 * FAITHFUL order-independent form (bit-reproducible steps, no weakly supervised row frozen).  The unit takes 8-byte integer
 * adds at 23.6 G requests/s (fp32 21.0, int32 26.9): about the fp32 scatter's speed after the extra pass over the
 * accumulator.  grad_embeddings receives the fp32 gradient from inr_grid_grad_finish_fx64 (levels without a scale are
 * scattered into it with fp32 atomics directly, as above).                                                            */
int inr_grid_encode_backward_levels_fx64(const float* x, const float* grad_out, const int32_t* order,
                                         const inr_grid_desc* desc, int64_t M, float bound, float* grad_embeddings,
                                         int64_t* acc64 /*[T,2]*/, int32_t level_lo, int32_t level_hi, float* fx_state,
                                         inr_stream_t s);
int inr_grid_grad_finish_fx64(int64_t* acc64, float* grad_embeddings, const inr_grid_desc* desc /*host*/, int32_t level_lo,
                              int32_t level_hi, float* fx_state, inr_stream_t s);

/* ---- SH (replaces shencoder sh_encode_forward / _backward, a10) ------------------------ */
int inr_sh_encode_forward(const float* d /*[M,3]*/, int64_t M, int32_t degree, float* out, inr_stream_t s);
int inr_sh_encode_backward(const float* grad_out, const float* d, int64_t M, int32_t degree,
                           float* grad_d, inr_stream_t s);
/* degree-4 SH of N ray directions in the lane order of the fused field kernel: out[n][q][ks] = sh[4 ks + q] */
int inr_sh_table_q(const float* d /*[N,3]*/, int64_t N, float* out /*[N,16], 16-byte aligned*/, inr_stream_t s);

/* ---- fused field evaluation (replaces NeRFNetwork.forward/density + the fork's
 * instance head, a9/a13; the MFMA kernel the north star asks for) ---------------------------
 * Weights are nn.Linear [out,in] row-major fp32, no bias:
 *   sigma_w0[64,32] sigma_w1[16,64] color_w0[64,31] color_w1[64,64] color_w2[3,64]
 *   inst_w0[64,32] inst_w1[64,64] inst_w2[K,64] (K <= 64, K % 16 == 0: a caller with another K zero-pads the rows, as NeRFNetwork does)
 * inr_field_pack_* reorder them (on the host) into MFMA fragment order; the
 * packed buffer is then copied to the device by the caller.                              */
int64_t inr_nerf_packed_floats(void);
int inr_nerf_pack_weights(const float* sigma_w0, const float* sigma_w1, const float* color_w0,
                          const float* color_w1, const float* color_w2, float* packed /*host*/);
int64_t inr_instance_packed_floats(int32_t K);
int inr_instance_pack_weights(const float* w0, const float* w1, const float* w2, int32_t K,
                              float* packed /*host*/);
/* sigma[M] (= exp(h0) * density_scale), rgb[M,3] (nullable -> density only),
 * geo_feat[M,15] (nullable).  n_samples_dev: optional device int32 holding the
 * live row count (<= M); rows past it are skipped.                                           */
int inr_nerf_forward(const float* x, const float* d, int64_t M, const int32_t* n_samples_dev,
                     float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                     const float* packed /*device*/, float density_scale, float* sigma, float* rgb,
                     float* geo_feat, inr_stream_t s);
/* inr_nerf_forward (sigma + rgb) with upstream's `-O` numerics, opt-in: fp16 copy of the table (T x 2 binary16),
 * weights from inr_nerf_pack_weights_f16, one fp16 MFMA pass per MLP GEMM.  Used for the FROZEN NeRF of the instance
 * stage when Trainer(fp16=True) / NeRFNetwork.half_table + mlp_fp16 are set.  Not in the -DINR_MLP_FP32 build.   */
int inr_nerf_forward_fast(const float* x, const float* d, int64_t M, const int32_t* n_samples_dev, float bound,
                          const void* embeddings_half, const inr_grid_desc* desc /*host*/,
                          const float* packed_f16 /*device*/, float density_scale, float* sigma, float* rgb,
                          inr_stream_t s);
/* Fused-frame variant of inr_nerf_forward: x01 [M,3] already normalised by inr_march_rays_patch_write
 * (normalise = 1), directions given as a per-sample ray id + the per-ray table of inr_sh_table_q
 * ([N,4,4]: row q holds SH components q, 4+q, 8+q, 12+q).  Same results, ~100 VALU instructions per tile less. */
int inr_nerf_forward_table(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M,
                           float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                           const float* packed /*device*/, float density_scale, float* sigma, float* rgb,
                           inr_stream_t s);
/* The same launch on a HALF-PRECISION copy of the table (fp16 [T,2], 4 bytes per row; upstream's `-O` / fp16 storage,
 * here an opt-in for inference: NeRFNetwork.half_table).  Index arithmetic, blending and the MLPs are unchanged (fp32);
 * the table VALUES carry 11 significant bits, so outputs differ from the fp32 table's by ~1e-3 relative.  512 instead of
 * 1024 algorithmic bytes per sample.  embeddings_half: device pointer to T x 2 IEEE binary16 values.               */
int inr_nerf_forward_table_half(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M,
                                float bound, const void* embeddings_half, const inr_grid_desc* desc /*host*/,
                                const float* packed /*device*/, float density_scale, float* sigma, float* rgb,
                                inr_stream_t s);
/* The same launch with upstream's `-O` numerics (opt-in, inference: NeRFNetwork.mlp_fp16): the MLP GEMMs take ONE fp16
 * MFMA pass - weights and activations rounded to fp16, fp32 accumulation - instead of the three-pass bf16 split that
 * keeps the default path fp32-class; 2^-12 relative per operand.  `packed` must come from inr_nerf_pack_weights_f16
 * (same size and fragment layout as inr_nerf_pack_weights, fp16 values in the head slots).  embeddings: the fp32
 * table, or the fp16 copy when table_is_half != 0.  Not available in the -DINR_MLP_FP32 build.                     */
int inr_nerf_pack_weights_f16(const float* sigma_w0, const float* sigma_w1, const float* color_w0,
                              const float* color_w1, const float* color_w2, float* packed /*host*/);
int inr_nerf_forward_table_fast(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M,
                                float bound, const void* embeddings, int32_t table_is_half,
                                const inr_grid_desc* desc /*host*/, const float* packed /*device*/,
                                float density_scale, float* sigma, float* rgb, inr_stream_t s);
/* Sliced variant of inr_nerf_forward_table (round 5; same results bit for bit; fp32 table, 16 levels, hashed fine
 * levels): the three finest levels (13..15) are evaluated first, one level at a time over all samples - a hashed level
 * is 4 MiB, exactly one XCD's L2, and where those levels are finer than the spacing of a frame's samples nothing but a
 * cache that holds the whole level can serve their gathers (tools/micro/level_xcd_bench.hip: 259 against 69 G lines/s)
 * - into `fine_ws` (device, caller-owned, >= inr_nerf_forward_table_sliced_workspace_bytes(M) bytes: 24 bytes per
 * sample); the fused kernel then gathers the other thirteen levels.  Two launches on `s`.  Pays where the finest levels
 * have no locality at all (a scene filling a bound >= 4 volume: 30.2 -> 27.5 ms per 141 M samples); slower elsewhere -
 * NeRFNetwork.frame_slices = "auto" measures both paths and keeps the faster.                                          */
int64_t inr_nerf_forward_table_sliced_workspace_bytes(int64_t M);
int inr_nerf_forward_table_sliced(const float* x01, const int32_t* ray_ids, const float* sh_table_q, int64_t M,
                                  float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                                  const float* packed /*device*/, float density_scale, float* sigma, float* rgb,
                                  float* fine_ws, inr_stream_t s);
/* Training path of the NeRF field (a9 under autograd): device-packed weights (forward layout of
 * inr_nerf_pack_weights + the transposed sections of the backward), a forward that also stores the activations
 * (enc [M,32], h1 [M,64], so [M,16] = raw sigma-net output, cin [M,32] = colour-net input with a zero pad column,
 * c1, c2 [M,64]) and ONE backward launch from (dL/dsigma [M], dL/drgb [M,3]) to
 * grad_o [M,4] (rgb logits, 3 live), grad_zc2, grad_zc1 [M,64], grad_so [M,16], grad_zh1 [M,64], grad_enc [M,32].
 * sigma here is exp(so[:,0]) (no density_scale); pass the scale the caller applies afterwards as density_scale = 1.
 * Weight gradients: inr_linear_wgrad(c2, grad_o), (c1, grad_zc2), (cin, grad_zc1), (h1, grad_so), (enc, grad_zh1). */
int64_t inr_nerf_bwd_packed_floats(void);
int inr_nerf_pack_weights_device(const float* sigma_w0, const float* sigma_w1, const float* color_w0,
                                 const float* color_w1, const float* color_w2, float* packed_fwd,
                                 float* packed_bwd, inr_stream_t s);
int inr_nerf_forward_train(const float* x, const float* d, int64_t M, float bound, const float* embeddings,
                           const inr_grid_desc* desc /*host*/, const float* packed_fwd, float* sigma, float* rgb,
                           float* enc, float* h1, float* so, float* cin, float* c1, float* c2, inr_stream_t s);
int inr_nerf_backward(const float* grad_sigma, const float* grad_rgb, const float* rgb, const float* so,
                      const float* h1, const float* c1, const float* c2, int64_t M, float density_scale,
                      const float* packed_bwd, float* grad_o, float* grad_zc2, float* grad_zc1, float* grad_so,
                      float* grad_zh1, float* grad_enc, inr_stream_t s);
/* The NeRF field of a training step in two launches per direction (round 3; the NeRF-stage twin of
 * inr_instance_forward_enc / inr_instance_head_backward).  Forward: as inr_nerf_forward_train but only the encoder
 * output [M,32] is kept.  Backward: ONE launch from (dL/dsigma [M], dL/drgb [M,3]) to dL/denc [M,32] and the five
 * weight gradients - grad_ws0 [64,32], grad_ws1 [16,64], grad_wc0 [64,31] (the colour net's 31 inputs, pitch 31), grad_wc1 [64,64],
 * grad_wc2 [16,64] (rows 0..2 live), written, not accumulated: the forward is recomputed from enc and the view
 * directions d [M,3] with packed_fwd, the input-gradient chain is inr_nerf_backward's, and the weight gradients are
 * accumulated on the fp32 matrix cores with the tiles transposed through LDS.  Replaces inr_nerf_backward and five
 * inr_linear_wgrad calls and the 1088 B per sample of saved activations.  workspace:
 * inr_instance_head_workspace_bytes() bytes.                                                                      */
int inr_nerf_forward_enc(const float* x, const float* d, int64_t M, float bound, const float* embeddings,
                         const inr_grid_desc* desc /*host*/, const float* packed_fwd, float* sigma, float* rgb,
                         float* enc, inr_stream_t s);
int inr_nerf_head_backward(const float* enc, const float* d, const float* grad_sigma, const float* grad_rgb, int64_t M,
                           float density_scale, const float* packed_fwd, const float* packed_bwd, float* grad_enc,
                           void* workspace, float* grad_ws0, float* grad_ws1, float* grad_wc0, float* grad_wc1,
                           float* grad_wc2, float* zero_buf /*nullable*/, int64_t zero_floats, inr_stream_t s);
/* Training path of the instance field (a13): weights are packed ON THE DEVICE every step (forward layout as
 * inr_instance_pack_weights, plus the transposed sections the input-gradient kernel uses); the forward also
 * stores the encoder output [M,32] and both hidden activations [M,64]; inr_instance_backward turns
 * dL/dlogits [M,K] into dL/dz2, dL/dz1 [M,64] (ReLU-masked) and dL/denc [M,32] in one launch.  Weight gradients
 * are inr_linear_wgrad(h2, dlogits), (h1, dz2), (enc, dz1); the table gradient is inr_grid_encode_backward(denc). */
int64_t inr_instance_bwd_packed_floats(void);
int inr_instance_pack_weights_device(const float* w0 /*[64,32]*/, const float* w1 /*[64,64]*/,
                                     const float* w2 /*[K,64]*/, int32_t K, float* packed_fwd, float* packed_bwd,
                                     inr_stream_t s);
int inr_instance_forward_train(const float* x, int64_t M, float bound, const float* embeddings,
                               const inr_grid_desc* desc /*host*/, const float* packed_fwd, int32_t K,
                               float* logits, float* enc, float* h1, float* h2, inr_stream_t s);
int inr_instance_backward(const float* grad_logits, int32_t K, const float* h1, const float* h2, int64_t M,
                          const float* packed_bwd, float* grad_z2, float* grad_z1, float* grad_enc,
                          inr_stream_t s);
/* The instance head of the instance stage in two launches per direction (round 3).  Forward: as
 * inr_instance_forward_train, but only the encoder output [M,32] is kept (n_samples_dev, nullable: device int32 with
 * the number of live rows; rows beyond it are not evaluated).  Backward: ONE launch from dL/d(rendered logits)
 * grad_pix [N,K] to dL/denc [M,32] (zero for rows >= *n_samples_dev) and the three weight gradients
 * grad_w0 [64,32], grad_w1 [64,64], grad_w2 [K,64] (written, not accumulated): per sample g = weights[m] *
 * grad_pix[sample_ray[m]] (the K-channel compositing backward, weights detached), the hidden layers are recomputed
 * from enc with packed_fwd, the input-gradient chain uses packed_bwd, and dW += g^T h on the fp32 matrix cores with
 * the tiles transposed through LDS - replaces inr_composite_rays_train_backward (K channels), inr_instance_backward
 * and three inr_linear_wgrad calls, and ~3 KB of HBM traffic per sample.  scale_a, scale_b: device scalars that
 * multiply grad_pix (1 / kept rows and dL/dloss of the fused cross entropy; NULL = 1).  zero_buf (nullable): zero_floats
 * floats that the launch zero-fills on the side - the table-gradient buffer of the scatter that follows.  workspace:
 * inr_instance_head_workspace_bytes() bytes.  The table gradient stays inr_grid_encode_backward(grad_enc).      */
int inr_instance_forward_enc(const float* x, int64_t M, const int32_t* n_samples_dev, float bound,
                             const float* embeddings, const inr_grid_desc* desc /*host*/, const float* packed_fwd,
                             int32_t K, float* logits, float* enc, inr_stream_t s);
int64_t inr_instance_head_workspace_bytes(void);
int inr_instance_head_backward(const float* enc, const float* weights, const int32_t* sample_ray,
                               const float* grad_pix, int32_t K, int64_t N, int64_t M, const int32_t* n_samples_dev,
                               const float* scale_a /*device scalar, nullable*/, const float* scale_b /*same*/,
                               const float* packed_fwd, const float* packed_bwd, float* grad_enc, void* workspace,
                               float* grad_w0, float* grad_w1, float* grad_w2, float* zero_buf /*nullable*/,
                               int64_t zero_floats, inr_stream_t s);
/* rgb-sigma lattice extraction (the step after the path that feeds NeRF-RCNN: /root/reference/nerf_rcnn/datasets.py:766-792
 * reads the result): out[m] = (mean over n_dirs fixed view directions of rgb(x_m, dir), raw density logit of x_m) -
 * one gather + one sigma-net pass per point, the colour net once per direction.  sh_dirs [n_dirs,16] = degree-4 SH rows
 * of the directions (inr_sh_encode_forward), n_dirs <= 8; out float [M,4], 16-byte aligned.                            */
int inr_nerf_forward_dirs(const float* x, int64_t M, float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                          const float* packed /*device*/, const float* sh_dirs, int32_t n_dirs, float* out, inr_stream_t s);
/* The same for a [W, L, H] lattice given by its coordinate axes (x = ax_w[iw], y = ax_l[il], z = ax_h[ih], clamped to
 * [-bound, bound]; the rgb-sigma extraction of SURVEY f1 / BASELINE configs[4]): no point tensor, and the kernel walks the
 * lattice in runs of 16 along W - the table's fastest row index - instead of the writer's h-fastest order (2.4x faster:
 * the kernel is gather-bound).  logit_min: lower clamp of the density logit (channel 3; the writer uses log(1e-30),
 * -INFINITY for none).  out float [W, L, H, 4], 16-byte aligned.                                                        */
int inr_nerf_forward_lattice(const float* ax_w, const float* ax_l, const float* ax_h, int32_t W, int32_t L, int32_t H,
                             float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                             const float* packed /*device*/, const float* sh_dirs, int32_t n_dirs, float logit_min,
                             float* out, inr_stream_t s);
int inr_instance_forward(const float* x, int64_t M, const int32_t* n_samples_dev, float bound,
                         const float* embeddings, const inr_grid_desc* desc /*host*/,
                         const float* packed /*device*/, int32_t K, float* logits /*[M,K]*/,
                         inr_stream_t s);

/* ---- 3-D RoIAlign ("next" row f2; replaces roi_align.roi_align.roi_align_3d, the one FFI call in the
 * reference tree: /root/reference/nerf_rcnn/model/utils.py:604-609).  torchvision roi_align semantics
 * (aligned=False, adaptive ceil(roi/out) sampling grid, average) on three axes: x<->W, y<->L, z<->H.
 * input [N,C,W,L,H], rois [K,6] = (x1,y1,z1,x2,y2,z2) in input-scale units, roi_inds int32 [K],
 * out [K,C,out_w,out_l,out_h].  backward ACCUMULATES into grad_input (caller zeroes it).          */
/* No reference counterpart.  Which kernels the two calls below use: 0 (default) the separable ones (per-axis weight
 * tables built once per RoI in LDS, z -> y -> x contraction, 4 channels per lane) whenever the tables fit the LDS
 * window, 1 one lane per output element (the torchvision kernel shape; also the fallback), 2 separable or INR_EINVAL,
 * 3 separable without the workspace form of the backward (below).
 * Same sample geometry in both; sums differ by fp32 rounding.  Process-wide.                                        */
int inr_roi_align_3d_set_mode(int32_t mode);
int inr_roi_align_3d_forward(const float* input, const float* rois, const int32_t* roi_inds, int32_t N,
                             int32_t C, int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w,
                             int32_t out_l, int32_t out_h, float spatial_scale, float* out, inr_stream_t s);
int inr_roi_align_3d_backward(const float* grad_out, const float* rois, const int32_t* roi_inds, int32_t N,
                              int32_t C, int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w,
                              int32_t out_l, int32_t out_h, float spatial_scale, float* grad_input,
                              inr_stream_t s);

/* No reference counterpart.  The backward with a caller-owned workspace: the separable backward is bound by the
 * memory-side atomic unit (requests of up to 64 bytes, ~21 G/s whatever they carry), and a region row of the gradient's
 * own layout fills a request with ~6 floats.  This form accumulates into a channels-fastest scratch volume
 * (workspace: N*C*W*L*H floats, zeroed by the call) - one full 64-byte request per voxel and 16 channels - and writes
 * grad_input with a transposing copy: grad_input is OVERWRITTEN (no zero fill by the caller, no accumulation).
 * _workspace_bytes returns 0 where the form does not apply (C not a multiple of 16, out_l*out_h > 256, tables beyond
 * the LDS window, modes 1 and 3 of inr_roi_align_3d_set_mode): the caller then uses inr_roi_align_3d_backward.
 * Mode 3 (added with it) = separable kernels, accumulation in place (the round-4 form) - for A/B tests.             */
int64_t inr_roi_align_3d_backward_workspace_bytes(int32_t N, int32_t C, int32_t W, int32_t L, int32_t H, int64_t K,
                                                  int32_t out_w, int32_t out_l, int32_t out_h);
/* Which backward a caller should take (cost model, round 6): 1 = the workspace form is available AND expected to be
 * faster - it trades fuller atomic requests (16 channels of a voxel instead of ~6 floats of a row) for three extra passes
 * over the N*C*V volume and a per-RoI set-up that its 16-channel groups amortise worse; linear time models of both forms
 * fitted to 28 measured shapes (profiles/r06_roialign_bwd_shapes.txt).  covered_voxels: voxels inside the RoIs' regions,
 * summed over the RoIs, if known; < 0 = unknown (the lower bound K * min(bins, V) is then used, which biases towards the
 * in-place form; mode 2 of inr_roi_align_3d_set_mode answers 1 wherever the form exists).  Pure host arithmetic.    */
int inr_roi_align_3d_backward_prefers_workspace(int32_t N, int32_t C, int32_t W, int32_t L, int32_t H, int64_t K,
                                                int32_t out_w, int32_t out_l, int32_t out_h, int64_t covered_voxels);
int inr_roi_align_3d_backward_ws(const float* grad_out, const float* rois, const int32_t* roi_inds, int32_t N,
                                 int32_t C, int32_t W, int32_t L, int32_t H, int64_t K, int32_t out_w,
                                 int32_t out_l, int32_t out_h, float spatial_scale, float* grad_input,
                                 void* workspace, int64_t workspace_bytes, inr_stream_t s);

/* Whole-ray rendering with early termination (inference, patch-interleaved layout): field evaluation and alpha
 * compositing in one launch; a 16-ray group stops being evaluated once all its rays are below T_thresh (what the
 * alive-ray loop of NeRFRenderer.run_cuda achieves, a5, without host round trips).  Same results as
 * inr_nerf_forward + inr_composite_rays_patch_forward up to fp32 rounding.  weights [M] nullable (w per sample,
 * 0 when skipped); evaluated: device uint64 [33], ZEROED BY THE CALLER - [0] receives the number of samples evaluated,
 * [1..32] are the cursors of the launch's dynamic group schedule (scratch).                                        */
int inr_nerf_render(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d /*[N,3]*/,
                    int64_t N, int64_t M, float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                    const float* packed /*device*/, float density_scale, float T_thresh, float* weights_sum,
                    float* depth, float* image, float* weights, uint64_t* evaluated, int32_t x_is_01 /* xyzs are the
                    normalised coordinates of the patch writer's table feed */, inr_stream_t s);
/* The same launch with upstream's `-O` numerics (opt-in: NeRFNetwork.half_table + mlp_fp16): fp16 copy of the table,
 * weights from inr_nerf_pack_weights_f16, one fp16 MFMA pass per MLP GEMM.  Not in the -DINR_MLP_FP32 build.      */
int inr_nerf_render_fast(const float* xyzs, const float* deltas, const int32_t* rays, const float* rays_d, int64_t N,
                         int64_t M, float bound, const void* embeddings_half, const inr_grid_desc* desc /*host*/,
                         const float* packed_f16 /*device*/, float density_scale, float T_thresh, float* weights_sum,
                         float* depth, float* image, float* weights, uint64_t* evaluated, int32_t x_is_01,
                         inr_stream_t s);

/* Instance logits rendered in place (inference, patch-interleaved layout): extra_out[ray][ch] =
 * sum_k weights[slot(ray,k)] * logits(xyzs[slot(ray,k)])[ch]; the [M,K] logits never exist in memory.
 * xyzs/weights [M] in the patch-interleaved layout (inr_march_rays_patch_write / the weights output of
 * inr_composite_rays_patch_forward), rays [N,3] from the count pass, extra_out [N,K].  x_is_01 != 0: xyzs holds
 * the normalised coordinates (x + bound) / (2 bound) the patch writer emits with normalise = 1.          */
int inr_instance_render(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M,
                        float bound, const float* embeddings, const inr_grid_desc* desc /*host*/,
                        const float* packed /*device*/, int32_t K, float* extra_out, int32_t x_is_01,
                        uint64_t* cursors /*device [32], ZEROED BY THE CALLER: the launch's dynamic group schedule*/,
                        inr_stream_t s);
/* The same launch with upstream's `-O` numerics (opt-in: NeRFNetwork.half_table + mlp_fp16): embeddings_half is the fp16
 * copy of the instance table (T x 2 binary16), packed_f16 comes from inr_instance_pack_weights_f16 (same size and
 * fragment layout as inr_instance_pack_weights, fp16 values in the head slots); one fp16 MFMA pass per MLP GEMM.
 * Not available in the -DINR_MLP_FP32 build.                                                                      */
int inr_instance_pack_weights_f16(const float* w0, const float* w1, const float* w2, int32_t K, float* packed /*host*/);
int inr_instance_render_fast(const float* xyzs, const int32_t* rays, const float* weights, int64_t N, int64_t M,
                             float bound, const void* embeddings_half, const inr_grid_desc* desc /*host*/,
                             const float* packed_f16 /*device*/, int32_t K, float* extra_out, int32_t x_is_01,
                             uint64_t* cursors /*device [32], zeroed by the caller*/, inr_stream_t s);

/* ---- weight gradient of the tiny bias-free MLP layers (replaces the BLAS call autograd makes for
 * nn.Linear in NeRFNetwork, a9/a13):  grad_w[o][i] += sum_m grad_y[m][o] * x[m][i],  n_in, n_out <= 64.
 * x [M, n_in], grad_y [M, n_out], grad_w [n_out, n_in] is ACCUMULATED into (caller zeroes it).
 * workspace: inr_linear_wgrad_workspace_bytes() bytes of scratch (per-workgroup partial sums; two passes, no
 * atomics - same-address float atomics from hundreds of workgroups were the slowest part of this op).  */
int64_t inr_linear_wgrad_workspace_bytes(void);
int inr_linear_wgrad(const float* x, const float* grad_y, int64_t M, int32_t n_in, int32_t n_out,
                     float* grad_w, void* workspace, inr_stream_t s);

/* ---- optimiser (replaces the Trainer's torch.optim.Adam sweep over the table, a15) ------ */
int inr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int32_t step, float grad_scale,
                  inr_stream_t s);
/* The same update for up to 16 tensors in one launch; host arrays of device pointers / sizes / learning rates. */
int inr_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                        float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                        float eps, int32_t step, float grad_scale, inr_stream_t s);
/* The same with the parameter EMA of upstream's Trainer (torch_ema: shadow += w * (param - shadow) after every step)
 * applied while the new parameter is in registers: ema_shadows[t] (device, nullable per tensor), w = ema_weight. */
int inr_adam_ema_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* numels, const float* lrs, float beta1, float beta2,
                            float eps, int32_t step, float grad_scale, float* const* ema_shadows, float ema_weight,
                            inr_stream_t s);
/* The same with the step-dependent scalars in DEVICE memory (hyper_dev = [eps_t, lr_t[0..15]], written by
 * inr_adam_set_hyper on the stream - the values travel as kernel arguments, no host buffer has to stay alive): a
 * captured hipGraph of a training step can then be replayed with a new learning rate and bias correction. */
int inr_adam_set_hyper(const float* lrs /*host*/, int32_t n_tensors, float beta1, float beta2, float eps,
                       int32_t step, float* hyper_dev /*[17]*/, inr_stream_t s);
int inr_adam_step_multi_dev(int32_t n_tensors, float* const* params, const float* const* grads,
                            float* const* exp_avgs, float* const* exp_avg_sqs, const int64_t* numels,
                            const float* hyper_dev, float beta1, float beta2, float grad_scale, inr_stream_t s);
/* No upstream counterpart.  n <= 8 device-to-device copies in one launch (the fixed input buffers of a captured training
 * step); dsts / srcs / nbytes are HOST arrays, addresses and byte counts multiples of 4.                                */
int inr_copy_multi(int32_t n, void* const* dsts, const void* const* srcs, const int64_t* nbytes, inr_stream_t s);
/* The same two calls with the parameter EMA inside the captured sweep (no upstream counterpart; torch_ema's update folded
 * into the optimiser's launch as inr_adam_ema_step_multi does for eager steps): hyper_dev holds 18 floats, the last one
 * the EMA weight 1 - decay_t of the step, written by inr_adam_set_hyper_ema; ema_shadows[t] nullable per tensor.      */
int inr_adam_set_hyper_ema(const float* lrs /*host*/, int32_t n_tensors, float beta1, float beta2, float eps, int32_t step,
                           float ema_weight, float* hyper_dev, inr_stream_t s);
int inr_adam_ema_step_multi_dev(int32_t n_tensors, float* const* params /*host array of device pointers*/,
                                const float* const* grads, float* const* exp_avgs, float* const* exp_avg_sqs,
                                const int64_t* numels /*host*/, const float* hyper_dev, float beta1, float beta2,
                                float grad_scale, float* const* ema_shadows /*host array*/, inr_stream_t s);
/* Mask-supervised loss of the instance stage (a13; replaces F.cross_entropy(logits, labels, ignore_index) + its
 * backward in the fork's instance trainer): loss[0] = mean over rows with 0 <= label < K and label != ignore_index of
 * logsumexp(row) - row[label]; grad_logits [N,K] = d loss / d logits (zero rows where ignored); acc = 128 floats of
 * scratch (per-workgroup partial sums: no atomics), 8-byte aligned.  K <= 64.  NaN loss when no row is kept (torch's
 * mean over an empty set) AND when a label is neither ignore_index nor in [0, K) (torch asserts on the device for such
 * a label; the row is never dropped silently).                                                                    */
int inr_cross_entropy(const float* logits /*[N,K]*/, const int64_t* labels /*[N]*/, int64_t N, int32_t K,
                      int64_t ignore_index, float* grad_logits, float* acc, float* loss, inr_stream_t s);
/* Tail of NeRFRenderer.run_cuda (a14): image_out = image + (1 - weights_sum) * bg;
 * depth_out = clamp(depth [+ t0 * weights_sum] - near, 0) / (far - near); outputs may alias the inputs.  t0 (nullable)
 * = start parameter of every ray: upstream's inference compositing accumulates depth over the ABSOLUTE ray parameter
 * (composite_rays: t = rays_t), its training compositing over t counted from the first step (composite_rays_train:
 * t = 0); the one-pass inference kernels here count like the latter, so inference passes t0 (= nears without jitter).
 * No autograd: callers that need gradients through the image use torch ops. */
int inr_finish_rays(const float* image /*[N,3]*/, const float* depth /*[N]*/, const float* weights_sum,
                    const float* nears, const float* fars, const float* t0 /*[N] nullable*/, float bg_r, float bg_g,
                    float bg_b, int64_t N, float* image_out, float* depth_out, inr_stream_t s);

/* The training tail of the NeRF stage in one launch (Trainer.train_step, stage "nerf", default criterion): the blend and
 * depth of inr_finish_rays (training: no t0), loss[0] = mean((image_out - target)^2) over the 3N channels - upstream's
 * ``criterion(pred, gt).mean()`` with MSELoss(reduction='none') -, and the gradients of that loss:
 * grad[0, 3N) = d loss / d image, grad[3N, 4N) = d loss / d weights_sum (the blend's -sum_c g_c * bg_c).
 * bg_rays (nullable, [N,3]) replaces the uniform colour per ray.  One workgroup, fixed summation order:
 * 0 < N <= INR_FINISH_MSE_MAX_RAYS; larger batches use inr_finish_rays / torch ops. */
#define INR_FINISH_MSE_MAX_RAYS 65536
int inr_finish_rays_mse(const float* image /*[N,3]*/, const float* depth /*[N]*/, const float* weights_sum,
                        const float* nears, const float* fars, float bg_r, float bg_g, float bg_b,
                        const float* bg_rays /*[N,3] nullable*/, const float* target /*[N,3]*/, int64_t N,
                        float* image_out, float* depth_out, float* grad /*[4N]*/, float* loss /*[1]*/, inr_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* INR_H */
