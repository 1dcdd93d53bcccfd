/*
 * splat_hip.h -- C ABI of libsplat_hip.so, the MI355X (gfx950) drop-in for the native half of the
 * Splat-SLAM mapping hot path.  Torch-free: plain device pointers, sizes and a stream handle.
 *
 * Which reference interface each entry point replaces (paths relative to /root/reference):
 *
 *   sgr_forward            -> diff_gaussian_rasterization._C.rasterize_gaussians, reached through
 *                             GaussianRasterizer.forward at
 *                             thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141
 *                             (settings built at :58-72).  The native source is the un-vendored submodule
 *                             thirdparty/diff-gaussian-rasterization-w-pose (.gitmodules:4-6, README.md:88-92).
 *   sgr_backward           -> diff_gaussian_rasterization._C.rasterize_gaussians_backward, triggered by
 *                             loss.backward() at src/mapper.py:329,490,699.
 *   sgr_saved_bytes,
 *   sgr_scratch_bytes      -> the resizeFunctional geom/binning/image buffer callbacks of the same module.
 *   sgr_mapping_loss       -> get_loss_mapping / get_loss_mapping_rgbd, thirdparty/monogs/utils/slam_utils.py:71-105
 *   sgr_adam_step          -> torch.optim.Adam(eps=1e-15) over the GaussianModel groups,
 *                             thirdparty/gaussian_splatting/scene/gaussian_model.py:264-313, stepped at
 *                             src/mapper.py:352,557,703
 *   sgr_activate,
 *   sgr_gaussian_adam_step,
 *   sgr_gaussian_adam_shard -> the activation getters (exp / normalize / sigmoid, gaussian_model.py:76-101) and the Adam step of
 *                             the five per-Gaussian groups incl. the isotropy regulariser of src/mapper.py:487-489
 *   sgr_masked_adam        -> the keyframe (exposure) optimiser of src/mapper.py:1096-1111, stepped at :561
 *   sgr_map_views          -> the per-view body of Mapper.map (src/mapper.py:426-490): render, loss, backward for <= 16 views
 *   sgr_map_step           -> one iteration of Mapper.map / initialize_map / final_refine (src/mapper.py:303-353, 414-568, 656-708)
 *   sgr_map_run            -> a run of such iterations between two densify / reset points (the `for` loops at :304, :414, :656)
 *   sgr_deform_points      -> Mapper.update_mapping_points, src/mapper.py:154-255
 *   sgr_keep_list,
 *   sgr_gather_rows        -> prune_points / _prune_optimizer and the row selects of densify_and_clone,
 *                             thirdparty/gaussian_splatting/scene/gaussian_model.py:519-557, 690-719
 *   sgr_query*, sgr_profile_*,
 *   sgr_set_option         -> (no reference counterpart) capacity protocol, work counters, per-kernel HIP-event timing, options
 *   sknn_dist2             -> simple_knn._C.distCUDA2, thirdparty/gaussian_splatting/scene/gaussian_model.py:18,194-200
 *   se3_*                  -> lietorch SE3 ops used on the mapping path, thirdparty/glorie_slam/depth_video.py:327-330
 *                             (SE3(pose).inv().matrix()), and the tau convention of
 *                             thirdparty/monogs/utils/pose_utils.py:66-98.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 / int32 unless it says "host";
 *   - [N,C] arrays are row-major; images are CHW; 4x4 matrices are 16 floats in the layout the reference
 *     passes them (transposed / row-vector convention: camera_utils.py:94-104, mapper.py:841-850);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it;
 *   - functions return 0 on success, a negative SgrStatus otherwise; sgr_last_error() gives the text;
 *   - the library never allocates device memory: the caller owns inputs, outputs and both workspaces.
 */
#ifndef SPLAT_HIP_H_
#define SPLAT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_ABI_VERSION 10

typedef enum SgrStatus {
  SGR_OK = 0,
  SGR_ERR_INVALID = -1,     /* bad argument (null pointer, inconsistent option set, ...)            */
  SGR_ERR_WORKSPACE = -2,   /* saved / scratch workspace smaller than sgr_*_bytes() demands          */
  SGR_ERR_CAPACITY = -3,    /* more (tile, Gaussian) pairs than `capacity`; *num_rendered_host = need */
  SGR_ERR_HIP = -4          /* a HIP runtime call failed                                              */
} SgrStatus;

/* The 13 fields of GaussianRasterizationSettings (gaussian_renderer/__init__.py:58-72) + array extents. */
typedef struct SgrSettings {
  int32_t num_gaussians;       /* N */
  int32_t image_height;
  int32_t image_width;
  int32_t sh_degree;           /* active degree (0..3) */
  int32_t sh_coeffs;           /* M: `shs` is [N, M, 3]; ignored when colors_precomp is given */
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  int32_t prefiltered;
  int32_t debug;
  const float* bg;             /* [3]  */
  const float* viewmatrix;     /* [16] world_view_transform  (W2C transposed) */
  const float* projmatrix;     /* [16] full_proj_transform   */
  const float* projmatrix_raw; /* [16] projection_matrix     */
  const float* campos;         /* [3]  */
} SgrSettings;

/* Arguments of GaussianRasterizer.forward (gaussian_renderer/__init__.py:130-141). NULL = "None". */
typedef struct SgrInputs {
  const float* means3D;        /* [N,3] */
  const float* opacities;      /* [N]   */
  const float* shs;            /* [N,M,3] or NULL */
  const float* colors_precomp; /* [N,3]   or NULL */
  const float* scales;         /* [N,3]   or NULL */
  const float* rotations;      /* [N,4]   or NULL (w,x,y,z; used as given, not re-normalised) */
  const float* cov3D_precomp;  /* [N,6]   or NULL */
} SgrInputs;

/* The 5-tuple returned at gaussian_renderer/__init__.py:130. */
typedef struct SgrOutputs {
  float* color;                /* [3,H,W] */
  float* depth;                /* [1,H,W] */
  float* opacity;              /* [1,H,W] */
  int32_t* radii;              /* [N] */
  int32_t* n_touched;          /* [N] */
} SgrOutputs;

typedef struct SgrWorkspace {
  void* saved;                 /* lives from forward until the matching backward (one per forward call) */
  size_t saved_bytes;
  void* scratch;               /* transient inside one call; may be shared by calls on the same stream   */
  size_t scratch_bytes;
  int64_t capacity;            /* max (tile, Gaussian) pairs the workspaces were sized for               */
  int32_t counters_clean;      /* != 0: the per-tile pair counters inside `saved` are zero -- every completed forward
                                  (same N, H, W, capacity) leaves them so; 0 for a fresh / foreign block: the library
                                  then spends one extra launch zeroing them */
  int32_t max_list_hint;       /* > 0: the LONGEST per-8x8-tile list the caller has measured for these cameras (header word 10 of a
                                  recent forward, sgr_query_header): picks the sort build of the compositing kernels -- a
                                  deterministic function of a measurement.  0: guessed from `capacity` / tiles */
} SgrWorkspace;

typedef struct SgrGradOutputs {
  const float* dL_dcolor;      /* [3,H,W] */
  const float* dL_ddepth;      /* [1,H,W] or NULL (= zeros) */
} SgrGradOutputs;

/* Every pointer may be NULL when the caller does not need that gradient. Written, not accumulated. */
typedef struct SgrGradInputs {
  float* dL_dmeans3D;          /* [N,3] */
  float* dL_dmeans2D;          /* [N,3]  (x,y in NDC-scaled pixel units, z = 0) */
  float* dL_dopacities;        /* [N]   */
  float* dL_dshs;              /* [N,M,3] */
  float* dL_dcolors_precomp;   /* [N,3] */
  float* dL_dscales;           /* [N,3] */
  float* dL_drotations;        /* [N,4] */
  float* dL_dcov3D_precomp;    /* [N,6] */
  float* dL_dtau;              /* [6] = (rho[3], theta[3]) summed over Gaussians */
  /* --- fused mapping-loop extensions (all optional; zero / NULL = plain backward) ------------------------------
   * accumulate != 0: the per-Gaussian gradients above are ADDED to the buffers (only Gaussians with radii > 0 are
   * touched), so the <= 12 views of one mapping iteration (src/mapper.py:426-490) sum without autograd.
   * stat_*: densification statistics of add_densification_stats + the max_radii2D update
   * (gaussian_model.py:738-742, src/mapper.py:522-529), fused into the same pass:
   *   stat_grad_accum[i] += |dL_dmeans2D[i, :2]|, stat_denom[i] += 1, stat_max_radii[i] = max(., radii[i])
   *   for every Gaussian with radii > 0. */
  int32_t accumulate;
  float* stat_grad_accum;      /* [N] */
  float* stat_denom;           /* [N] */
  float* stat_max_radii;       /* [N] */
} SgrGradInputs;

int sgr_abi_version(void);
const char* sgr_last_error(void);

size_t sgr_saved_bytes(int32_t num_gaussians, int32_t image_height, int32_t image_width, int64_t capacity);
size_t sgr_scratch_bytes(int32_t num_gaussians, int32_t image_height, int32_t image_width, int64_t capacity);

/* num_rendered_host: host pointer or NULL.
 *   non-NULL: the call synchronises once on `stream` to learn the pair count R, sorts exactly R pairs, stores R
 *             there, and fails with SGR_ERR_CAPACITY (outputs untouched) when R > ws->capacity;
 *   NULL    : fully asynchronous; pairs beyond ws->capacity are dropped and the overflow word of the saved
 *             block is raised -- poll it with sgr_query().
 * Overflow word: 0 = fine; 1 = R > capacity (grow the workspace and redo); 2 = more than 65280 splats fell on ONE 8x8 tile
 * (the per-tile pair counters are 16-bit fields of a word shared by a 2x2 block of tiles; only a degenerate map gets there)
 * -- the synchronous form fails with SGR_ERR_INVALID.  A view whose overflow word is non-zero contributes zeros to
 * sgr_backward / sgr_map_* gradients instead of sums over truncated lists. */
int sgr_forward(const SgrSettings* settings, const SgrInputs* in, const SgrOutputs* out,
                const SgrWorkspace* ws, int64_t* num_rendered_host, void* stream);

int sgr_backward(const SgrSettings* settings, const SgrInputs* in, const int32_t* radii,
                 const SgrGradOutputs* grad_out, const SgrGradInputs* grad_in,
                 const SgrWorkspace* ws, void* stream);

/* The backward of SEVERAL forwards that share the Gaussian inputs, in one host call and -- when the views share N, H, W, the
 * view-independent settings and the capacity, and have private scratch blocks -- one launch per stage: what a mapping
 * iteration's loss.backward() asks of the rasterizer (src/mapper.py:426-490: up to 12 forwards, then ONE backward).  The
 * per-Gaussian gradients in `grad_in` are the SUMS over the views (written in full when grad_in->accumulate == 0, added
 * otherwise); every view additionally receives its own dL_dmeans2D ([N,3]; only rows of Gaussians with radii > 0 are written:
 * pass a zeroed buffer) and dL_dtau ([6]).  grad_in->dL_dmeans2D / dL_dtau are ignored.  The drop-in package's autograd
 * collector calls this once per backward pass. */
typedef struct SgrBackwardView {
  SgrSettings settings;
  const int32_t* radii;        /* [N] as written by the view's sgr_forward */
  SgrWorkspace ws;             /* the view's saved block + a scratch block of its own */
  const float* dL_dcolor;      /* [3,H,W] */
  const float* dL_ddepth;      /* [1,H,W] or NULL */
  float* dL_dmeans2D;          /* [N,3] zero-initialised by the caller, or NULL */
  float* dL_dtau;              /* [6] or NULL */
} SgrBackwardView;
int sgr_backward_views(int32_t num_views, const SgrBackwardView* views, const SgrInputs* in, const SgrGradInputs* grad_in,
                       void* stream);

/* add_densification_stats + the max_radii2D update of one view (gaussian_model.py:738-742, src/mapper.py:522-529) in one pass:
 * for every Gaussian with radii > 0: grad_accum += |dL_dmeans2D[i, :2]|, denom += 1, max_radii = max(max_radii, radii). */
int sgr_densify_stats(int64_t n, const float* dL_dmeans2D, const int32_t* radii, float* stat_grad_accum, float* stat_denom,
                      float* stat_max_radii, void* stream);

/* Synchronous read-back of (pair count, overflow flag) from a saved block produced by sgr_forward. */
int sgr_query(const void* saved, int64_t* num_rendered_host, int32_t* overflow_host, void* stream);

/* The whole 64-byte header of a saved block, synchronously: uint32 words [0] pair count R the workspace must hold (the larger of
 * pairs binned and partial slots reserved), [1] overflow flag, [2] pairs sorted, [3] visible Gaussians, [4] tiles with more
 * than 64 pairs, [5..8] internal, [9] pairs actually binned, [10] longest per-tile list, [11] internal, [12] STICKY: number of
 * forwards of this workspace whose overflow flag came out non-zero since the block's counters were last zeroed (counters_clean = 0),
 * [13] STICKY: the largest [0] of any of those forwards -- [1] only describes the LAST forward, and a span of sgr_map_run puts dozens
 * of forwards through one workspace between two host checks -- [14] internal (a flag between two blocks of the binning kernel: zero
 * whenever no forward is running), [15] zero. */
int sgr_query_header(const void* saved, uint32_t words_host[16], void* stream);

/* Asynchronous variant: enqueues a 64-byte copy of the same header into PINNED host memory on `stream`.
 * The caller pre-sets word 15 to a non-zero sentinel and knows the copy has landed when it reads 0 there: a later call can
 * then learn R without ever waiting (the drop-in package sizes its capacity this way). */
int sgr_header_to_host(const void* saved, void* pinned_host64, void* stream);

/* Work counters of one forward, read back synchronously (bench / roofline accounting only):
 *   stats[0] = V  Gaussians with radii > 0          stats[1] = R  (tile, Gaussian) pairs binned
 *   stats[2] = R_eff = sum over tiles of min(list length, last contributor): pairs the blend kernels walk
 *   stats[3] = number of non-empty 8x8 tiles */
int sgr_query_stats(const SgrWorkspace* ws, int32_t num_gaussians, int32_t image_height, int32_t image_width,
                    const int32_t* radii, int64_t stats_host[4], void* stream);
/* The fp32 view-space depth of every Gaussian with radii > 0 as the forward computed it (0 elsewhere) -- the bit pattern
 * that orders the splats of a tile.  Parity tooling: lets a comparison break depth near-ties (equal to the last ulp) the way
 * this forward did.  depth_out: device [N]. */
int sgr_query_depth_keys(const SgrWorkspace* ws, int32_t num_gaussians, int32_t image_height, int32_t image_width,
                         const int32_t* radii, float* depth_out, void* stream);
/* Histogram of the per-tile list lengths the blend kernels walk (same synchronous, accounting-only use):
 * bins 0, 1-4, 5-8, 9-16, 17-32, 33-64, 65-256, >256 splats. */
int sgr_query_list_histogram(const SgrWorkspace* ws, int32_t num_gaussians, int32_t image_height, int32_t image_width,
                             int64_t hist_host[8], void* stream);

/* Run-time options of the library (process-wide; no reference counterpart).
 *   SGR_OPT_FUSED_BLEND (default 1): sgr_map_views / sgr_map_step / sgr_map_run composite a tile, evaluate the mapping loss
 *     and run the tile's backward in ONE kernel (the same wave, pixel state in registers).  0: the two halves run as the
 *     separate kernels sgr_forward / sgr_backward use (bitwise identical results) -- for timing the halves on their own.
 *   SGR_OPT_UPSTREAM_POSE_JACOBIAN (default 0): how the projected-mean path enters the camera-pose gradient dL/dtau.
 *     0: the exact derivative of the projection (x_ndc = (P00 X + P02 Z) / Z ...), what autograd through the reference's
 *        own SE3_exp / update_pose convention gives (thirdparty/monogs/utils/pose_utils.py:66-98);
 *     1: the form the pinned CUDA rasterizer is believed to use: five scalars of projmatrix_raw
 *        (P00, P11, P22, P23, P32), d x_ndc / d p_cam = (P00 / w, 0, -x_hom / w^2) -- i.e. WITHOUT the principal-point terms
 *        P02 / w, P12 / w.  Identical when cx = W/2 and cy = H/2; differs by O(|P02|) otherwise (8e-4 on Replica).  Only
 *        dL/dtau changes; every other gradient is the same.  The oracle has the same switch (UPSTREAM_POSE_JACOBIAN).
 *   SGR_OPT_SEGMENT_TEST (default 0): the forward tests every 256-Gaussian segment's bounding box against each view before
 *     testing its Gaussians one by one (a map that grows keyframe by keyframe is spatially coherent: whole segments miss
 *     whole views).  Conservative: results are identical either way.  Off by default because it does not pay on MI355X:
 *     preprocess_fwd is bound by its counting atomics and output writes, not by the per-Gaussian visibility arithmetic
 *     (measured on a keyframe-ordered 300 k map: 65.6 us with, 64.7 us without). */
#define SGR_OPT_FUSED_BLEND 0
#define SGR_OPT_UPSTREAM_POSE_JACOBIAN 1
#define SGR_OPT_SEGMENT_TEST 2
#define SGR_OPT_COUNT 3
int sgr_set_option(int32_t option, int32_t value);
int sgr_get_option(int32_t option);

/* Per-kernel HIP-event timing.  kind: 0 preprocess_fwd (+ binning), 1 tile_scan, 2 scatter, 3 fused tile kernel (blend
 * forward + loss + blend backward), 4 blend_fwd (+ in-wave tile sort), 5 blend_bwd, 6 preprocess_bwd (+ gather / optimiser
 * pass, pose reduce).  sgr_profile_enable(mask) arms event pairs around the kinds whose
 * bit is set (0 disarms); sgr_profile_read() synchronises, returns accumulated milliseconds and launch counts per
 * kind since the last read, and resets them. Events are recorded on the stream the kernel is launched on. */
#define SGR_PROFILE_KINDS 7
int sgr_profile_enable(uint32_t kind_mask);
int sgr_profile_read(float ms_host[SGR_PROFILE_KINDS], int64_t launches_host[SGR_PROFILE_KINDS]);

/* Fused mapping loss (slam_utils.py:71-105): loss = alpha*mean|m*(e^a*I+b) - m*gt| + (1-alpha)*mean|md*D - md*gtD|
 * with m = (sum_c gt > rgb_boundary_threshold), md = (gtD > 0.01).  Writes loss[1] and the four gradients
 * scaled by `upstream` (dLoss/dloss).  exposure may be NULL (initialization=True branch, :72-73). */
int sgr_mapping_loss(int32_t H, int32_t W, const float* image, const float* depth,
                     const float* gt_image, const float* gt_depth,
                     const float* exposure_a, const float* exposure_b,
                     float alpha, float rgb_boundary_threshold, float upstream,
                     float* loss, float* dL_dimage, float* dL_ddepth, float* dL_dexp_a, float* dL_dexp_b,
                     void* scratch, size_t scratch_bytes, void* stream);

/* One torch.optim.Adam step (no weight decay, no amsgrad) on a flat parameter slab. step = the value AFTER
 * increment (1 on the first call).  lr may differ per call (update_learning_rate, gaussian_model.py:315-329). */
int sgr_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  float lr, float beta1, float beta2, float eps, int64_t step, void* stream);

/* The same step for MANY SMALL tensors that share lr / betas / eps, in one launch per 48 tensors (block = tensor): the keyframe
 * optimiser of src/mapper.py:1096-1111 holds two one-element exposure parameters (and two 3-vectors of pose deltas) per window
 * keyframe.  Each tensor carries its own step count (AFTER increment). */
typedef struct SgrAdamTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;                   /* elements (<= 2^20) */
  int64_t step;
} SgrAdamTensor;
int sgr_adam_step_multi(int32_t count, const SgrAdamTensor* tensors, float lr, float beta1, float beta2, float eps, void* stream);

/* Activations of the GaussianModel getters (gaussian_model.py:76-101) in one pass:
 * scales_out = exp(scaling), rot_out = rotation / max(|rotation|, 1e-12), opac_out = sigmoid(opacity). */
int sgr_activate(int64_t n, const float* scaling, const float* rotation, const float* opacity,
                 float* scales_out, float* rot_out, float* opac_out, void* stream);

/* One fused optimiser step of the mapping loop for all Gaussian parameter groups (src/mapper.py:487-489,557):
 * takes the gradients wrt the ACTIVATED rasterizer inputs (as accumulated by sgr_backward), applies the chain rule
 * through exp / sigmoid / normalize, adds the gradient of the isotropy regulariser
 * iso_weight * mean|s - mean(s)| (0 disables it: initialize_map / final_refine), then runs torch.optim.Adam's update
 * (eps as given, no weight decay) in place on the raw parameters and their exp_avg / exp_avg_sq, and zeroes the
 * gradient accumulators for the next iteration.  lr order: xyz, f_dc, opacity, scaling, rotation.
 * f_dc is [N,3] (sh_degree 0 layout [N,1,3]). */
typedef struct SgrAdamGroup {
  float* param;
  float* grad;                 /* accumulator wrt the activated input; zeroed on return */
  float* exp_avg;
  float* exp_avg_sq;
  float lr;
  int32_t skip;                /* != 0: leave param/moments untouched, only zero the accumulator (a group whose
                                  tensor was just replaced has grad None in the reference: Adam skips it) */
  int64_t step;                /* this group's Adam step count AFTER increment (>= 1 unless skip) */
} SgrAdamGroup;
int sgr_gaussian_adam_step(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps,
                           float iso_weight, void* stream);

/* The same step for a SLICE of the optimiser (multi-GPU ZeRO-1, SURVEY.md 8e: gradients reduce-scattered over the ranks,
 * every rank steps the rows it owns, parameters all-gathered afterwards): group k is stepped for the Gaussians
 * row0[k] <= i < row1[k] only; its four pointers still address ROW 0 of the (virtual) full arrays and are only
 * dereferenced inside that range, so `grad` may point into a rank-local shard buffer.  The isotropy term is normalised by
 * n_total (the whole map).  No activations are written: the caller re-activates after its all-gather. */
int sgr_gaussian_adam_shard(int64_t n_total, const SgrAdamGroup groups[5], const int64_t row0[5], const int64_t row1[5],
                            float beta1, float beta2, float eps, float iso_weight, void* stream);

/* One mapping-loop view: render, mapping loss and its gradient, backward (src/mapper.py:426-456 for one viewpoint).
 * sgr_map_views runs the sequence  sgr_forward(async) -> sgr_mapping_loss -> sgr_backward  for `num_views` views
 * that share the Gaussian inputs `in` and the gradient sinks `grads` (use accumulate = 1) with ONE host call, so the
 * host cost of a mapping iteration is the kernel launches only. */
typedef struct SgrMapView {
  SgrSettings settings;
  SgrOutputs out;              /* color / depth / opacity may ALL be NULL when only loss + gradients are wanted;
                                  n_touched may be NULL (not counted) */
  SgrWorkspace ws;
  const float* gt_image;       /* [3,H,W] */
  const float* gt_depth;       /* [H,W]   */
  const float* exposure_a;     /* [1] or NULL */
  const float* exposure_b;     /* [1] or NULL */
  float* loss;                 /* [1] */
  float* dL_dimage;            /* [3,H,W] scratch for this view.  Uniform batches (the normal case) pass the pixel
                                  gradients of the L1 loss from the compositing epilogue to the backward as ONE code
                                  byte per pixel in its first H*W bytes (2 bits per value: 0, +, -); only the
                                  one-view-at-a-time path of heterogeneous batches leaves float gradients here */
  float* dL_ddepth;            /* [1,H,W] scratch (same remark) */
  float* dL_dexposure;         /* [2] = (d/da, d/db) or NULL */
  float* dL_dtau;              /* [6] or NULL */
  void* loss_scratch;
  size_t loss_scratch_bytes;
} SgrMapView;
int sgr_map_views(int32_t num_views, const SgrMapView* views, const SgrInputs* in, const SgrGradInputs* grads,
                  float alpha, float rgb_boundary_threshold, int32_t forward_only, void* stream);

/* One whole mapping iteration (src/mapper.py:414-568 without densification) in ONE host call:
 *   sgr_activate -> sgr_map_views -> sgr_gaussian_adam_step -> sgr_masked_adam (exposures).
 * Any stage is skipped when its pointer block is NULL / its count is 0.
 * An optimiser-only step (num_views == 0 with adam_groups: the second half of a multi-GPU iteration, after the gradient
 * all-reduce) does not activate first: its Adam pass writes scales_out / rot_out / opac_out of the UPDATED parameters
 * (for the non-NULL ones of scaling / rotation / opacity), so the next views step can skip sgr_activate. */
typedef struct SgrMapStep {
  int64_t num_gaussians;
  const float* scaling;        /* raw parameters for sgr_activate (NULL = skip activation) */
  const float* rotation;
  const float* opacity;
  float* scales_out;
  float* rot_out;
  float* opac_out;
  int32_t num_views;
  int32_t forward_only;
  const SgrMapView* views;
  const SgrInputs* in;
  const SgrGradInputs* grads;
  float alpha;
  float rgb_boundary_threshold;
  const SgrAdamGroup* adam_groups;   /* [5] or NULL (no optimiser step this iteration) */
  float beta1, beta2, eps, iso_weight;
  int32_t exp_rows;            /* rows of the exposure slab to consider (0 = skip) */
  int32_t exp_row_width;
  float* exp_param;
  const float* exp_grad;
  float* exp_avg;
  float* exp_avg_sq;
  int32_t* exp_step;
  const int32_t* exp_active;
  float exp_lr, exp_beta1, exp_beta2, exp_eps;
  int32_t grads_clean;         /* > 0: the gradient sinks are known to be all-zero on entry (as every Adam step leaves
                                  them); the fused gather+Adam pass then never touches them.  0: unknown.
                                  -1: keep gather and Adam as separate passes (verification)
                                  -2 (with adam_groups == NULL): no optimiser step, but the gather pass of the fused form
                                  ADDS the views' gradient sums to the sinks and carries the loss sums and the exposure
                                  step -- the first half of a multi-GPU iteration (an all-reduce of the sinks and an
                                  optimiser-only step follow)
                                  -3: like -2, but the pass STORES the sums (zeros for Gaussians no view of the batch sees)
                                  into the sinks of all num_gaussians rows instead of adding: the sinks need not be zeroed
                                  between two iterations (one 56 B x N memset per exchange less).  A batch that cannot take the
                                  fused gather pass (a view asks for dL_dtau, heterogeneous views, > 16 views) has its sinks
                                  zeroed by the library first and is then accumulated: the sinks hold this call's sums on
                                  every path */
} SgrMapStep;
int sgr_map_step(const SgrMapStep* step, void* stream);

/* A run of `num_iters` REGULAR mapping iterations (no densification / opacity reset between them) enqueued by one host
 * call: what `for _ in range(iters)` of Mapper.map (src/mapper.py:414-568) or Mapper.final_refine (:656-708) does
 * between two map-surgery points.  Iteration k renders window[0..num_window) plus pool[picks[k*picks_per_iter + j]]
 * (the reference's random keyframes, drawn by the caller so that its RNG stream is unchanged), steps Adam with
 * adam_groups[0].lr = lr0[k] (the xyz schedule, src/mapper.py:564) and bumps every non-skipped group's step counter;
 * `adam_groups` is updated in place so that the caller can read the counters back.
 * pool_exp_row (optional): exposure-slab row of each pool entry (-1 = none); when given, iteration k steps only the
 * row of its first pick (final_refine: torch's Adam skips parameters without a gradient) -- step.exp_* then point at
 * row 0 of the slab and step.exp_active at an all-ones array. */
typedef struct SgrMapRun {
  SgrMapStep step;              /* template of one iteration; its views / num_views / adam_groups are ignored */
  int32_t num_iters;
  int32_t num_window;
  const SgrMapView* window;
  int32_t pool_size;
  int32_t picks_per_iter;
  const SgrMapView* pool;
  const int32_t* picks;         /* host, [num_iters * picks_per_iter] */
  const float* lr0;             /* host, [num_iters] or NULL (keep adam_groups[0].lr) */
  SgrAdamGroup* adam_groups;    /* host, [5] or NULL */
  const int32_t* pool_exp_row;  /* host, [pool_size] or NULL */
  int32_t n_touched_last_only;  /* != 0: n_touched is only produced by the last iteration (nobody can observe the others) */
  const SgrWorkspace* pick_ws;  /* host, [picks_per_iter] or NULL.  Non-NULL: pick j of every iteration renders in pick_ws[j]
                                   instead of its pool entry's own workspace -- a map holds hundreds of keyframes, an iteration
                                   touches picks_per_iter of them (src/mapper.py:458-485), and a workspace is ~300 MB.  The
                                   pool entries' `ws` are then ignored. */
} SgrMapRun;
int sgr_map_run(const SgrMapRun* run, void* stream);

/* Adam on a small slab with a per-row switch: row r (width `row_width`) is updated iff active[r] != 0, using its own
 * step counter step[r] (incremented in place).  The exposure parameters of the keyframe optimiser
 * (src/mapper.py:1096-1111: lr 0.01, default eps 1e-8) live in such a slab. */
int sgr_masked_adam(int32_t rows, int32_t row_width, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    int32_t* step, const int32_t* active, float lr, float beta1, float beta2, float eps, void* stream);

/* Mapper.update_mapping_points (src/mapper.py:154-255) as ONE pass: the Gaussians anchored to keyframe `frame_idx`
 * (unique_kfIDs == frame_idx) are depth-rescaled along the old camera's ray (unless rigid), moved by `transform`
 * (= inv(inv(w2c_old) @ w2c_new), host, row-major) and rotated by its quaternion; every rotation leaves normalised (the
 * reference writes the activated rotations back).  In place on the raw parameter tensors; resetting the Adam moments of
 * the three tensors (replace_tensor_to_optimizer, gaussian_model.py:488-501) stays with the caller. */
typedef struct SgrDeformFrame {
  int32_t frame_idx;
  int32_t rigid;               /* != 0: pose change only (no depth rescale, depth maps unused) */
  float w2c_old[16];           /* host, row-major 4x4 */
  float c2w_old[16];           /* inverse of w2c_old */
  float transform[16];
  float quat_wxyz[4];          /* rotation part of `transform` */
  float intrinsics[9];         /* row-major K */
  int32_t height, width;
  const float* depth_new;      /* device [H,W] */
  const float* depth_old;      /* device [H,W] */
} SgrDeformFrame;
int sgr_deform_points(int64_t n, const int32_t* unique_kfIDs, const SgrDeformFrame* frame, float* xyz, float* rotation,
                      float* scaling, void* stream);

/* Densify / prune compaction (gaussian_model.py:519-557 prune_points, _prune_optimizer; :690-719 the row selects of
 * densify_and_clone / densify_and_split): sgr_keep_list turns a byte mask into the ascending list of kept row indices
 * (count written to a device int64, no host sync); sgr_gather_rows copies rows src_rows[k] -> k for up to any number of
 * tensors (parameters, Adam moments, statistics, keyframe ids ...) in one launch per 32 tensors. */
typedef struct SgrRowTensor {
  const void* in;              /* [n_in, row_bytes] */
  void* out;                   /* [m, row_bytes], must not alias `in` */
  int32_t row_bytes;           /* multiple of 4 */
} SgrRowTensor;
size_t sgr_compact_scratch_bytes(int64_t n);
int sgr_keep_list(int64_t n, const uint8_t* keep, int32_t* src_rows, int64_t* count_device, void* scratch, size_t scratch_bytes,
                  void* stream);
int sgr_gather_rows(int64_t m, const int32_t* src_rows, int32_t num_tensors, const SgrRowTensor* tensors, void* stream);

/* simple_knn distCUDA2: mean squared distance to the 3 nearest neighbours (self excluded). */
size_t sknn_scratch_bytes(int32_t n);
int sknn_dist2(const float* xyz, int32_t n, float* mean_dist2, void* scratch, size_t scratch_bytes, void* stream);

/* SE3 ops, batched over n.  Pose = (tx,ty,tz,qx,qy,qz,qw) as in lietorch / depth_video.py:69; tau = (rho, theta). */
int se3_exp(const float* tau, int64_t n, float* pose_out, void* stream);
int se3_log(const float* pose, int64_t n, float* tau_out, void* stream);
int se3_inv(const float* pose, int64_t n, float* pose_out, void* stream);
int se3_mul(const float* pose_a, const float* pose_b, int64_t n, float* pose_out, void* stream);
int se3_act(const float* pose, const float* pts, int64_t n, float* pts_out, void* stream);
int se3_adjT(const float* pose, const float* a, int64_t n, float* out, void* stream);
int se3_matrix(const float* pose, int64_t n, float* mat_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPLAT_HIP_H_ */
