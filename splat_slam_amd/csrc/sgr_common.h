// Internal layout + device helpers shared by the gfx950 rasterizer kernels.
// Design notes live in DESIGN.md; the public surface is include/splat_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/splat_hip.h"

namespace sgr {

// ---- constants of the rasterization algorithm (3DGS / MonoGS w-pose fork, see oracle/raster_oracle.py)
constexpr float kNearPlane = 0.001f;   // /root/reference/README.md:88-92 (patched from 0.2)
constexpr float kAlphaMin = 1.0f / 255.0f;
constexpr float kAlphaMax = 0.99f;
constexpr float kTEps = 1e-4f;
constexpr float kDilation = 0.3f;
constexpr float kTouchedT = 0.5f;
constexpr int kRefTile = 16;           // upstream tile edge: defines which pixels a splat may reach
constexpr int kTile = 8;               // our binning tile = one wave64 = 8x8 pixels
constexpr int kWave = 64;
// Per-tile pair counters.  Binning is bound by the NUMBER of counting atomics (they execute at the memory side: ~12 per ns
// chip-wide, measured: 245 of the 357 us of K1 on a converged map), so ONE 64-bit word holds the four 16-bit counters of a
// 2x2 block of tiles and a splat takes the ranks of every tile it covers in that block with one atomic.  A field that
// reaches kTileCountLimit marks the view as degenerate (header.overflow = 2): beyond it a carry could reach the
// neighbour's field.  One word per 32 bytes keeps the per-line contention low (one counter per line measured no better).
constexpr int kCntSlotWords = 8;       // uint32 units per 2x2-block slot
constexpr uint32_t kTileCountLimit = 0xff00u;
__host__ __device__ inline size_t tile_counter_word(int x, int y, int gxp) { return ((size_t)(y >> 1) * gxp + (x >> 1)) * (kCntSlotWords / 2); }   // in uint64 units
__host__ __device__ inline int tile_counter_shift(int x, int y) { return 16 * ((y & 1) * 2 + (x & 1)); }
// K1's block = one 256-Gaussian segment x a SUBSET of the batch's views (view v belongs to part v % parts).  One part when the
// map has enough segments to fill the chip; a young SLAM map (60 k Gaussians = 235 segments, 40 % of them visible, splats of
// hundreds of bins) would otherwise run its whole forward binning on a quarter of the CUs.
constexpr int kK1MaxParts = 4;
// A DENSE map (the caller measured lists beyond a bucket) takes two parts even when it has segments enough: the second round of
// blocks overlaps its loads and projections with the first round's counting atomics, which is where such a map's K1 waits
// (opaque bench scene 0.21 -> 0.185 ms; a fresh map loses 8 us to the doubled cull and keeps one part, and so does a map of
// more than half a million Gaussians, where the cull itself is the larger half of K1: 1.5 M Gaussians, 0.913 ms per iteration
// with one part, 0.997 with two).
#ifndef SGR_K1_DENSE_PARTS
#define SGR_K1_DENSE_PARTS 2
#endif
#ifndef SGR_K1_LARGE_PARTS
#define SGR_K1_LARGE_PARTS 1
#endif
__host__ __device__ inline int k1_parts_for(int nseg, int longest_list_hint = 0) {
  return nseg >= 768 ? ((longest_list_hint > 64 && nseg < 2048) ? SGR_K1_DENSE_PARTS : SGR_K1_LARGE_PARTS) : (nseg >= 384 ? 2 : kK1MaxParts);
}
constexpr int kRngStride = 2;          // uint2 units between two tiles' (start, cursor) ranges
// Binning fast path: the counting atomic of K1 RETURNS the pair's rank inside its tile, and while the tile has room the
// key goes straight into the tile's fixed bucket -- no second pass.  Only tiles with more than kBucket pairs are re-scattered
// by scatter_kernel into the exactly sized runs (ranges[t].x carries kOverfull for them).
// Round 6: 256 entries (rounds 2-5: 64).  On a converged map (lists of 65-256) a third to a half of all pairs used to leave K1 through
// the overflow list -- a 16-byte entry appended behind a per-view cursor inside a wave-serial loop, read and filed again by K3 --;
// with buckets that hold such lists whole K1 takes 0.186 -> 0.153 ms on the opaque bench scene, K3 is not launched at all (0.018 ms),
// the iteration goes from 0.88 to 0.835 ms and a 40-frame session from 51-55 to 48-49 ms per keyframe (same-box A/B of three builds,
// scripts/micro/r06_variant_ab.sh on builds with -DSGR_BUCKET=64 / 128 / 256; bitwise the same maps).  Costs address space only: 2 KB per tile and view that
// nobody touches beyond the tile's count (9.8 MB per 640x480 view).  A fresh map (lists <= 32) is unaffected.
#ifndef SGR_BUCKET
#define SGR_BUCKET 256
#endif
constexpr int kBucket = SGR_BUCKET;
constexpr uint32_t kOverfull = 0x80000000u;
constexpr int kSeg = 256;              // Gaussians per preprocess segment (one 256-thread block), see preprocess_fwd_kernel
constexpr int kSegShift = 8;

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

// ---- header at the start of the saved block
// ---- exact footprint test of a splat against ONE 8x8 bin -----------------------------------------------------------------
// A pixel can only pass alpha >= 1/255 where q = A dx^2 + 2 B dx dy + C dy^2 <= tau = 2 ln(255 opacity): the bin is unreachable
// if the minimum of q over the bin's 8x8 pixel centres' bounding box exceeds tau.  K1 drops such bins from the rectangle's
// operations (no atomic, no key, no compositing work: the axis-aligned box of the level set keeps ~15-25 % bins too many for a
// rotated or simply round splat of a converged map); the backward skips the same bins' partial slots, which nobody wrote.
// The two sides need NOT decide bit-identically: K1 drops a bin only beyond kFootDrop * thr + kFootGuard, the backward skips
// already beyond thr, and everything in between was binned, composited with alpha < 1/255 everywhere, i.e. wrote exact zeros.
// kFootGuard is ABSOLUTE and twice the largest rounding error of q the test is enabled for (err_limit <= 0.006 below): the two
// evaluations -- inlined into different kernels -- may round differently (and thr is ~0.02 for a faint splat, so a relative
// band alone would be 4e-5), but never by more than the band; footprint_qmin() is additionally compiled without contraction.
struct Footprint { float px, py, A, B, C, invA, invC, thr; };      // thr < 0: test disabled (rounding could decide), keep every bin
constexpr float kFootDrop = 1.002f;
constexpr float kFootGuard = 0.012f;
// the test pays for itself on rectangles of >= 6 bins with both sides >= 2 (a one-bin-wide strip has no bin to drop: the level
// set touches both ends of its box); a fresh map's splats (3 bins on average) skip it
__host__ __device__ inline bool footprint_worthwhile(int w, int h) { return w >= 2 && h >= 2 && w * h >= 6; }
__device__ __forceinline__ Footprint make_footprint(float px, float py, float A, float B, float C, float opac, float err_limit) {
  Footprint f{px, py, A, B, C, 0.f, 0.f, -1.f};
  const float detc = A * C - B * B, o255 = 255.f * opac;
  if (o255 >= 1.f && A > 0.f && C > 0.f && detc > 0.f) {
    const float tau = 2.f * __logf(o255) * 1.001f + 0.02f;
    const float ex = __builtin_amdgcn_sqrtf(tau * C / detc), ey = __builtin_amdgcn_sqrtf(tau * A / detc);
    const float ext = fmaxf(ex, ey) + 2.f * kTile;
    const float err = 4e-7f * (A + C + 2.f * fabsf(B)) * ext * ext;      // fp32 rounding of q near the boundary
    if (err < err_limit && isfinite(ex) && isfinite(ey)) { f.thr = tau; f.invA = 1.f / A; f.invC = 1.f / C; }
  }
  return f;
}
// minimum of q over the box [xl, xh] x [yl, yh] of offsets from the centre (0 when the centre is inside)
__device__ __forceinline__ float footprint_qmin(const Footprint& f, int tx, int ty) {
#pragma clang fp contract(off)
  const float xl = (float)(tx * kTile) - f.px, xh = xl + (float)(kTile - 1);
  const float yl = (float)(ty * kTile) - f.py, yh = yl + (float)(kTile - 1);
  if (xl <= 0.f && xh >= 0.f && yl <= 0.f && yh >= 0.f) return 0.f;
  auto edge_x = [&](float c) {      // dx = c, dy free in [yl, yh]
#pragma clang fp contract(off)
    const float d = fminf(yh, fmaxf(yl, -f.B * c * f.invC));
    return f.A * c * c + 2.f * f.B * c * d + f.C * d * d;
  };
  auto edge_y = [&](float c) {      // dy = c, dx free in [xl, xh]
#pragma clang fp contract(off)
    const float d = fminf(xh, fmaxf(xl, -f.B * c * f.invA));
    return f.A * d * d + 2.f * f.B * d * c + f.C * c * c;
  };
  return fminf(fminf(edge_x(xl), edge_x(xh)), fminf(edge_y(yl), edge_y(yh)));
}

struct SavedHeader {
  uint32_t num_rendered;   // R: total (tile, Gaussian) pairs demanded (may exceed capacity)
  uint32_t overflow;       // 1: R > capacity (pairs were dropped); 2: a 16-bit tile counter saturated (> kTileCountLimit splats on one tile)
  uint32_t sorted_count;   // number of pairs actually binned = min(R, capacity)
  uint32_t num_visible;    // V: Gaussians with radii > 0
  uint32_t num_overfull;   // tiles with more than kBucket pairs (scatter_kernel completes only those)
  uint32_t ovf_cursor;     // K1's append cursor into the overflow list (pairs whose rank in their tile is >= kBucket); K2 resets it
  uint32_t ovf_count;      // ... its final value for this forward (written by K2, read by K3)
  uint32_t count_saturated;  // K1: some pair drew a rank >= kTileCountLimit (sticky until K2 folds it into `overflow` and clears it)
  uint32_t slot_total;     // partial slots the view's Gaussians reserve (sum of their bin-rectangle areas, >= pairs binned: bins of the
                           // rectangle that the exact footprint test dropped keep their slot); K2 writes it, K3 folds it into num_rendered
  uint32_t num_binned;     // pairs actually binned (K3; num_rendered then holds the capacity-relevant max(pairs binned, slot_total))
  uint32_t max_tile_count; // longest per-tile list of this forward (K2): what the caller picks the tile kernels' sort build by
  uint32_t k2_tickets;     // K2's blocks 0 and 1 take a ticket when done; the later one folds the header and resets this
  // STICKY across forwards (only zero_heads clears them, together with the tile counters of a fresh block): `overflow` describes the
  // LAST forward only, and a workspace runs up to ~52 forwards between two host checks (sgr_map_run; a slot renders a different
  // camera every iteration) -- a truncated forward in the middle of a span would be overwritten before anybody looks.  The host
  // remembers the count it saw at its previous check: a changed count = some forward since then was truncated.
  uint32_t overflow_events;   // forwards of this workspace whose `overflow` came out non-zero
  uint32_t max_rendered;      // largest num_rendered of any forward of this workspace (what a replay sizes the capacity by)
  uint32_t counts_read;    // K2: block 0 (tile starts) tells block 3 (launch order), which zeroes the tile counters, that it has read them
  uint32_t pad;
};
static_assert(sizeof(SavedHeader) == 64, "the header is one 64-byte record (sgr_query_header copies 16 words)");

inline __host__ __device__ size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- batched launches: every kernel takes the per-view pointers of up to kMaxViews views BY VALUE (kernarg) and picks
// its view with blockIdx.y, so the <= 12 views of a mapping iteration (src/mapper.py:426-485) are ONE launch per stage.
constexpr int kMaxViews = 16;
struct ViewTab {
  const float* viewmatrix[kMaxViews];
  const float* projmatrix[kMaxViews];
  const float* campos[kMaxViews];
  const float* projraw[kMaxViews];   // projection_matrix of the view (per view since round 4: cameras of one batch may hold their own copies,
                                     // and intrinsics that differ only in the principal point batch together)
  char* saved[kMaxViews];
  char* scratch[kMaxViews];
  float* color[kMaxViews];
  float* depth[kMaxViews];
  float* opacity[kMaxViews];
  int32_t* radii[kMaxViews];
  int32_t* n_touched[kMaxViews];
  const float* dL_dcolor[kMaxViews];
  const float* dL_ddepth[kMaxViews];
  float* dL_dtau[kMaxViews];
  float* dL_dmeans2D[kMaxViews];   // per-view [N,3] screen-space mean gradients (rows of visible Gaussians are written) or NULL
};
// device-visible copy of the Layout offsets (identical for all views of a batch: same N, H, W, capacity)
struct LOff {
  int N, H, W, gx, gy, gxp, sgx, sgy, ntiles, pre_blocks, nseg, dbg, mean_hint, k1_parts;
  uint32_t sgx_magic;      // floor(2^32 / sgx) + 1: q = mulhi(st, sgx_magic) = st / sgx exactly for st * sgx < 2^32 (a scalar multiply instead of a VALU division)
  // the caller's measured longest list (0: unknown): everything derived from it changes with it, in ONE place
  __host__ void set_hint(int h) { mean_hint = h; k1_parts = k1_parts_for(nseg, h); }
  int64_t cap;
  size_t o_hdr, o_tile_count, o_grec, o_point_list, o_ranges,
      o_tile_maxc, o_final_T, o_tile_order, o_block_touched, o_block_vis, o_block_base_t, o_block_base_v, o_vis_list,
      o_seg_list, o_vismask, o_entries, o_bucket, o_ovf, o_partials, o_tau_part, o_gradrec, o_taurec;
};
// Longest-first launch order of the super tiles (K2 block 3 -> tile_of_block): worth its microsecond of K2 and the extra dependent
// load in every tile wave's prologue where lists are long enough for a late heavy tile to hold the launch up (opaque scene: tile
// kernel -5 %; a map whose measured lists stay within 64 splats: +1 us of K2 and +0.7 us of tile kernel for nothing).  A function of
// the CALL (the caller's measured longest list; unknown = ordered), like every other build choice.
#ifndef SGR_TILE_ORDER_MIN_LIST
#define SGR_TILE_ORDER_MIN_LIST 64
#endif
__host__ __device__ inline bool tile_order_used(const LOff& L) { return !(L.mean_hint != 0 && L.mean_hint <= SGR_TILE_ORDER_MIN_LIST); }
// one pair that did not fit its tile's bucket: K1 knows its tile and its rank inside the tile (the counting atomic returned it)
// but not yet where the tile's run starts -- K3 files it at start(tile) + rank once K2 has scanned the counts
struct __attribute__((aligned(16))) OvfEntry { uint32_t tile, rank; uint64_t key; };
// shared (view independent) scalars of a batch
struct Common {
  int deg, M;
  int upstream_pose_jac;   // SGR_OPT_UPSTREAM_POSE_JACOBIAN
  float tanfovx, tanfovy, mod;
  const float* bg;
};

// SGR_DEBUG environment variable -- stage-cost experiments only (scripts/stage_times.py), results are wrong when set:
//   bit 0: preprocess_fwd stops after the cull / compact phase     bit 1: ... and skips the visibility test's arithmetic
//   bit 2: preprocess_fwd does NOT use the whole-segment test (results stay correct: A/B timing of that test)
//   bit 3: the cull uses the old per-Gaussian projection test instead of the plane test (results stay correct)
//   bit 4: the cull does not zero radii / n_touched     bit 5: no counting atomics (every rank 0)
//   bit 6: counting atomics add 0 and nobody consumes the ranks (atomics issued, lists stay empty)     bit 7: no record writes
//   bit 11: the fused tile kernel skips the forward walk and the backward     bit 12: ... the backward only
//           (scripts/micro/pmc_tile.py under these bits: which part of a tile's wave the instructions belong to)
int debug_flags();

// Everything later stages gather BY GAUSSIAN for one view, as ONE 64-byte record (= one HBM sector pair, one L2 line
// half) instead of seven SoA arrays: a gather of 8...16 bytes costs a whole sector, so seven arrays meant up to seven
// sectors per (tile, Gaussian) pair.
struct __attribute__((aligned(16))) GRec {
  float px, py;                 // projected centre (pixels)
  uint32_t rect01, rect23;      // 8x8-tile rectangle x0 | y0<<16, x1 | y1<<16
  float A, B, C, opacity;       // conic + activated opacity
  float r, g, b, depth;         // colour, view-space depth
  uint32_t touched, offset, vis_pos, clamped;   // pairs, in-segment prefix of pairs, slot in the visible list, SH clamp bits
};
static_assert(sizeof(GRec) == 64, "GRec must be one 64-byte record");
struct Rect { int x0, y0, x1, y1; };
__device__ __forceinline__ Rect unpack_rect(uint32_t r01, uint32_t r23) {
  return {(int)(r01 & 0xffffu), (int)(r01 >> 16), (int)(r23 & 0xffffu), (int)(r23 >> 16)};
}

// Carves the two workspaces. Pure function of (N, H, W, capacity): forward and backward agree by construction.
struct Layout {
  int N, H, W;
  int64_t cap;
  int gx, gy, ntiles;      // 8x8 tile grid
  int sgx, sgy;            // 16x16 super-tile grid (the reference's tiles; four of our 8x8 tiles each)
  int tile_bits;
  // saved
  size_t o_hdr, o_tile_count, o_grec, o_point_list, o_ranges,
      o_tile_maxc, o_final_T, o_tile_order, o_block_touched, o_block_vis, o_block_base_t, o_block_base_v, o_vis_list,
      o_seg_list, o_vismask, saved_bytes, zero_bytes;
  // scratch (forward)
  size_t o_entries, o_bucket, o_ovf;
  // scratch (backward)
  size_t o_partials, o_tau_part, o_gradrec, o_taurec, scratch_bytes;
  int pre_blocks, nseg;

  __host__ Layout(int N_, int H_, int W_, int64_t cap_) : N(N_), H(H_), W(W_), cap(cap_) {
    gx = (W + kTile - 1) / kTile;
    gy = (H + kTile - 1) / kTile;
    ntiles = gx * gy;
    sgx = (W + kRefTile - 1) / kRefTile;
    sgy = (H + kRefTile - 1) / kRefTile;
    tile_bits = 1;
    while ((1ll << tile_bits) < (long long)ntiles + 1) ++tile_bits;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes); return r; };
    size_t n = (size_t)(N > 0 ? N : 1), c = (size_t)(cap > 0 ? cap : 1);
    pre_blocks = (N + 255) / 256;
    nseg = (N + kSeg - 1) / kSeg;
    size_t nb = (size_t)(pre_blocks > 0 ? pre_blocks : 1);
    o_hdr = take(sizeof(SavedHeader));
    o_tile_count = take((size_t)((gy + 1) / 2) * ((gx + 1) / 2) * 4 * kCntSlotWords);     // hdr + tile_count are zeroed by ONE launch (fresh blocks)
    zero_bytes = o;
    o_grec = take(n * sizeof(GRec));   // one 64-byte record per Gaussian: what binning / blending / backward gather
    o_point_list = take(c * 4);
    o_ranges = take((size_t)ntiles * 8 * kRngStride);
    o_tile_maxc = take((size_t)ntiles * 4);
    o_final_T = take((size_t)ntiles * 64 * 8);   // per pixel (final T, last contributor), TILE-major: a wave's 64 pixels are 512 contiguous bytes
    o_tile_order = take((size_t)sgx * sgy * 4 + 16);   // the launch order of the view's super tiles: longest lists first (K2 writes it, every tile
                                                       // kernel of this forward AND its backward reads it: tile_of_block, sgr_blend.hip)
    o_block_touched = take(nb * 4);
    o_block_vis = take(nb * 4);
    o_block_base_t = take(nb * 4);
    o_block_base_v = take(nb * 4);
    o_vis_list = take(n * 4);
    o_seg_list = take(n * 4);        // per-segment visible lists written by K1; K3 turns them into the compact o_vis_list
    o_vismask = take(n * 4 * kK1MaxParts);   // per Gaussian: the views of the BATCH that see it (bit v; kept in the batch's first saved block;
                                             // one word per K1 view part, OR-ed by the reader)
    saved_bytes = o;

    o = 0;
    o_entries = take(c * 8);
    o_bucket = take((size_t)ntiles * kBucket * 8);
    o_ovf = take(c * sizeof(OvfEntry));
    // (the backward arrays FOLLOW the forward ones: in the fused tile kernel one tile's wave writes its partials while the
    // waves of other tiles still read their buckets / runs)
    o_partials = take(c * 48);
    o_tau_part = take((size_t)(pre_blocks > 0 ? pre_blocks : 1) * 6 * 4);
    o_gradrec = take(n * 64);          // per visible Gaussian: 16-float gradient record of this view
    o_taurec = take(n * 24);           // per visible Gaussian: its 6 pose-gradient terms (only when requested)
    scratch_bytes = o;
  }
  __host__ LOff dev() const {
    LOff d;
    d.N = N; d.H = H; d.W = W; d.gx = gx; d.gy = gy; d.gxp = (gx + 1) / 2; d.sgx = sgx; d.sgy = sgy; d.ntiles = ntiles; d.pre_blocks = pre_blocks; d.nseg = nseg; d.dbg = debug_flags(); d.mean_hint = 0; d.k1_parts = k1_parts_for(nseg); d.sgx_magic = (uint32_t)((1ull << 32) / (uint64_t)(sgx > 0 ? sgx : 1)) + 1u;
    d.cap = cap;
    d.o_hdr = o_hdr; d.o_tile_count = o_tile_count; d.o_grec = o_grec;
    d.o_point_list = o_point_list; d.o_ranges = o_ranges; d.o_tile_maxc = o_tile_maxc; d.o_final_T = o_final_T;
    d.o_tile_order = o_tile_order; d.o_block_touched = o_block_touched; d.o_block_vis = o_block_vis;
    d.o_block_base_t = o_block_base_t; d.o_block_base_v = o_block_base_v; d.o_vis_list = o_vis_list;
    d.o_seg_list = o_seg_list; d.o_vismask = o_vismask; d.o_entries = o_entries; d.o_bucket = o_bucket; d.o_ovf = o_ovf; d.o_partials = o_partials; d.o_tau_part = o_tau_part;
    d.o_gradrec = o_gradrec; d.o_taurec = o_taurec;
    return d;
  }
};

// per-view pointers of the fused mapping loss (sgr_mapping_loss / sgr_map_views)
struct LossTab {
  const float* image[kMaxViews];
  const float* depth[kMaxViews];
  const float* gt_image[kMaxViews];
  const float* gt_depth[kMaxViews];
  const float* exp_a[kMaxViews];
  const float* exp_b[kMaxViews];
  float* loss[kMaxViews];
  float* dimage[kMaxViews];
  float* ddepth[kMaxViews];
  float* da[kMaxViews];
  float* db[kMaxViews];
  void* parts[kMaxViews];
};
void launch_mapping_loss(const LossTab& tab, int nviews, int HW, float alpha, float thr, float upstream, hipStream_t st);
// second stage only: `nparts` partial sums per view were already written (by blend_fwd's fused loss epilogue)
void launch_mapping_loss_final(const LossTab& tab, int nviews, int HW, int nparts, float alpha, hipStream_t st);
struct LossPart { float rgb, dep, da, db; };
struct LossCoef { float w_rgb, w_dep, thr; };

// ---- Adam constants of one step (host side computes the bias corrections in double like torch.optim.Adam does)
struct AdamConst { float b1, b2, eps; float bc2_sqrt[5], step_size[5]; };
// xyz, f_dc, opacity, scaling, rotation.  Group k is stepped for Gaussians r0[k] <= i < r1[k] only (the whole map unless a
// rank owns a slice of the optimiser: sgr_gaussian_adam_shard); outside that range its pointers are never dereferenced.
struct AdamGroups { SgrAdamGroup g[5]; int64_t r0[5], r1[5]; };
// gradient gather (over the views of a batch) + activation chain rule + isotropy + Adam + next activations in ONE
// pass over the Gaussians: the tail of a single-GPU mapping iteration
struct FusedAdam {
  AdamGroups G;
  AdamConst c;
  float iso_coef;
  int grads_clean;            // gradient sinks are all-zero on entry: neither read nor written
  int gather_only;            // no optimiser step: the gathered sums are ADDED to (1) / STORED in (2) the sinks (multi-GPU: an exchange follows)
  float *s_out, *r_out, *o_out;
  float *stat_accum, *stat_denom, *stat_maxr;
  // optional riders of the same launch (one extra block): the fixed-order sum of the per-tile loss parts of every view
  // (mapping_loss_final) and the exposure (keyframe) Adam step that consumes the exposure gradients it produces
  int tail_views, tail_nparts;
  float tail_inv_rgb, tail_inv_dep, tail_alpha;
  const void* tail_parts[kMaxViews];
  float* tail_loss[kMaxViews];
  float* tail_da[kMaxViews];
  float* tail_db[kMaxViews];
  int exp_rows, exp_width;
  float* exp_param; const float* exp_grad; float* exp_avg; float* exp_avg_sq; int32_t* exp_step; const int32_t* exp_active;
  float exp_lr, exp_b1, exp_b2, exp_eps;
};
int make_fused_adam(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps, float iso_weight, FusedAdam* out);
int gaussian_adam_step_act(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps, float iso_weight,
                           float* s_out, float* r_out, float* o_out, void* stream);
void launch_gather_adam(const ViewTab& tab, int nviews, const LOff& L, const FusedAdam& fa, hipStream_t st);

// ---- optional per-kernel event timing (sgr_profile_enable / sgr_profile_read)
enum ProfKind { PK_PRE_FWD = 0, PK_SCAN, PK_SCATTER, PK_BLEND_FUSED, PK_BLEND_FWD, PK_BLEND_BWD, PK_PRE_BWD };
void prof_begin(int kind, hipStream_t st);
void prof_end(int kind, hipStream_t st);
struct ProfScope {
  int kind; hipStream_t st;
  ProfScope(int k, hipStream_t s) : kind(k), st(s) { prof_begin(kind, st); }
  ~ProfScope() { prof_end(kind, st); }
};

// absolute partial-slot offset of Gaussian g: in-segment prefix (written by preprocess_fwd) + its segment's base (tile_scan)
__device__ __forceinline__ uint32_t abs_offset(const char* saved, const LOff& L, uint32_t g, uint32_t rel_offset) {
  return rel_offset + ((const uint32_t*)(saved + L.o_block_base_t))[g >> kSegShift];
}
__device__ __forceinline__ const GRec* grec_of(const char* saved, const LOff& L) { return (const GRec*)(saved + L.o_grec); }

// 3-float rows as ONE 12-byte access (three dword accesses with a 12-byte lane stride use a third of each cache line)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* p) { return *(const F3*)p; }
__device__ __forceinline__ void st3(float* p, F3 v) { *(F3*)p = v; }

// ---- tiny fixed-size linear algebra on registers
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// ---- wave64 DPP scans (gfx9 DPP: row_shr within 16 lanes, row_bcast across rows)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                __builtin_bit_cast(int, v), CTRL, ROW_MASK,
                                                                BANK_MASK, false));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// inclusive prefix sum over lanes 0..63
__device__ __forceinline__ float wave_scan_add(float v) {
  v += dpp_f<DPP_ROW_SHR1>(0.f, v);
  v += dpp_f<DPP_ROW_SHR2>(0.f, v);
  v += dpp_f<DPP_ROW_SHR4>(0.f, v);
  v += dpp_f<DPP_ROW_SHR8>(0.f, v);
  v += dpp_f<DPP_ROW_BCAST15, 0xa>(0.f, v);
  v += dpp_f<DPP_ROW_BCAST31, 0xc>(0.f, v);
  return v;
}
// inclusive prefix product over lanes 0..63
__device__ __forceinline__ float wave_scan_mul(float v) {
  v *= dpp_f<DPP_ROW_SHR1>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR2>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR4>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR8>(1.f, v);
  v *= dpp_f<DPP_ROW_BCAST15, 0xa>(1.f, v);
  v *= dpp_f<DPP_ROW_BCAST31, 0xc>(1.f, v);
  return v;
}
// inclusive prefix sum of unsigned ints over lanes 0..63
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
  v += dpp_u<DPP_ROW_SHR1>(v);
  v += dpp_u<DPP_ROW_SHR2>(v);
  v += dpp_u<DPP_ROW_SHR4>(v);
  v += dpp_u<DPP_ROW_SHR8>(v);
  v += dpp_u<DPP_ROW_BCAST15, 0xa>(v);
  v += dpp_u<DPP_ROW_BCAST31, 0xc>(v);
  return v;
}
// maximum of a non-negative int over the wave (in every lane)
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  return v;
}
// exclusive prefix of `v` inside a 256-thread block (4 waves) + block total; `red` = 4 uints of LDS
__device__ __forceinline__ uint32_t block256_exclusive_scan(uint32_t v, uint32_t* red, uint32_t& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = wave_scan_add_u32(v);
  if (lane == 63) red[wv] = inc;
  __syncthreads();
  uint32_t w0 = red[0], w1 = red[1], w2 = red[2], w3 = red[3];
  __syncthreads();
  total = w0 + w1 + w2 + w3;
  uint32_t base = (wv > 0 ? w0 : 0u) + (wv > 1 ? w1 : 0u) + (wv > 2 ? w2 : 0u);
  return base + inc - v;
}

// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ float wave_shr1(float v, float fill) { return dpp_f<DPP_WAVE_SHR1>(fill, v); }

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {   // total in every lane
  v = wave_scan_add(v);
  return readlane_f(v, 63);
}

}  // namespace sgr
