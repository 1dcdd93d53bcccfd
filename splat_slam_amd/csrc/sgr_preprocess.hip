// Per-Gaussian stages of the rasterizer for gfx950: projection / EWA footprint / tile rectangle (forward) and the
// whole chain rule from blend gradients back to (mean, scale, rotation, SH, opacity, camera pose) (backward).
//
// Replaces preprocessCUDA / computeCov2DCUDA of the un-vendored diff-gaussian-rasterization-w-pose module
// (/root/reference/.gitmodules:4-6, README.md:88-92) that GaussianRasterizer.forward reaches from
// /root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141.
//
// MI355X notes: both kernels are HBM-streaming (one pass over the Gaussian SoA, 64-wide waves, 256-thread blocks).
// Nothing that can be recomputed from the inputs is stored for the backward (cov3D, cov2D, J, T are rebuilt in
// registers) -- HBM bytes, not flops, bound these kernels.  Per-Gaussian pose gradients are reduced in-kernel
// (wave DPP + LDS) to one 6-vector per block, then by a fixed-order second stage: deterministic, no [N,6] buffer.
#include "sgr_common.h"

namespace sgr {

struct Cam {
  float vm[16];   // viewmatrix   (transposed layout: W2C[r][c] = vm[c*4+r])
  float pm[16];   // projmatrix
};

__device__ __forceinline__ void load16(const float* __restrict__ src, float* dst) {
#pragma unroll
  for (int i = 0; i < 16; ++i) dst[i] = src[i];
}

__device__ __forceinline__ void quat_to_R(const float q[4], float R[3][3]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma (symmetric 3x3 as 6 floats: 00 01 02 11 12 22)
__device__ __forceinline__ void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float S6[6]) {
  float R[3][3];
  quat_to_R(q, R);
  float M[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) M[i][k] = R[i][k] * (mod * s[k]);
  S6[0] = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
  S6[1] = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
  S6[2] = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
  S6[3] = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
  S6[4] = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
  S6[5] = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
}

struct Ewa {
  float tx, ty, tz;          // clamped view-space mean used by the Jacobian
  bool clamp_x, clamp_y;
  float T[2][3];             // J * W
  float a, b, c;             // dilated 2D covariance
};

// first half of the EWA projection: clamped view-space mean and T = J * W
__device__ __forceinline__ void ewa_T(const float pv[3], const float* vm, float fx, float fy, float limx, float limy, Ewa& e) {
  float tz = pv[2];
  float txtz = pv[0] / tz, tytz = pv[1] / tz;
  e.clamp_x = (txtz < -limx) || (txtz > limx);
  e.clamp_y = (tytz < -limy) || (tytz > limy);
  e.tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  e.ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  e.tz = tz;
  float itz = 1.f / tz;
  float J00 = fx * itz, J02 = -fx * e.tx * itz * itz;
  float J11 = fy * itz, J12 = -fy * e.ty * itz * itz;
  // W[r][c] = vm[c*4+r]
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    e.T[0][k] = J00 * vm[k * 4 + 0] + J02 * vm[k * 4 + 2];
    e.T[1][k] = J11 * vm[k * 4 + 1] + J12 * vm[k * 4 + 2];
  }
}

__device__ __forceinline__ void ewa_project(const float pv[3], const float* vm, const float S6[6], float fx, float fy,
                                            float limx, float limy, Ewa& e) {
  ewa_T(pv, vm, fx, fy, limx, limy, e);
  float S[3][3] = {{S6[0], S6[1], S6[2]}, {S6[1], S6[3], S6[4]}, {S6[2], S6[4], S6[5]}};
  float TS[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) TS[i][k] = e.T[i][0] * S[0][k] + e.T[i][1] * S[1][k] + e.T[i][2] * S[2][k];
  e.a = TS[0][0] * e.T[0][0] + TS[0][1] * e.T[0][1] + TS[0][2] * e.T[0][2] + kDilation;
  e.b = TS[0][0] * e.T[1][0] + TS[0][1] * e.T[1][1] + TS[0][2] * e.T[1][2];
  e.c = TS[1][0] * e.T[1][0] + TS[1][1] * e.T[1][1] + TS[1][2] * e.T[1][2] + kDilation;
}

// SH -> RGB (+0.5, clamp >= 0).  sh: [M,3] of this Gaussian.  Returns clamp bits.
__device__ __forceinline__ unsigned sh_to_rgb(int deg, const float* __restrict__ sh, const float dir[3], float rgb[3]) {
  float x = dir[0], y = dir[1], z = dir[2];
  unsigned clamped = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float r = SH_C0 * sh[c];
    if (deg > 0) {
      r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r += SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.f * zz - xx - yy) * sh[18 + c] +
             SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
        if (deg > 2) {
          r += SH_C3[0] * y * (3.f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
               SH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
               SH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
               SH_C3[6] * x * (xx - 3.f * yy) * sh[45 + c];
        }
      }
    }
    r += 0.5f;
    if (r < 0.f) { clamped |= 1u << c; r = 0.f; }
    rgb[c] = r;
  }
  return clamped;
}

// ---------------------------------------------------------------------------------------------- forward
struct PreOut {
  bool visible;      // radii > 0
  float rad, px, py, A, B, C, opac, depth, rgb[3];
  int x0, y0, x1, y1;
  unsigned clamped;
};

struct PreIn {
  float p[3], opac, S6[6], c_in[3];
};

__device__ __forceinline__ void preprocess_view(const PreIn& in, int i, int H, int W, int deg, int M, float tanfovx,
                                                float tanfovy, const float* vm, const float* pm,
                                                const float* __restrict__ campos, const float* __restrict__ shs,
                                                bool has_colors, int gx, int gy, int sgx, int sgy, PreOut& o) {
  o.visible = false;
  o.x0 = o.x1 = o.y0 = o.y1 = 0;
  const float* p = in.p;
  float pv[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pv[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
  if (!(pv[2] > kNearPlane)) return;
  float ph[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ph[r] = pm[r] * p[0] + pm[4 + r] * p[1] + pm[8 + r] * p[2] + pm[12 + r];
  float pw = 1.f / (ph[3] + 1e-7f);
  float ndcx = ph[0] * pw, ndcy = ph[1] * pw;
  float fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy);
  Ewa e;
  ewa_project(pv, vm, in.S6, fx, fy, 1.3f * tanfovx, 1.3f * tanfovy, e);
  float det = e.a * e.c - e.b * e.b;
  if (det == 0.f) return;
  float det_inv = 1.f / det;
  float A = e.c * det_inv, B = -e.b * det_inv, C = e.a * det_inv;
  float mid = 0.5f * (e.a + e.c);
  float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
  float lam = fmaxf(mid + disc, mid - disc);
  float rad = ceilf(3.f * sqrtf(lam));
  float px = ((ndcx + 1.f) * W - 1.f) * 0.5f, py = ((ndcy + 1.f) * H - 1.f) * 0.5f;
  if (!(isfinite(px) && isfinite(py) && isfinite(rad))) return;

  // which pixels this splat may reach is defined by the reference's 16x16 tile rectangle
  int rx0 = min(sgx, max(0, (int)((px - rad) / (float)kRefTile)));
  int ry0 = min(sgy, max(0, (int)((py - rad) / (float)kRefTile)));
  int rx1 = min(sgx, max(0, (int)((px + rad + (kRefTile - 1)) / (float)kRefTile)));
  int ry1 = min(sgy, max(0, (int)((py + rad + (kRefTile - 1)) / (float)kRefTile)));
  if ((rx1 - rx0) * (ry1 - ry0) == 0) return;

  const float opac = in.opac;
  // our bins are 8x8 (one wave): refine the rectangle, then drop bins no pixel of which can pass alpha >= 1/255.
  int x0 = min(gx, 2 * rx0), x1 = min(gx, 2 * rx1), y0 = min(gy, 2 * ry0), y1 = min(gy, 2 * ry1);
  {
    float detc = A * C - B * B;
    float o255 = 255.f * opac;
    if (!(o255 >= 1.f)) {
      x1 = x0;   // alpha = min(.99, o*G) <= o < 1/255 everywhere: contributes to no pixel
    } else if (A > 0.f && C > 0.f && detc > 0.f) {
      float tau = 2.f * __logf(o255) * 1.001f + 0.02f;     // q = -2*power <= tau is necessary for alpha >= 1/255
      float ex = sqrtf(tau * C / detc), ey = sqrtf(tau * A / detc);   // half extents of {q <= tau}
      float ext = fmaxf(ex, ey) + 2.f * kTile;
      float err = 4e-7f * (A + C + 2.f * fabsf(B)) * ext * ext;       // fp32 rounding of q near the boundary
      if (err < 0.005f && isfinite(ex) && isfinite(ey)) {
        int cx0 = (int)ceilf((px - ex - (kTile - 1)) / (float)kTile - 1e-3f);
        int cx1 = (int)floorf((px + ex) / (float)kTile + 1e-3f) + 1;
        int cy0 = (int)ceilf((py - ey - (kTile - 1)) / (float)kTile - 1e-3f);
        int cy1 = (int)floorf((py + ey) / (float)kTile + 1e-3f) + 1;
        x0 = max(x0, cx0); x1 = min(x1, cx1); y0 = max(y0, cy0); y1 = min(y1, cy1);
        if (x1 < x0) x1 = x0;
        if (y1 < y0) y1 = y0;
      }
    }
  }

  o.clamped = 0;
  if (has_colors) {
    o.rgb[0] = in.c_in[0]; o.rgb[1] = in.c_in[1]; o.rgb[2] = in.c_in[2];
  } else if (deg == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float r = SH_C0 * in.c_in[c] + 0.5f;
      if (r < 0.f) { o.clamped |= 1u << c; r = 0.f; }
      o.rgb[c] = r;
    }
  } else {
    float dir[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
    float inv = 1.f / sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] *= inv; dir[1] *= inv; dir[2] *= inv;
    o.clamped = sh_to_rgb(deg, shs + (size_t)i * M * 3, dir, o.rgb);
  }
  o.visible = true;
  o.rad = rad; o.px = px; o.py = py; o.A = A; o.B = B; o.C = C; o.opac = opac; o.depth = pv[2];
  o.x0 = x0; o.y0 = y0; o.x1 = x1; o.y1 = y1;
}

// Cheap, CONSERVATIVE visibility test (phase A of K1): false only if preprocess_view is certain to reject the Gaussian.
// Same near-plane test; the reference-tile rectangle is evaluated with a radius bound rad_b >= rad + 1:
//   lam = mid + sqrt(max(0.1, mid^2 - det)) <= a + c + 0.317,   a + c = tr(T Sigma T^T) + 0.6 <= |T|_F^2 tr(Sigma) + 0.6
// (trS = an upper bound of tr(Sigma)); a larger radius only grows the rectangle, so "empty with rad_b" implies "empty".
__device__ __forceinline__ bool maybe_visible(const float p[3], float trS, int H, int W, float tanfovx, float tanfovy,
                                              const float* vm, const float* pm, int sgx, int sgy) {
  float pv[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pv[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
  if (!(pv[2] > kNearPlane)) return false;
  float ph0 = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
  float ph1 = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
  float ph3 = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
  float pw = __builtin_amdgcn_rcpf(ph3 + 1e-7f);            // (approximate: the +1 pixel / 0.1 % margins absorb it)
  float px = ((ph0 * pw + 1.f) * W - 1.f) * 0.5f, py = ((ph1 * pw + 1.f) * H - 1.f) * 0.5f;
  float fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy);
  // |T|_F^2 of T = J W with the reference's clamped Jacobian, W rows from the view matrix
  const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  const float itz = __builtin_amdgcn_rcpf(pv[2]);
  const float cx = fminf(limx, fmaxf(-limx, pv[0] * itz)), cy = fminf(limy, fmaxf(-limy, pv[1] * itz));
  const float J00 = fx * itz, J02 = -J00 * cx, J11 = fy * itz, J12 = -J11 * cy;
  float tn = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t0 = J00 * vm[k * 4 + 0] + J02 * vm[k * 4 + 2], t1 = J11 * vm[k * 4 + 1] + J12 * vm[k * 4 + 2];
    tn += t0 * t0 + t1 * t1;
  }
  float rad_b = ceilf(3.f * __builtin_amdgcn_sqrtf(tn * trS * 1.002f + 1.0f)) + 1.f;
  if (!(isfinite(px) && isfinite(py) && isfinite(rad_b))) return true;      // let the exact path decide
  int rx0 = min(sgx, max(0, (int)((px - rad_b) * (1.f / kRefTile))));
  int ry0 = min(sgy, max(0, (int)((py - rad_b) * (1.f / kRefTile))));
  int rx1 = min(sgx, max(0, (int)((px + rad_b + (kRefTile - 1)) * (1.f / kRefTile))));
  int ry1 = min(sgy, max(0, (int)((py + rad_b + (kRefTile - 1)) * (1.f / kRefTile))));
  return (rx1 - rx0) * (ry1 - ry0) != 0;
}

// The same question as four half-space tests (what phase A of K1 runs per (view, Gaussian): ~30 instructions instead of ~100).
// The rectangle of preprocess_view is non-empty only if  px + rad >= 1,  px - rad < 16 sgx  (and the same in y), with
//   rad = ceil(3 sqrt(lam)) <= 3 sqrt(|T|_F^2 trS + 0.917) + 1 <= 3 sqrt(|T|_F^2 trS) + 3.88,
//   |T|_F^2 = |J W|_F^2 <= gmax |J|_F^2,  |J|_F^2 <= (fx^2 (1 + limx^2) + fy^2 (1 + limy^2)) / z^2   (the clamped Jacobian),
// gmax = a Gershgorin bound of the largest eigenvalue of W W^T (1 for a rigid view matrix).  So rad <= K sqrt(trS) / z + 3.88
// with a per-view constant K, and with px = hx ph0 / w + hx - 1/2 (w = ph3 + 1e-7 > 0, z > 0) each condition becomes linear
// after multiplying by w z:   (hx ph0 + (hx + 4.5) w) z + K s w >= 0   etc. (5 px of slack instead of 3.88 + rounding).
// Conservative like maybe_visible (a few per cent more candidates near the image border); anything not finite passes.
__device__ __forceinline__ float cull_radius_factor(const float* vm, int H, int W, float tanfovx, float tanfovy) {
  float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {       // G = W W^T, W[r][c] = vm[c*4+r]
      const float gij = vm[i] * vm[j] + vm[4 + i] * vm[4 + j] + vm[8 + i] * vm[8 + j];
      g[i] += fabsf(gij);
    }
  const float gmax = fmaxf(g[0], fmaxf(g[1], g[2]));
  const float fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy), limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  return 3.f * sqrtf(gmax * (fx * fx * (1.f + limx * limx) + fy * fy * (1.f + limy * limy))) * 1.003f;
}
__device__ __forceinline__ bool maybe_visible_planes(const float p[3], float s, const float* vm, const float* pm, float K, int H, int W,
                                                     int sgx, int sgy) {
  const float z = vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14];
  if (!(z > kNearPlane * 0.999f)) return false;
  const float w = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15] + 1e-7f;
  if (!(w > 0.f)) return true;                                   // let the exact path decide
  const float ph0 = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
  const float ph1 = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
  const float hx = 0.5f * (float)W, hy = 0.5f * (float)H, ksw = K * s * w;
  const float ax = hx * ph0, ay = hy * ph1;
  const float l = (ax + (hx + 4.5f) * w) * z + ksw;                              // px + rad_b >= 0
  const float r = (ax + (hx - 5.5f - (float)(sgx * kRefTile)) * w) * z - ksw;    // px - rad_b < 16 sgx
  const float t = (ay + (hy + 4.5f) * w) * z + ksw;
  const float b = (ay + (hy - 5.5f - (float)(sgy * kRefTile)) * w) * z - ksw;
  return !(l < 0.f) && !(r >= 0.f) && !(t < 0.f) && !(b >= 0.f);
}

// K1.  grid = ceil(N/256) blocks of 256 threads; a block owns a SEGMENT of 256 consecutive Gaussians for ALL views of
// the batch (<= 16):
//   phase A  thread = Gaussian: its position / scale / rotation are loaded ONCE and tested against every view with
//            maybe_visible() (per-view launches re-read the map once per view: 12 x 40 B x N was the kernel's HBM
//            floor).  A SLAM view sees a few % of the map, so the survivors are compacted (wave ballots + one LDS
//            exchange, deterministic order) into one candidate list per view;
//   phase B  thread = (view, candidate) pair, all views back to back, so the lanes are dense: full projection,
//            footprint, 8x8 bin rectangle, per-tile pair counts (fire-and-forget atomics).  A block-wide scan over the
//            flattened pairs, rebased at every view boundary, gives each visible Gaussian its in-segment prefix of
//            pairs (-> partial-slot offsets, finished by tile_scan_kernel) and its slot in the (view, segment) visible
//            list seg_list[seg*256 + k]; scatter_kernel later concatenates those lists.
// Everything is fixed-order: results are bitwise reproducible.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) preprocess_fwd_kernel(
    ViewTab tab, int nviews, LOff L, Common cm, const float* __restrict__ means3D, const float* __restrict__ opacities,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp) {
  __shared__ float mats[kMaxViews][32];            // viewmatrix | projmatrix of every view
  __shared__ char* p_saved[kMaxViews];
  __shared__ int32_t* p_radii[kMaxViews];
  __shared__ int32_t* p_ntouched[kMaxViews];
  __shared__ const float* p_campos[kMaxViews];
  __shared__ char* p_scratch[kMaxViews];
  __shared__ const float* p_projraw[kMaxViews];
  __shared__ uint8_t cand[kMaxViews][kSeg];
  __shared__ uint32_t wtot[kMaxViews][4];
  __shared__ uint32_t vstart[kMaxViews + 1];
  __shared__ uint32_t vbase_t[kMaxViews + 1], vbase_v[kMaxViews + 1];
  __shared__ uint32_t red[4];
  __shared__ uint32_t vis_bits[kSeg];            // per Gaussian of the segment: bit v = view v of the batch sees it
  __shared__ uint32_t op_ex[kSeg];               // load-balanced counting atomics: per owner thread, prefix of its remaining operations,
  __shared__ uint4 op_rect[kSeg];                // ... its rectangle (x0 | x1 << 16, y0 | y1 << 16), view, Gaussian
  __shared__ uint32_t op_depth[kSeg];            // ... the depth half of its key
  __shared__ float4 op_foot[kSeg][2];            // ... and its footprint (exact bin test): px py A B | C 1/A 1/C thr
  __shared__ float cullK[kMaxViews];             // per view: radius bound factor of the plane test (see cull_planes)
  __shared__ float seg_box[4][8];                // per wave: min xyz, max xyz, max trS of its 64 Gaussians
  __shared__ uint32_t seg_views;                 // bit v = some Gaussian of this segment MAY be visible in view v
  const int N = L.N;
  const int seg = blockIdx.x, seg0 = seg * kSeg;
  const int part = blockIdx.y, nparts = gridDim.y;      // this block's views: v % nparts == part
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

  // ---- per-view constants into LDS (constant indices only: a dynamically indexed by-value struct would go to scratch)
#pragma unroll
  for (int u = 0; u < kMaxViews; ++u) {
    if (u < nviews) {
      if (tid < 16) mats[u][tid] = tab.viewmatrix[u][tid];
      else if (tid < 32) mats[u][tid] = tab.projmatrix[u][tid - 16];
      if (tid == 32) { p_saved[u] = tab.saved[u]; p_radii[u] = tab.radii[u]; p_ntouched[u] = tab.n_touched[u]; p_campos[u] = tab.campos[u]; p_scratch[u] = tab.scratch[u]; p_projraw[u] = tab.projraw[u]; }
    }
  }
  if (tid <= kMaxViews) { vbase_t[tid] = 0u; vbase_v[tid] = 0u; }
  if (tid == 0) seg_views = 0xffffffffu;         // (every view, unless the whole-segment test below says otherwise)
  vis_bits[tid] = 0u;
  __syncthreads();
  if (tid < nviews) cullK[tid] = cull_radius_factor(mats[tid], L.H, L.W, cm.tanfovx, cm.tanfovy);
  __syncthreads();

  // ---- phase A
  const int ia = seg0 + tid;
  uint32_t passbits = 0, ranks[4] = {0u, 0u, 0u, 0u};      // rank of this lane among its wave's survivors, 8 bits per view
  {
    float p[3] = {0.f, 0.f, 0.f};
    float trS = 0.f;
    if (ia < N) {
      { const F3 t = ld3(means3D + 3 * ia); p[0] = t.x; p[1] = t.y; p[2] = t.z; }
      if (cov3D_precomp) {
        trS = cov3D_precomp[6 * ia] + cov3D_precomp[6 * ia + 3] + cov3D_precomp[6 * ia + 5];
      } else {
        const F3 sc3 = ld3(scales + 3 * ia);
        const float s0 = sc3.x, s1 = sc3.y, s2 = sc3.z;
        const float4 q = *(const float4*)(rotations + 4 * ia);
        const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
        if (fabsf(n2 - 1.f) < 1e-3f) {     // R(q) = (1-n2) I + n2 R(q/|q|): column norms <= 1.002
          trS = cm.mod * cm.mod * (s0 * s0 + s1 * s1 + s2 * s2) * 1.01f;
        } else {
          float s_in[3] = {s0, s1, s2}, q_in[4] = {q.x, q.y, q.z, q.w}, S6[6];
          cov3d_from_scale_rot(s_in, cm.mod, q_in, S6);
          trS = (S6[0] + S6[3] + S6[5]) * 1.001f;
        }
      }
    }
    // Whole-segment test first.  A map that grows keyframe by keyframe (extend_from_pcd_seq appends each keyframe's points) is
    // spatially coherent: most 256-Gaussian segments lie entirely outside most views.  The segment's bounding box + the
    // largest tr(Sigma) give, per view, an interval bound of exactly the quantities maybe_visible() tests (view-space
    // depth, projected centre, radius bound): if even those intervals miss the image the per-Gaussian tests of that view are
    // skipped.  Conservative by construction (every bound is widened, 0.1 % + 1 px); a randomly ordered map just pays
    // the ~100 instructions of the reduction.
    if (!(L.dbg & 4)) {
      const bool in = ia < N && isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && isfinite(trS);
      const bool bad = ia < N && !in;                       // non-finite input: let the exact path deal with it
      float lo[3], hi[3], ts = in ? trS : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) { lo[k] = in ? p[k] : 3.0e38f; hi[k] = in ? p[k] : -3.0e38f; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
        ts = fmaxf(ts, __shfl_xor(ts, off));
      }
      const unsigned long long anybad = __ballot(bad);
      if (lane == 0) {
        seg_box[wv][0] = lo[0]; seg_box[wv][1] = lo[1]; seg_box[wv][2] = lo[2];
        seg_box[wv][3] = hi[0]; seg_box[wv][4] = hi[1]; seg_box[wv][5] = hi[2];
        seg_box[wv][6] = ts; seg_box[wv][7] = anybad ? 1.f : 0.f;
      }
      __syncthreads();
      if (tid == 0) seg_views = 0u;
      __syncthreads();
      if (tid < nviews) {
        float b0[3], b1[3], tsm = 0.f, anyb = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) { b0[k] = 3.0e38f; b1[k] = -3.0e38f; }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
          for (int k = 0; k < 3; ++k) { b0[k] = fminf(b0[k], seg_box[w][k]); b1[k] = fmaxf(b1[k], seg_box[w][3 + k]); }
          tsm = fmaxf(tsm, seg_box[w][6]); anyb = fmaxf(anyb, seg_box[w][7]);
        }
        bool maybe = true;
        if (anyb == 0.f && b0[0] <= b1[0]) {
          const float* vm = mats[tid];
          // view-space box of the 8 corners (W2C[r][c] = vm[c*4+r])
          float v0[3], v1[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            float a = vm[12 + r], bsum = vm[12 + r];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float m = vm[c * 4 + r];
              a += fminf(m * b0[c], m * b1[c]);
              bsum += fmaxf(m * b0[c], m * b1[c]);
            }
            const float pad = 1e-3f * (fabsf(a) + fabsf(bsum)) + 1e-4f;
            v0[r] = a - pad; v1[r] = bsum + pad;
          }
          if (v1[2] <= kNearPlane) {
            maybe = false;                                   // the whole box is behind the near plane
          } else if (v0[2] > kNearPlane) {
            const float fx = L.W / (2.f * cm.tanfovx), fy = L.H / (2.f * cm.tanfovy);
            const float iz0 = 1.f / v0[2], iz1 = 1.f / v1[2];
            // x / z and y / z over the box (z > 0): extremes at the corners
            const float xz0 = fminf(fminf(v0[0] * iz0, v0[0] * iz1), fminf(v1[0] * iz0, v1[0] * iz1));
            const float xz1 = fmaxf(fmaxf(v0[0] * iz0, v0[0] * iz1), fmaxf(v1[0] * iz0, v1[0] * iz1));
            const float yz0 = fminf(fminf(v0[1] * iz0, v0[1] * iz1), fminf(v1[1] * iz0, v1[1] * iz1));
            const float yz1 = fmaxf(fmaxf(v0[1] * iz0, v0[1] * iz1), fmaxf(v1[1] * iz0, v1[1] * iz1));
            // pixel centre px = fx * x/z + (W/2)(P02 + 1) - 0.5 (P[r][c] = projraw[c*4+r]); radius bound of maybe_visible with
            // |T|_F^2 = (fx/z)^2 (1 + cx^2) + (fy/z)^2 (1 + cy^2),  |cx| <= 1.3 tanfovx, |cy| <= 1.3 tanfovy
            const float limx = 1.3f * cm.tanfovx, limy = 1.3f * cm.tanfovy;
            const float tn = (fx * fx * (1.f + limx * limx) + fy * fy * (1.f + limy * limy)) * iz0 * iz0;
            const float rad = ceilf(3.f * sqrtf(tn * tsm * 1.01f + 1.0f)) + 3.f;
            const float ox = 0.5f * L.W * (p_projraw[tid][8] + 1.f) - 0.5f, oy = 0.5f * L.H * (p_projraw[tid][9] + 1.f) - 0.5f;
            const float px0 = fx * xz0 + ox - rad - 1.f, px1 = fx * xz1 + ox + rad + (kRefTile - 1) + 1.f;
            const float py0 = fy * yz0 + oy - rad - 1.f, py1 = fy * yz1 + oy + rad + (kRefTile - 1) + 1.f;
            if (isfinite(px0) && isfinite(px1) && isfinite(py0) && isfinite(py1) && isfinite(rad))
              maybe = !(px1 < 0.f || py1 < 0.f || px0 >= (float)(L.sgx * kRefTile) || py0 >= (float)(L.sgy * kRefTile));
          }
        }
        if (maybe) atomicOr(&seg_views, 1u << tid);
      }
      __syncthreads();
    }
    const uint32_t segv = seg_views;
    const float sroot = __builtin_amdgcn_sqrtf(fmaxf(trS, 0.f)) * 1.0005f;      // (NaN stays NaN: the test then passes the Gaussian on)
#pragma unroll 2
    for (int v = 0; v < nviews; ++v) {
      bool pass = false;
      if (ia < N && (v % nparts) == part) {
        if (!(L.dbg & 16)) {
        p_radii[v][ia] = 0;                // outputs of the culled majority; phase B overwrites the visible ones
        if (p_ntouched[v]) p_ntouched[v][ia] = 0;
        }
        if (L.dbg & 2) pass = (p[0] + trS == 12345.678f);
        else if (segv & (1u << v)) pass = (L.dbg & 8) ? maybe_visible(p, trS, L.H, L.W, cm.tanfovx, cm.tanfovy, mats[v], mats[v] + 16, L.sgx, L.sgy)
                                                      : maybe_visible_planes(p, sroot, mats[v], mats[v] + 16, cullK[v], L.H, L.W, L.sgx, L.sgy);
      }
      const unsigned long long m = __ballot(pass);
      if (lane == 0) wtot[v][wv] = (uint32_t)__popcll(m);
      if (pass) {
        passbits |= 1u << v;
        ranks[v >> 2] |= (uint32_t)__popcll(m & ((1ull << lane) - 1ull)) << ((v & 3) * 8);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int v = 0; v < nviews; ++v) {
      vstart[v] = run;
      run += wtot[v][0] + wtot[v][1] + wtot[v][2] + wtot[v][3];
    }
    for (int v = nviews; v <= kMaxViews; ++v) vstart[v] = run;
  }
#pragma unroll 1
  for (int v = 0; v < nviews; ++v) {
    if (passbits & (1u << v)) {
      const uint32_t base = (wv > 0 ? wtot[v][0] : 0u) + (wv > 1 ? wtot[v][1] : 0u) + (wv > 2 ? wtot[v][2] : 0u);
      cand[v][base + ((ranks[v >> 2] >> ((v & 3) * 8)) & 0xffu)] = (uint8_t)tid;
    }
  }
  __syncthreads();
  const uint32_t total = (L.dbg & 1) ? 0u : vstart[kMaxViews];

  // ---- phase B
  uint32_t carry_t = 0, carry_v = 0;
#pragma unroll 1
  for (uint32_t f0 = 0; f0 < total; f0 += 256) {
    const uint32_t f = f0 + tid;
    const bool live = f < total;
    int v = 0;
#pragma unroll
    for (int u = 1; u < kMaxViews; ++u) v += (f >= vstart[u] && u < nviews) ? 1 : 0;
    PreOut o;
    o.visible = false;
    o.x0 = o.x1 = o.y0 = o.y1 = 0;
    int i = 0;
    char* saved = p_saved[v];
    if (live) {
      i = seg0 + (int)cand[v][f - vstart[v]];
      const float* vm = mats[v];                 // read from LDS where used: 32 matrix registers put the loop over its budget
      const float* pm = mats[v] + 16;
      PreIn in;
      { const F3 t = ld3(means3D + 3 * i); in.p[0] = t.x; in.p[1] = t.y; in.p[2] = t.z; }
      in.opac = opacities[i];
      if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; ++k) in.S6[k] = cov3D_precomp[6 * i + k];
      } else {
        const F3 sc3 = ld3(scales + 3 * i);
        const float4 q4 = *(const float4*)(rotations + 4 * i);
        float s_in[3] = {sc3.x, sc3.y, sc3.z};
        float q_in[4] = {q4.x, q4.y, q4.z, q4.w};
        cov3d_from_scale_rot(s_in, cm.mod, q_in, in.S6);
      }
      in.c_in[0] = in.c_in[1] = in.c_in[2] = 0.f;
      if (colors_precomp) { in.c_in[0] = colors_precomp[3 * i]; in.c_in[1] = colors_precomp[3 * i + 1]; in.c_in[2] = colors_precomp[3 * i + 2]; }
      else if (cm.deg == 0) { in.c_in[0] = shs[(size_t)i * cm.M * 3]; in.c_in[1] = shs[(size_t)i * cm.M * 3 + 1]; in.c_in[2] = shs[(size_t)i * cm.M * 3 + 2]; }
      preprocess_view(in, i, L.H, L.W, cm.deg, cm.M, cm.tanfovx, cm.tanfovy, vm, pm, p_campos[v], shs,
                      colors_precomp != nullptr, L.gx, L.gy, L.sgx, L.sgy, o);
    }
    const uint32_t cnt = o.visible ? (uint32_t)((o.x1 - o.x0) * (o.y1 - o.y0)) : 0u;
    const uint32_t vis = o.visible ? 1u : 0u;
    if (o.visible && !(L.dbg & 128)) {
      atomicOr(&vis_bits[i - seg0], 1u << v);
      p_radii[v][i] = (int32_t)o.rad;
      float4* rec = (float4*)((GRec*)(saved + L.o_grec) + i);          // q3 (prefix, list slot) follows after the scans
      rec[0] = make_float4(o.px, o.py, __uint_as_float((uint32_t)o.x0 | ((uint32_t)o.y0 << 16)),
                           __uint_as_float((uint32_t)o.x1 | ((uint32_t)o.y1 << 16)));
      rec[1] = make_float4(o.A, o.B, o.C, o.opac);
      rec[2] = make_float4(o.rgb[0], o.rgb[1], o.rgb[2], o.depth);
    }
    // Count the pairs per tile; the returning atomic is the pair's rank in its tile: while the tile's bucket has room the
    // key is binned right here, beyond that the pair joins the view's overflow list (K3 files it at start(tile) + rank once
    // the tile starts are known).  The kernel's time follows the NUMBER of atomics (they execute at the memory side), so the
    // four 16-bit counters of a 2x2 block of tiles share ONE 64-bit word and a splat takes all its ranks in that block with
    // one atomic: an operation = one 2x2 block of the rectangle.  Every splat issues its first 4 operations itself, right away
    // (a fresh map's splats need no more), and their round trip hides behind the block scans; a converged map's splats need
    // tens to hundreds: the rest is dealt out evenly over the block's threads (item -> owner by a search in the owners'
    // prefix sums), four per thread and round.
    const uint32_t my_rx = (uint32_t)o.x0 | ((uint32_t)o.x1 << 16), my_ry = (uint32_t)o.y0 | ((uint32_t)o.y1 << 16);
    const uint32_t nops = cnt > 0u ? (uint32_t)((((o.x1 - 1) >> 1) - (o.x0 >> 1) + 1) * (((o.y1 - 1) >> 1) - (o.y0 >> 1) + 1)) : 0u;
    op_rect[tid] = make_uint4(my_rx, my_ry, (uint32_t)v, (uint32_t)i);
    op_depth[tid] = __float_as_uint(o.depth);
    // the rectangle is the axis-aligned box of the alpha >= 1/255 level set; the bins of it that the level set itself misses are
    // dropped operation by operation (sgr_common.h: footprint test) -- no atomic, no key, no compositing work for them
    Footprint my_foot{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, -1.f};
    if (o.visible && footprint_worthwhile(o.x1 - o.x0, o.y1 - o.y0)) my_foot = make_footprint(o.px, o.py, o.A, o.B, o.C, o.opac, 0.005f);
    op_foot[tid][0] = make_float4(my_foot.px, my_foot.py, my_foot.A, my_foot.B);
    op_foot[tid][1] = make_float4(my_foot.C, my_foot.invA, my_foot.invC, my_foot.thr);
    // operation k of the rectangle rx = x0 | x1 << 16, ry = y0 | y1 << 16: its 2x2 block (bx, by = the block's first tile, both
    // even) and which of its four tiles the rectangle covers (bit s = (y & 1) * 2 + (x & 1)), packed bx | by << 14 | cover << 28
    // so that ONE register per operation stays live across the atomic's round trip
    auto op_geom = [](uint32_t rx, uint32_t ry, int k, const Footprint& foot) -> uint32_t {
      const int x0 = (int)(rx & 0xffffu), x1 = (int)(rx >> 16), y0 = (int)(ry & 0xffffu), y1 = (int)(ry >> 16);
      const int per_row = ((x1 - 1) >> 1) - (x0 >> 1) + 1;
      // k / per_row for k < 2^20: float quotient of (k + 0.5), exact after one correction step
      int q = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)per_row));
      int rem = k - q * per_row;
      if (rem < 0) { --q; rem += per_row; }
      if (rem >= per_row) { ++q; rem -= per_row; }
      const int bx = ((x0 >> 1) + rem) * 2, by = ((y0 >> 1) + q) * 2;
      const bool c0 = bx >= x0, c1 = bx + 1 < x1, r0 = by >= y0, r1 = by + 1 < y1;
      uint32_t cover = (uint32_t)(r0 && c0) | ((uint32_t)(r0 && c1) << 1) | ((uint32_t)(r1 && c0) << 2) | ((uint32_t)(r1 && c1) << 3);
      if (foot.thr >= 0.f) {
        const float drop = foot.thr * kFootDrop + kFootGuard;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          if (((cover >> s4) & 1u) && footprint_qmin(foot, bx + (s4 & 1), by + (s4 >> 1)) > drop) cover &= ~(1u << s4);
      }
      return (uint32_t)bx | ((uint32_t)by << 14) | (cover << 28);
    };
    // ... and its counting atomic: the old word holds the rank of this splat in each covered tile
    auto issue = [&](int view, uint32_t geo) -> unsigned long long {
      const uint32_t cover = geo >> 28;
      if (cover == 0u || (L.dbg & 32)) return 0ull;  // (every bin of this block dropped)
      const unsigned long long inc = (unsigned long long)(cover & 1u) | ((unsigned long long)((cover >> 1) & 1u) << 16) |
                                     ((unsigned long long)((cover >> 2) & 1u) << 32) | ((unsigned long long)((cover >> 3) & 1u) << 48);
      unsigned long long* c = (unsigned long long*)(p_saved[view] + L.o_tile_count) +
                              tile_counter_word((int)(geo & 0x3fffu), (int)((geo >> 14) & 0x3fffu), L.gxp);
      return atomicAdd(c, (L.dbg & 64) ? 0ull : inc);
    };
    // Binning of a batch of (up to) four operations per lane; desc(jj) -> (on, view, key) of operation jj, geo[jj] its block.
    // A pair whose rank fits the tile's bucket is written there; the others join their view's overflow list.  On a converged
    // map a third of all pairs overflowed the 64-entry buckets of rounds 2-5 (lists of ~100; round 6: kBucket = 256 holds them whole) and the list cursor is ONE word per view, so the
    // append is aggregated as far as it goes: per batch the wave takes ONE returning atomic per view present in it (5-bit-plane
    // ballots give every lane its offset), where one atomic per (operation, tile) -- sixteen dependent round trips per batch
    // -- cost 220 of K1's 395 us.
    struct Desc { int view; uint64_t key; };
    auto desc_of = [&](int owner) -> Desc {          // (owner >= 0) from the owner tables in LDS
      const uint2 vz = *(const uint2*)&op_rect[owner].z;          // (view, Gaussian)
      return Desc{(int)vz.x, ((uint64_t)op_depth[owner] << 32) | vz.y};
    };
    auto consume4 = [&](const unsigned long long old[4], const uint32_t geo[4], const int owner[4]) {
      if (L.dbg & 64) return;
      uint32_t sp = 0u;                         // 4 bits per operation: which of its four tiles overflow
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const bool on = owner[jj] >= 0;
        const Desc d = desc_of(on ? owner[jj] : 0);
        const uint32_t t00 = ((geo[jj] >> 14) & 0x3fffu) * (uint32_t)L.gx + (geo[jj] & 0x3fffu);
        uint64_t* bucket = (uint64_t*)(p_scratch[d.view] + L.o_bucket);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const bool mine = on && ((geo[jj] >> (28 + s4)) & 1u);
          const uint32_t t = t00 + (uint32_t)((s4 >> 1) * L.gx + (s4 & 1));
          const uint32_t r = (uint32_t)(old[jj] >> (16 * s4)) & 0xffffu;
          if (mine && r < (uint32_t)kBucket) bucket[(size_t)t * kBucket + r] = d.key;
          if (mine && r >= (uint32_t)kBucket) sp |= 1u << (4 * jj + s4);
        }
      }
      unsigned long long m = __ballot(sp != 0u);
      if (m == 0ull) return;                                   // (a fresh map: nothing overflows)
      // the slow path is written ROLLED over the four operations (selects instead of register arrays): unrolled it put the
      // whole kernel 50 registers over its budget
      const unsigned long long lt = (1ull << lane) - 1ull;
      auto pick = [](auto a0, auto a1, auto a2, auto a3, int j) { return j == 0 ? a0 : (j == 1 ? a1 : (j == 2 ? a2 : a3)); };
      while (m != 0ull) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
        int mine_v = 0;
        if (sp != 0u) mine_v = desc_of(pick(owner[0], owner[1], owner[2], owner[3], (__ffs((int)sp) - 1) >> 2)).view;
        const int lv = __builtin_amdgcn_readlane(mine_v, leader);
        uint32_t c = 0;                         // this lane's overflowing pairs of view lv: 0..16
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
          const uint32_t nib = (sp >> (4 * jj)) & 15u;
          if (nib != 0u && desc_of(pick(owner[0], owner[1], owner[2], owner[3], jj)).view == lv) c += (uint32_t)__popc(nib);
        }
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) {
          const unsigned long long bal = __ballot((c >> bit) & 1u);
          pre += (uint32_t)__popcll(bal & lt) << bit;
          tot += (uint32_t)__popcll(bal) << bit;
        }
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&((SavedHeader*)(p_saved[lv] + L.o_hdr))->ovf_cursor, tot);
        uint64_t pos = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)base, leader) + pre;
        OvfEntry* ovf = (OvfEntry*)(p_scratch[lv] + L.o_ovf);
#pragma unroll 1
        for (int jj = 0; jj < 4; ++jj) {
          const uint32_t nib = (sp >> (4 * jj)) & 15u;
          if (nib == 0u) continue;
          const Desc d = desc_of(pick(owner[0], owner[1], owner[2], owner[3], jj));
          if (d.view != lv) continue;
          const uint32_t g = pick(geo[0], geo[1], geo[2], geo[3], jj);
          const unsigned long long oj = pick(old[0], old[1], old[2], old[3], jj);
          const uint32_t t00 = ((g >> 14) & 0x3fffu) * (uint32_t)L.gx + (g & 0x3fffu);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            if ((nib >> s4) & 1u) {
              const uint32_t r = (uint32_t)(oj >> (16 * s4)) & 0xffffu;
              if ((int64_t)pos < L.cap) ovf[pos] = OvfEntry{t00 + (uint32_t)((s4 >> 1) * L.gx + (s4 & 1)), r, d.key};
              if (r >= kTileCountLimit) ((SavedHeader*)(p_saved[lv] + L.o_hdr))->count_saturated = 1u;    // (before the field can wrap)
              ++pos;
            }
          sp &= ~(15u << (4 * jj));
        }
        m = __ballot(sp != 0u);
      }
    };
    unsigned long long old4[4] = {0ull, 0ull, 0ull, 0ull};
    uint32_t geo4[4] = {0u, 0u, 0u, 0u};
    int owner4[4];                                               // (-1: no operation)
    auto issue_round = [&](uint32_t it0, uint32_t tot_r) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const uint32_t item = it0 + (uint32_t)jj * 256u + (uint32_t)tid;
        owner4[jj] = -1;
        if (item < tot_r) {
          int lo = 0, hi = 256;                      // owner = last thread whose prefix is <= item
          while (lo < hi) { const int mid = (lo + hi) >> 1; if (op_ex[mid] <= item) lo = mid + 1; else hi = mid; }
          owner4[jj] = lo - 1;
          const uint4 r = op_rect[lo - 1];
          const float4 f0 = op_foot[lo - 1][0], f1 = op_foot[lo - 1][1];
          geo4[jj] = op_geom(r.x, r.y, 4 + (int)(item - op_ex[lo - 1]), Footprint{f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w});
          old4[jj] = issue((int)r.z, geo4[jj]);
        }
      }
    };
    auto consume_round = [&]() { consume4(old4, geo4, owner4); };
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      owner4[jj] = jj < (int)nops ? tid : -1;
      if (jj < (int)nops) { geo4[jj] = op_geom(my_rx, my_ry, jj, my_foot); old4[jj] = issue(v, geo4[jj]); }
    }
    uint32_t tot_t, tot_v;
    const uint32_t ex_t = carry_t + block256_exclusive_scan(cnt, red, tot_t);
    const uint32_t ex_v = carry_v + block256_exclusive_scan(vis, red, tot_v);
    // the running sums just after the last pair of a view are the next view's bases (empty views inherit them)
    if (live) {
#pragma unroll
      for (int u = 1; u <= kMaxViews; ++u)
        if (u > v && vstart[u] == f + 1) { vbase_t[u] = ex_t + cnt; vbase_v[u] = ex_v + vis; }
    }
    __syncthreads();
    if (o.visible) {
      const uint32_t k = ex_v - vbase_v[v];
      // touched, in-segment prefix (abs_offset() adds the segment base), list slot (relative; scatter_kernel makes it
      // absolute), SH clamp bits
      ((uint4*)((GRec*)(saved + L.o_grec) + i))[3] = make_uint4(cnt, ex_t - vbase_t[v], k, o.clamped);
      ((uint32_t*)(saved + L.o_seg_list))[seg0 + k] = (uint32_t)i;
    }
    consume_round();
    const uint32_t rem = nops > 4u ? nops - 4u : 0u;
    if (__syncthreads_or(rem > 0u)) {
      uint32_t tot_r;
      op_ex[tid] = block256_exclusive_scan(rem, red, tot_r);
      __syncthreads();
#pragma unroll 1
      for (uint32_t it0 = 0; it0 < tot_r; it0 += 4u * 256u) {
        issue_round(it0, tot_r);
        consume_round();
      }
    }
    __syncthreads();                                 // (the next chunk rewrites the owner tables)
    carry_t += tot_t;
    carry_v += tot_v;
  }
  __syncthreads();
  // which views of the batch see Gaussian ia (one word instead of one radii word per view for the optimiser pass)
  if (ia < N) ((uint32_t*)(p_saved[0] + L.o_vismask))[(size_t)part * N + ia] = vis_bits[tid];
  if (tid < nviews && (tid % nparts) == part) {
    char* saved = p_saved[tid];
    ((uint32_t*)(saved + L.o_block_touched))[seg] = vbase_t[tid + 1] - vbase_t[tid];
    ((uint32_t*)(saved + L.o_block_vis))[seg] = vbase_v[tid + 1] - vbase_v[tid];
  }
}

// ---------------------------------------------------------------------------------------------- backward
// Per-entry partials written by blend_bwd (12 floats = 48 B per (tile, Gaussian) pair), RAW sums over the tile's pixels:
//   0,1 : sum G dL/dG (dx, dy)   2,3 : the same times (dx dx, dx dy)   4 : dL/dopacity   5,6,7 : dL/drgb
//   8 : sum G dL/dG dy dy   9 : dL/ddepth   10,11 : unused
// (the dense backward below turns the per-Gaussian totals of 0..4 into dL/dmean2D and dL/dconic)
struct GaussGrad {
  float p[3], s[3], q[4], S6[6], op, m2[2], rgb_or_sh0[3];
};

// chain rule of ONE view for Gaussian i; adds into `acc`, returns the view's pose gradient in tau[6]; SH / precomputed
// colour gradients of degree > 0 are accumulated straight into the output arrays (dshs / dcolors)
__device__ __forceinline__ void preprocess_bwd_one_view(
    int i, int H, int W, int deg, int M, float tanfovx, float tanfovy, float mod, const float* __restrict__ viewmatrix,
    const float* __restrict__ projmatrix, const float* __restrict__ projraw, const float* __restrict__ campos,
    const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const float gsum[10] /* the Gaussian's per-tile partials, summed */, unsigned clamped_bits,
    float* __restrict__ dshs, int sh_accumulate, GaussGrad& acc, float tau[6], int upstream_pose_jac) {
  float g_m2[2] = {gsum[0], gsum[1]}, g_con[3] = {gsum[2], gsum[3], gsum[4]}, g_op = gsum[5], g_rgb[3] = {gsum[6], gsum[7], gsum[8]},
        g_dep = gsum[9];
  float g_p[3] = {0.f, 0.f, 0.f};
  float g_S6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float g_s[3] = {0.f, 0.f, 0.f}, g_q[4] = {0.f, 0.f, 0.f, 0.f};
  float sh0[3] = {0.f, 0.f, 0.f};
  const int accumulate = sh_accumulate;
  float* dcolors = nullptr;   // precomputed-colour gradient is returned through acc.rgb_or_sh0
  {
    float vm[16], pm[16], pr[16];
    load16(viewmatrix, vm);
    load16(projmatrix, pm);
    load16(projraw, pr);
    float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) pv[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
    float ph[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ph[r] = pm[r] * p[0] + pm[4 + r] * p[1] + pm[8 + r] * p[2] + pm[12 + r];
    float pw = 1.f / (ph[3] + 1e-7f);

    float s[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, S6[6];
    if (cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; ++k) S6[k] = cov3D_precomp[6 * i + k];
    } else {
      s[0] = scales[3 * i]; s[1] = scales[3 * i + 1]; s[2] = scales[3 * i + 2];
      q[0] = rotations[4 * i]; q[1] = rotations[4 * i + 1]; q[2] = rotations[4 * i + 2]; q[3] = rotations[4 * i + 3];
      cov3d_from_scale_rot(s, mod, q, S6);
    }
    float fx = W / (2.f * tanfovx), fy = H / (2.f * tanfovy);
    Ewa e;
    ewa_project(pv, vm, S6, fx, fy, 1.3f * tanfovx, 1.3f * tanfovy, e);

    // ---- conic -> 2D covariance
    float a = e.a, b = e.b, c = e.c;
    float det = a * c - b * b;
    float d2i = 1.f / (det * det + 1e-7f);   // upstream regulariser
    float dA = g_con[0], dB = g_con[1], dC = g_con[2];
    float dL_da = d2i * (-c * c * dA + b * c * dB - b * b * dC);
    float dL_db = d2i * (2.f * b * c * dA - (det + 2.f * b * b) * dB + 2.f * a * b * dC);
    float dL_dc = d2i * (-b * b * dA + a * b * dB - a * a * dC);
    float G2[2][2] = {{dL_da, 0.5f * dL_db}, {0.5f * dL_db, dL_dc}};

    // ---- dL/dSigma = T^T G T (full symmetric matrix), dL/dT = 2 G T Sigma
    float GT[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) GT[r][k] = G2[r][0] * e.T[0][k] + G2[r][1] * e.T[1][k];
    float dS[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int l = 0; l < 3; ++l) dS[k][l] = e.T[0][k] * GT[0][l] + e.T[1][k] * GT[1][l];
    float S[3][3] = {{S6[0], S6[1], S6[2]}, {S6[1], S6[3], S6[4]}, {S6[2], S6[4], S6[5]}};
    float dT[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) dT[r][k] = 2.f * (GT[r][0] * S[0][k] + GT[r][1] * S[1][k] + GT[r][2] * S[2][k]);

    // ---- T = J W:  dL/dJ = dT W^T,  dL/dW = J^T dT      (W[r][c] = vm[c*4+r])
    float itz = 1.f / e.tz, itz2 = itz * itz, itz3 = itz2 * itz;
    float J00 = fx * itz, J02 = -fx * e.tx * itz2, J11 = fy * itz, J12 = -fy * e.ty * itz2;
    float dJ00 = dT[0][0] * vm[0] + dT[0][1] * vm[4] + dT[0][2] * vm[8];
    float dJ02 = dT[0][0] * vm[2] + dT[0][1] * vm[6] + dT[0][2] * vm[10];
    float dJ11 = dT[1][0] * vm[1] + dT[1][1] * vm[5] + dT[1][2] * vm[9];
    float dJ12 = dT[1][0] * vm[2] + dT[1][1] * vm[6] + dT[1][2] * vm[10];
    float dW[3][3];   // dW[r][c]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dW[0][k] = J00 * dT[0][k];
      dW[1][k] = J11 * dT[1][k];
      dW[2][k] = J02 * dT[0][k] + J12 * dT[1][k];
    }
    // v = dL/d(p_view): covariance path (clamped axes carry no gradient), ...
    float v[3];
    v[0] = e.clamp_x ? 0.f : (-fx * itz2 * dJ02);
    v[1] = e.clamp_y ? 0.f : (-fy * itz2 * dJ12);
    v[2] = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + 2.f * fx * e.tx * itz3 * dJ02 + 2.f * fy * e.ty * itz3 * dJ12;
    // ... projected-mean path through the raw projection P (P[r][c] = pr[c*4+r]), ...
    float dph[4] = {g_m2[0] * pw, g_m2[1] * pw, 0.f, -(g_m2[0] * ph[0] + g_m2[1] * ph[1]) * pw * pw};
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] += dph[0] * pr[k * 4 + 0] + dph[1] * pr[k * 4 + 1] + dph[3] * pr[k * 4 + 3];
    // ... and the depth path.
    v[2] += g_dep;

    // world-space mean: W^T v
#pragma unroll
    for (int k = 0; k < 3; ++k) g_p[k] = vm[k * 4 + 0] * v[0] + vm[k * 4 + 1] * v[1] + vm[k * 4 + 2] * v[2];

    // camera pose, left perturbation W2C <- exp(tau) W2C, tau = (rho, theta)  (pose_utils.py:66-98).
    // SGR_OPT_UPSTREAM_POSE_JACOBIAN: the pose path sees the projected mean without the principal-point terms P02, P12
    // (P[r][c] = pr[c*4+r]); the world-space mean gradient above keeps the exact projection either way.
    float vz = v[2];
    if (upstream_pose_jac) vz -= dph[0] * pr[2 * 4 + 0] + dph[1] * pr[2 * 4 + 1];
    tau[0] = v[0]; tau[1] = v[1]; tau[2] = vz;
    V3 pc = cross(V3{pv[0], pv[1], pv[2]}, V3{v[0], v[1], vz});
    tau[3] = pc.x; tau[4] = pc.y; tau[5] = pc.z;
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // columns of W
      V3 wc = cross(V3{vm[k * 4 + 0], vm[k * 4 + 1], vm[k * 4 + 2]}, V3{dW[0][k], dW[1][k], dW[2][k]});
      tau[3] += wc.x; tau[4] += wc.y; tau[5] += wc.z;
    }

    // ---- colour
    if (colors_precomp) {
      sh0[0] = g_rgb[0]; sh0[1] = g_rgb[1]; sh0[2] = g_rgb[2];
    } else {
      unsigned cl = clamped_bits;
      float dc[3] = {(cl & 1u) ? 0.f : g_rgb[0], (cl & 2u) ? 0.f : g_rgb[1], (cl & 4u) ? 0.f : g_rgb[2]};
      float* out = dshs ? dshs + (size_t)i * M * 3 : nullptr;
      if (deg == 0) {
        sh0[0] = SH_C0 * dc[0]; sh0[1] = SH_C0 * dc[1]; sh0[2] = SH_C0 * dc[2];
        (void)out;
      } else {
        const float* sh = shs + (size_t)i * M * 3;
        float dir0[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
        float len2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
        float inv = 1.f / sqrtf(len2);
        float x = dir0[0] * inv, y = dir0[1] * inv, z = dir0[2] * inv;
        float basis[16];
        float dbx[16], dby[16], dbz[16];   // d(basis)/d(dir)
        for (int k = 0; k < 16; ++k) { basis[k] = 0.f; dbx[k] = 0.f; dby[k] = 0.f; dbz[k] = 0.f; }
        basis[0] = SH_C0;
        basis[1] = -SH_C1 * y; dby[1] = -SH_C1;
        basis[2] = SH_C1 * z;  dbz[2] = SH_C1;
        basis[3] = -SH_C1 * x; dbx[3] = -SH_C1;
        if (deg > 1) {
          float xx = x * x, yy = y * y, zz = z * z;
          basis[4] = SH_C2[0] * x * y; dbx[4] = SH_C2[0] * y; dby[4] = SH_C2[0] * x;
          basis[5] = SH_C2[1] * y * z; dby[5] = SH_C2[1] * z; dbz[5] = SH_C2[1] * y;
          basis[6] = SH_C2[2] * (2.f * zz - xx - yy); dbx[6] = SH_C2[2] * -2.f * x; dby[6] = SH_C2[2] * -2.f * y; dbz[6] = SH_C2[2] * 4.f * z;
          basis[7] = SH_C2[3] * x * z; dbx[7] = SH_C2[3] * z; dbz[7] = SH_C2[3] * x;
          basis[8] = SH_C2[4] * (xx - yy); dbx[8] = SH_C2[4] * 2.f * x; dby[8] = SH_C2[4] * -2.f * y;
          if (deg > 2) {
            basis[9] = SH_C3[0] * y * (3.f * xx - yy); dbx[9] = SH_C3[0] * 6.f * x * y; dby[9] = SH_C3[0] * (3.f * xx - 3.f * yy);
            basis[10] = SH_C3[1] * x * y * z; dbx[10] = SH_C3[1] * y * z; dby[10] = SH_C3[1] * x * z; dbz[10] = SH_C3[1] * x * y;
            basis[11] = SH_C3[2] * y * (4.f * zz - xx - yy); dbx[11] = SH_C3[2] * -2.f * x * y; dby[11] = SH_C3[2] * (4.f * zz - xx - 3.f * yy); dbz[11] = SH_C3[2] * 8.f * y * z;
            basis[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); dbx[12] = SH_C3[3] * -6.f * x * z; dby[12] = SH_C3[3] * -6.f * y * z; dbz[12] = SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
            basis[13] = SH_C3[4] * x * (4.f * zz - xx - yy); dbx[13] = SH_C3[4] * (4.f * zz - 3.f * xx - yy); dby[13] = SH_C3[4] * -2.f * x * y; dbz[13] = SH_C3[4] * 8.f * x * z;
            basis[14] = SH_C3[5] * z * (xx - yy); dbx[14] = SH_C3[5] * 2.f * x * z; dby[14] = SH_C3[5] * -2.f * y * z; dbz[14] = SH_C3[5] * (xx - yy);
            basis[15] = SH_C3[6] * x * (xx - 3.f * yy); dbx[15] = SH_C3[6] * (3.f * xx - 3.f * yy); dby[15] = SH_C3[6] * -6.f * x * y;
          }
        }
        int K = (deg + 1) * (deg + 1);
        float ddir[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < M; ++k) {
          float bk = k < K ? basis[k] : 0.f;
          for (int ch = 0; ch < 3; ++ch) {
            if (out) { if (accumulate) out[k * 3 + ch] += bk * dc[ch]; else out[k * 3 + ch] = bk * dc[ch]; }
            if (k < K) {
              float coef = sh[k * 3 + ch] * dc[ch];
              ddir[0] += dbx[k] * coef; ddir[1] += dby[k] * coef; ddir[2] += dbz[k] * coef;
            }
          }
        }
        // d(normalised dir)/d(mean): (I - d d^T) / |dir|
        float dot = x * ddir[0] + y * ddir[1] + z * ddir[2];
        g_p[0] += (ddir[0] - x * dot) * inv;
        g_p[1] += (ddir[1] - y * dot) * inv;
        g_p[2] += (ddir[2] - z * dot) * inv;
      }
    }

    // ---- 3D covariance
    if (cov3D_precomp) {
      g_S6[0] = dS[0][0]; g_S6[1] = 2.f * dS[0][1]; g_S6[2] = 2.f * dS[0][2];
      g_S6[3] = dS[1][1]; g_S6[4] = 2.f * dS[1][2]; g_S6[5] = dS[2][2];
    } else {
      float R[3][3];
      quat_to_R(q, R);
      // Sigma = M M^T, M = R diag(mod*s):  dL/dM = 2 dS M
      float Mx[3][3], dM[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) Mx[r][k] = R[r][k] * (mod * s[k]);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) dM[r][k] = 2.f * (dS[r][0] * Mx[0][k] + dS[r][1] * Mx[1][k] + dS[r][2] * Mx[2][k]);
      float dR[3][3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        g_s[k] = mod * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
#pragma unroll
        for (int r = 0; r < 3; ++r) dR[r][k] = dM[r][k] * (mod * s[k]);
      }
      float r = q[0], x = q[1], y = q[2], z = q[3];
      g_q[0] = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
      g_q[1] = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
      g_q[2] = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
      g_q[3] = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
    }
  }
  (void)dcolors;
#pragma unroll
  for (int k = 0; k < 3; ++k) { acc.p[k] += g_p[k]; acc.s[k] += g_s[k]; acc.rgb_or_sh0[k] += sh0[k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) acc.q[k] += g_q[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc.S6[k] += g_S6[k];
  acc.op += g_op;
  acc.m2[0] = g_m2[0]; acc.m2[1] = g_m2[1];     // per view (densification statistics use the per-view norm)
}

// Phase 1 (dense): thread = entry of a view's compact visible list; grid = (ceil(N/256), views), blocks beyond the
// list exit at once.  Writes one 64-byte gradient record per (view, visible Gaussian) and the view's pose partials.
// With a single view the rarely used extras (SH degree > 0, precomputed colour / covariance) go straight to the outputs.
// LPG = lanes per Gaussian.  1: thread = list entry (a fresh map: runs of a few slots).  4: a QUAD of lanes = list entry (dense maps, round 6):
// the quad sums its Gaussian's run together -- lane q takes slots q, q + 4, ... --, sixteen Gaussians per wave instead of 64, so the
// chain of dependent round trips a wave goes through is a quarter as long and four times as many waves hide it; only the quad's
// first lane goes on to the projection backward.
template <int LPG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) preprocess_bwd_dense_kernel(
    ViewTab tab, LOff L, Common cm, const float* __restrict__ means3D, const float* __restrict__ shs,
    const float* __restrict__ colors_precomp, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, float* __restrict__ dshs, float* __restrict__ dcov3D, int accumulate) {
  const int v = blockIdx.y;
  const char* saved = tab.saved[v];
  const int V = (int)((const SavedHeader*)(saved + L.o_hdr))->num_visible;
  constexpr int kPerBlock = 256 / LPG;
  if ((int)(blockIdx.x * kPerBlock) >= V) return;             // whole block beyond the list (uniform)
  const int t = blockIdx.x * kPerBlock + (int)threadIdx.x / LPG;
  const bool have = t < V;
  const int i = have ? (int)((const uint32_t*)(saved + L.o_vis_list))[t] : 0;
  uint4 q3 = make_uint4(0u, 0u, 0u, 0u);
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
  if (have) {
    const float4* rec = (const float4*)(grec_of(saved, L) + i);      // one 64-byte line
    g0 = rec[0]; g1 = rec[1]; q3 = ((const uint4*)rec)[3];
  }
  // bins of the rectangle that K1's exact footprint test dropped were never composited: nobody wrote their partial slot.  The
  // same test (a little more eager, sgr_common.h) says which slots to leave out of the sum.
  const uint32_t my_r01 = __float_as_uint(g0.z), my_r23 = __float_as_uint(g0.w);
  const bool tested = have && footprint_worthwhile((int)(my_r23 & 0xffffu) - (int)(my_r01 & 0xffffu), (int)(my_r23 >> 16) - (int)(my_r01 >> 16));
  const Footprint foot = tested ? make_footprint(g0.x, g0.y, g1.x, g1.y, g1.z, g1.w, 0.006f)
                                : Footprint{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, -1.f};
  // slot k of a run <-> bin (x0 + k % w, y0 + k / w) of the rectangle (r01 = x0 | y0 << 16, r23 = x1 | y1 << 16)
  auto slot_unwritten = [](const Footprint& f, uint32_t r01, uint32_t r23, uint32_t k) -> bool {
    if (!(f.thr >= 0.f)) return false;
    const int x0 = (int)(r01 & 0xffffu), y0 = (int)(r01 >> 16), w = (int)(r23 & 0xffffu) - x0;
    int q = (int)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));       // k / w, exact after one correction (k < 2^20)
    int rem = (int)k - q * w;
    if (rem < 0) { --q; rem += w; }
    if (rem >= w) { ++q; rem -= w; }
    return footprint_qmin(f, x0 + rem, y0 + q) > f.thr;
  };
  // Sum of this Gaussian's per-tile partials (48 B per covered tile, one contiguous run per Gaussian).  A converged map has
  // splats that cover tens to hundreds of tiles: a lane walking its own run alone makes the wave wait for its longest run
  // and fetches 48 scattered bytes per lane and step.  Instead every group of 8 lanes sums ONE run together (lane l takes
  // slots l, l + 8, ...: 384 contiguous bytes per step), eight rounds cover the wave's 64 Gaussians; fixed order, no atomics.
  constexpr uint32_t kLongRun = 96u;
  float gsum[10];
  {
    const float4* __restrict__ partials = (const float4*)(tab.scratch[v] + L.o_partials);
    const int lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    const uint32_t my_off = have ? abs_offset(saved, L, (uint32_t)i, q3.y) : 0u, my_cnt = have ? q3.x : 0u;
#pragma unroll
    for (int j = 0; j < 10; ++j) gsum[j] = 0.f;
    const int longest = __builtin_amdgcn_readfirstlane(wave_max_i32((int)my_cnt));
    constexpr uint32_t kLongRunQuad = 192u;       // (a quad takes 12 slots per batch: beyond 16 batches the whole wave is faster even one Gaussian at a time)
    if constexpr (LPG == 4) {
      const uint32_t q = (uint32_t)lane & 3u;
      const uint32_t c = my_cnt > kLongRunQuad ? 0u : my_cnt;
      for (uint32_t k0 = q; k0 < c; k0 += 12u) {
        float4 ld[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const uint32_t k = k0 + 4u * (uint32_t)u;
          const uint64_t e = (uint64_t)my_off + k;
          const bool in = k < c && (int64_t)e < L.cap && !slot_unwritten(foot, my_r01, my_r23, k);
#pragma unroll
          for (int w = 0; w < 3; ++w) ld[3 * u + w] = in ? partials[e * 3 + w] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 p0 = ld[3 * u], p1 = ld[3 * u + 1], p2 = ld[3 * u + 2];
          gsum[0] += p0.x; gsum[1] += p0.y; gsum[2] += p0.z; gsum[3] += p0.w; gsum[4] += p1.x;
          gsum[5] += p1.y; gsum[6] += p1.z; gsum[7] += p1.w; gsum[8] += p2.x; gsum[9] += p2.y;
        }
      }
#pragma unroll
      for (int off = 1; off < 4; off <<= 1)
#pragma unroll
        for (int j = 0; j < 10; ++j) gsum[j] += __shfl_xor(gsum[j], off);
    } else if (longest <= 6) {              // a fresh map (splats of a few tiles): every lane sums its own short run
      for (uint32_t k0 = 0; k0 < my_cnt; k0 += 3u) {         // (same: three slots' loads in flight together)
        float4 ld[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const uint64_t e = (uint64_t)my_off + k0 + (uint32_t)u;
          const bool in = k0 + (uint32_t)u < my_cnt && (int64_t)e < L.cap && !slot_unwritten(foot, my_r01, my_r23, k0 + (uint32_t)u);
#pragma unroll
          for (int w = 0; w < 3; ++w) ld[3 * u + w] = in ? partials[e * 3 + w] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 p0 = ld[3 * u], p1 = ld[3 * u + 1], p2 = ld[3 * u + 2];
          gsum[0] += p0.x; gsum[1] += p0.y; gsum[2] += p0.z; gsum[3] += p0.w; gsum[4] += p1.x;
          gsum[5] += p1.y; gsum[6] += p1.z; gsum[7] += p1.w; gsum[8] += p2.x; gsum[9] += p2.y;
        }
      }
    } else
#pragma unroll 1
    for (int r = 0; r < (LPG == 1 ? 8 : 0); ++r) {
      const int src = r * 8 + grp;                    // the lane whose Gaussian this group sums in round r
      const uint32_t o = (uint32_t)__shfl((int)my_off, src), c_all = (uint32_t)__shfl((int)my_cnt, src);
      const uint32_t c = c_all > kLongRun ? 0u : c_all;           // (long runs: the whole wave, below)
      const uint32_t s01 = (uint32_t)__shfl((int)my_r01, src), s23 = (uint32_t)__shfl((int)my_r23, src);
      const Footprint sf{__shfl(foot.px, src), __shfl(foot.py, src), __shfl(foot.A, src), __shfl(foot.B, src),
                         __shfl(foot.C, src), __shfl(foot.invA, src), __shfl(foot.invC, src), __shfl(foot.thr, src)};
      float part[10];
#pragma unroll
      for (int j = 0; j < 10; ++j) part[j] = 0.f;
      // three steps (24 slots) per batch, all nine loads issued before the first use: the kernel is bound by the latency of
      // these dependent round trips (8 rounds x steps), not by bytes (0.148 -> 0.10 ms on the opaque bench scene)
      for (uint32_t k0 = (uint32_t)sub; k0 < c; k0 += 24u) {
        float4 ld[9];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const uint32_t k = k0 + 8u * (uint32_t)u;
          const uint64_t e = (uint64_t)o + k;
          const bool in = k < c && (int64_t)e < L.cap && !slot_unwritten(sf, s01, s23, k);
#pragma unroll
          for (int w = 0; w < 3; ++w) ld[3 * u + w] = in ? partials[e * 3 + w] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const float4 p0 = ld[3 * u], p1 = ld[3 * u + 1], p2 = ld[3 * u + 2];
          part[0] += p0.x; part[1] += p0.y; part[2] += p0.z; part[3] += p0.w; part[4] += p1.x;
          part[5] += p1.y; part[6] += p1.z; part[7] += p1.w; part[8] += p2.x; part[9] += p2.y;
        }
      }
#pragma unroll
      for (int off = 1; off < 8; off <<= 1)
#pragma unroll
        for (int j = 0; j < 10; ++j) part[j] += __shfl_xor(part[j], off);
      // every lane of group g now holds the sum of lane (r * 8 + g)'s Gaussian: lane L of row r fetches its own from group L & 7
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const float got = __shfl(part[j], sub * 8);
        if (grp == r) gsum[j] = got;
      }
    }
    // A run of more than kLongRun slots -- a young SLAM map has splats that cover a quarter of the image, thousands of bins --
    // would keep ONE 8-lane group busy for c / 24 dependent round trips while the rest of the wave waits (the dense backward
    // was the most expensive kernel of a session's first 40 keyframes: 0.29 ms per launch).  Those runs are summed by the
    // whole wave, one Gaussian at a time: 192 slots per batch.
    constexpr uint32_t kWholeWave = LPG == 4 ? kLongRunQuad : kLongRun;
    if (longest > (int)kWholeWave) {
      unsigned long long lm = __ballot(my_cnt > kWholeWave && (LPG == 1 || (lane & (LPG - 1)) == 0));
      while (lm != 0ull) {
        const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)lm) - 1);
        lm &= lm - 1ull;
        const uint32_t o = (uint32_t)__shfl((int)my_off, src), c = (uint32_t)__shfl((int)my_cnt, src);
        const uint32_t s01 = (uint32_t)__shfl((int)my_r01, src), s23 = (uint32_t)__shfl((int)my_r23, src);
        const Footprint sf{__shfl(foot.px, src), __shfl(foot.py, src), __shfl(foot.A, src), __shfl(foot.B, src),
                           __shfl(foot.C, src), __shfl(foot.invA, src), __shfl(foot.invC, src), __shfl(foot.thr, src)};
        float part[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) part[j] = 0.f;
        for (uint32_t k0 = (uint32_t)lane; k0 < c; k0 += 192u) {
          float4 ld[9];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const uint32_t k = k0 + 64u * (uint32_t)u;
            const uint64_t e = (uint64_t)o + k;
            const bool in = k < c && (int64_t)e < L.cap && !slot_unwritten(sf, s01, s23, k);
#pragma unroll
            for (int w = 0; w < 3; ++w) ld[3 * u + w] = in ? partials[e * 3 + w] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const float4 p0 = ld[3 * u], p1 = ld[3 * u + 1], p2 = ld[3 * u + 2];
            part[0] += p0.x; part[1] += p0.y; part[2] += p0.z; part[3] += p0.w; part[4] += p1.x;
            part[5] += p1.y; part[6] += p1.z; part[7] += p1.w; part[8] += p2.x; part[9] += p2.y;
          }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
#pragma unroll
          for (int j = 0; j < 10; ++j) part[j] += __shfl_xor(part[j], off);
        if ((lane & ~(LPG - 1)) == src)      // (the Gaussian's lane; quad mode: its four lanes)
#pragma unroll
          for (int j = 0; j < 10; ++j) gsum[j] = part[j];
      }
    }
  }
  if (!have || (LPG > 1 && (threadIdx.x & (LPG - 1)) != 0)) return;
  {
    // the slots hold RAW sums (sgr_blend.hip: gx gy gxx gxy | o r g b | gyy d - -); this Gaussian's conic (A, B, C) and the
    // half-image factors turn their totals into dL/dmean2D (NDC-scaled pixel units) and the true partials of the conic
    // (summed in slot order -- gsum[0..9] = gx gy gxx gxy o r g b gyy d -- and put into the order the projection backward takes)
    const float G0 = gsum[0], G1 = gsum[1], Gxx = gsum[2], Gxy = gsum[3], Go = gsum[4], Gr = gsum[5], Gg = gsum[6], Gb = gsum[7],
                Gyy = gsum[8], Gd = gsum[9];
    gsum[0] = (-(g1.x * G0) - g1.y * G1) * (0.5f * (float)L.W);
    gsum[1] = (-(g1.z * G1) - g1.y * G0) * (0.5f * (float)L.H);
    gsum[2] = -0.5f * Gxx;
    gsum[3] = -Gxy;
    gsum[4] = -0.5f * Gyy;
    gsum[5] = Go; gsum[6] = Gr; gsum[7] = Gg; gsum[8] = Gb; gsum[9] = Gd;
  }
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  GaussGrad acc;
#pragma unroll
  for (int j = 0; j < 3; ++j) { acc.p[j] = 0.f; acc.s[j] = 0.f; acc.rgb_or_sh0[j] = 0.f; }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc.q[j] = 0.f;
#pragma unroll
  for (int j = 0; j < 6; ++j) acc.S6[j] = 0.f;
  acc.op = 0.f; acc.m2[0] = 0.f; acc.m2[1] = 0.f;
  preprocess_bwd_one_view(i, L.H, L.W, cm.deg, cm.M, cm.tanfovx, cm.tanfovy, cm.mod, tab.viewmatrix[v], tab.projmatrix[v],
                          tab.projraw[v], tab.campos[v], means3D, shs, colors_precomp, scales, rotations, cov3D_precomp,
                          gsum, q3.w, dshs, accumulate, acc, tau, cm.upstream_pose_jac);
  // the record sits at the GAUSSIAN's index (one full 64-byte sector): the gather pass then needs no list-slot lookup
  float4* rec = (float4*)(tab.scratch[v] + L.o_gradrec) + (size_t)i * 4;
  rec[0] = make_float4(acc.p[0], acc.p[1], acc.p[2], acc.rgb_or_sh0[0]);
  rec[1] = make_float4(acc.rgb_or_sh0[1], acc.rgb_or_sh0[2], acc.op, acc.s[0]);
  rec[2] = make_float4(acc.s[1], acc.s[2], acc.q[0], acc.q[1]);
  rec[3] = make_float4(acc.q[2], acc.q[3], acc.m2[0], acc.m2[1]);
  if (float* m2 = tab.dL_dmeans2D[v]) {      // this view's own screen-space gradient (the drop-in API's `means2D.grad`): the caller
    if (((const SavedHeader*)(saved + L.o_hdr))->overflow == 0u) {        // zeroed the buffer; a truncated view contributes nothing
      m2[3 * (size_t)i] = acc.m2[0];
      m2[3 * (size_t)i + 1] = acc.m2[1];
    }
  }
  if (dcov3D) {
#pragma unroll
    for (int j = 0; j < 6; ++j) { if (accumulate) dcov3D[6 * i + j] += acc.S6[j]; else dcov3D[6 * i + j] = acc.S6[j]; }
  }
  if (tab.dL_dtau[v]) {              // pose gradient requested: keep this Gaussian's 6 terms for the ordered reduction
    float* tr = (float*)(tab.scratch[v] + L.o_taurec) + (size_t)t * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) tr[j] = tau[j];
  }
}

// Phase 2 (light): thread = Gaussian; adds up its records over the views of the batch in fixed order (deterministic, no
// atomics), then writes (single view, every element defined) or accumulates (only Gaussians some view saw).
__global__ void __launch_bounds__(256) grad_gather_kernel(
    ViewTab tab, int nviews, LOff L, int deg, int M, int has_colors, float* __restrict__ dmeans3D,
    float* __restrict__ dmeans2D, float* __restrict__ dopac, float* __restrict__ dshs, float* __restrict__ dcolors,
    float* __restrict__ dscales, float* __restrict__ drots, float* __restrict__ dcov3D, int accumulate,
    float* __restrict__ stat_accum, float* __restrict__ stat_denom, float* __restrict__ stat_maxr) {
  __shared__ float red[4][6];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < L.N;
  float a[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) a[k] = 0.f;
  float m2x = 0.f, m2y = 0.f, st_norm = 0.f, st_cnt = 0.f, st_maxr = 0.f;
  bool any = false;
  for (int v = 0; v < nviews; ++v) {
    // (a view whose forward overflowed the pair capacity has partial slots nobody wrote: it contributes zeros; the caller sees
    //  header.overflow -- sgr_query / the drop-in's deferred check -- and redoes the work with a larger workspace)
    const bool truncated = ((const SavedHeader*)tab.saved[v])->overflow != 0u;
    const int r = in_range && !truncated ? tab.radii[v][i] : 0;
    uint32_t pos = 0;
    if (r > 0) {
      any = true;
      if (tab.dL_dtau[v]) pos = grec_of(tab.saved[v], L)[i].vis_pos;       // (only the pose terms are stored by list slot)
      const float4* rec = (const float4*)(tab.scratch[v] + L.o_gradrec) + (size_t)i * 4;
      float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
      a[0] += r0.x; a[1] += r0.y; a[2] += r0.z; a[3] += r0.w; a[4] += r1.x; a[5] += r1.y; a[6] += r1.z; a[7] += r1.w;
      a[8] += r2.x; a[9] += r2.y; a[10] += r2.z; a[11] += r2.w; a[12] += r3.x; a[13] += r3.y;
      m2x += r3.z; m2y += r3.w;
      st_norm += sqrtf(r3.z * r3.z + r3.w * r3.w);
      st_cnt += 1.f;
      st_maxr = fmaxf(st_maxr, (float)r);
    }
    if (tab.dL_dtau[v]) {      // uniform: ordered (by Gaussian index) reduction of this view's pose gradient
      const float* tr = (const float*)(tab.scratch[v] + L.o_taurec) + (size_t)pos * 6;
      int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        float s = wave_sum(r > 0 ? tr[k] : 0.f);
        if (lane == 0) red[wv][k] = s;
      }
      __syncthreads();
      if (threadIdx.x < 6) {
        int k = threadIdx.x;
        ((float*)(tab.scratch[v] + L.o_tau_part))[(size_t)blockIdx.x * 6 + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
      }
      __syncthreads();
    }
  }
  if (!in_range) return;
  float* sh0 = has_colors ? dcolors : dshs;
  const size_t sh_stride = has_colors ? 3 : (size_t)M * 3;
  if (accumulate) {
    if (!any) return;
    if (dmeans3D) { dmeans3D[3 * i] += a[0]; dmeans3D[3 * i + 1] += a[1]; dmeans3D[3 * i + 2] += a[2]; }
    if (sh0 && (has_colors || deg == 0)) { sh0[i * sh_stride] += a[3]; sh0[i * sh_stride + 1] += a[4]; sh0[i * sh_stride + 2] += a[5]; }
    if (dopac) dopac[i] += a[6];
    if (dscales) { dscales[3 * i] += a[7]; dscales[3 * i + 1] += a[8]; dscales[3 * i + 2] += a[9]; }
    if (drots) { drots[4 * i] += a[10]; drots[4 * i + 1] += a[11]; drots[4 * i + 2] += a[12]; drots[4 * i + 3] += a[13]; }
    if (dmeans2D) { dmeans2D[3 * i] += m2x; dmeans2D[3 * i + 1] += m2y; }
  } else {
    if (dmeans3D) { dmeans3D[3 * i] = a[0]; dmeans3D[3 * i + 1] = a[1]; dmeans3D[3 * i + 2] = a[2]; }
    if (sh0 && (has_colors || deg == 0)) {
      sh0[i * sh_stride] = a[3]; sh0[i * sh_stride + 1] = a[4]; sh0[i * sh_stride + 2] = a[5];
      for (size_t k = 3; k < sh_stride; ++k) sh0[i * sh_stride + k] = 0.f;
    } else if (dshs && !any) {
      for (size_t k = 0; k < sh_stride; ++k) dshs[i * sh_stride + k] = 0.f;    // degree > 0: phase 1 wrote the visible rows
    }
    if (dopac) dopac[i] = a[6];
    if (dscales) { dscales[3 * i] = a[7]; dscales[3 * i + 1] = a[8]; dscales[3 * i + 2] = a[9]; }
    if (drots) { drots[4 * i] = a[10]; drots[4 * i + 1] = a[11]; drots[4 * i + 2] = a[12]; drots[4 * i + 3] = a[13]; }
    if (dmeans2D) { dmeans2D[3 * i] = m2x; dmeans2D[3 * i + 1] = m2y; dmeans2D[3 * i + 2] = 0.f; }
    if (dcov3D && !any) {
#pragma unroll
      for (int k = 0; k < 6; ++k) dcov3D[6 * i + k] = 0.f;
    }
  }
  if (any && stat_accum) {
    stat_accum[i] += st_norm;
    stat_denom[i] += st_cnt;
    stat_maxr[i] = fmaxf(stat_maxr[i], st_maxr);
  }
}

// second stage: one block per view sums the per-block partials in a fixed order
__global__ void __launch_bounds__(384) tau_reduce_kernel(ViewTab tab, LOff L) {
  __shared__ float red[6][64];
  const int v = blockIdx.x;
  float* dtau = tab.dL_dtau[v];
  if (!dtau) return;
  const float* tau_part = (const float*)(tab.scratch[v] + L.o_tau_part);
  const int nblk = L.pre_blocks;
  int k = threadIdx.x / 64, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int b = lane; b < nblk; b += 64) acc += tau_part[(size_t)b * 6 + k];
  red[k][lane] = acc;
  __syncthreads();
  if (lane == 0) {
    float t = 0.f;
    for (int l = 0; l < 64; ++l) t += red[k][l];
    dtau[k] = t;
  }
}

void launch_preprocess_fwd(const ViewTab& tab, int nviews, const LOff& L, const Common& cm, const SgrInputs& in, hipStream_t st) {
  if (L.N <= 0) return;
  ProfScope prof(PK_PRE_FWD, st);
  hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(L.nseg, min(L.k1_parts, nviews)), dim3(256), 0, st, tab, nviews, L, cm, in.means3D, in.opacities,
                     in.shs, in.colors_precomp, in.scales, in.rotations, in.cov3D_precomp);
}

void launch_preprocess_bwd(const ViewTab& tab, int nviews, const LOff& L, const Common& cm, const SgrInputs& in,
                           const SgrGradInputs& g, const FusedAdam* fused, hipStream_t st, bool mapping_loop) {
  if (L.N <= 0) return;
  ProfScope prof(PK_PRE_BWD, st);
  // Quad mode (four lanes per list entry) for the mapping loops' SINGLE-view iterations -- initialize_map's 1050 and final_refine's
  // thousands: one view's list is 300-500 waves at 64 entries per wave, less than half a wave per SIMD, and a young map's splats cover
  // hundreds to thousands of bins: the waves walk their long runs one Gaussian at a time.  A quarter of the entries per wave = four
  // times the waves and a quarter of the serial chain: the initialisation keyframe 358 -> 290 ms (same-box A/B, SGR_DENSE_QUAD=0/1);
  // a 12-view batch has waves enough and is 3 % slower in quad mode (light +4 us, opaque +4 us): it keeps thread = entry.  The rule
  // is a function of the CALL (sgr_map_views / sgr_map_step / sgr_map_run with one view -- with or without the fused optimiser tail, so
  // that the two stay bitwise equal --, never of measured history: two identical calls round alike); the drop-in's sgr_backward keeps
  // thread = entry (its batched and per-view backward are compared bit for bit).  SGR_DENSE_QUAD=0 / 1 forces a mode.
  static const int forced = []() { const char* e = getenv("SGR_DENSE_QUAD"); return e ? atoi(e) : -1; }();
  const bool quad = forced >= 0 ? forced != 0 : (mapping_loop && nviews == 1);
  if (quad)
    hipLaunchKernelGGL(preprocess_bwd_dense_kernel<4>, dim3(L.pre_blocks * 4, nviews), dim3(256), 0, st, tab, L, cm, in.means3D, in.shs,
                       in.colors_precomp, in.scales, in.rotations, in.cov3D_precomp, g.dL_dshs, g.dL_dcov3D_precomp, g.accumulate);
  else
    hipLaunchKernelGGL(preprocess_bwd_dense_kernel<1>, dim3(L.pre_blocks, nviews), dim3(256), 0, st, tab, L, cm, in.means3D, in.shs,
                       in.colors_precomp, in.scales, in.rotations, in.cov3D_precomp, g.dL_dshs, g.dL_dcov3D_precomp, g.accumulate);
  if (fused) {               // single-GPU mapping iteration: the gather rides in the optimiser pass (no gradient round trip)
    launch_gather_adam(tab, nviews, L, *fused, st);
    return;
  }
  hipLaunchKernelGGL(grad_gather_kernel, dim3(L.pre_blocks), dim3(256), 0, st, tab, nviews, L, cm.deg, cm.M,
                     in.colors_precomp != nullptr ? 1 : 0, g.dL_dmeans3D, g.dL_dmeans2D, g.dL_dopacities, g.dL_dshs,
                     g.dL_dcolors_precomp, g.dL_dscales, g.dL_drotations, g.dL_dcov3D_precomp, g.accumulate,
                     g.stat_grad_accum, (g.stat_grad_accum ? g.stat_denom : nullptr),
                     (g.stat_grad_accum ? g.stat_max_radii : nullptr));
  bool any_tau = false;
  for (int v = 0; v < nviews; ++v) any_tau = any_tau || tab.dL_dtau[v] != nullptr;
  if (any_tau) hipLaunchKernelGGL(tau_reduce_kernel, dim3(nviews), dim3(384), 0, st, tab, L);
}

}  // namespace sgr
