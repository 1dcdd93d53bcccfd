// Tile blending for gfx950: front-to-back alpha compositing (forward) and its gradient (backward).
//
// Replaces renderCUDA (forward/backward) of the un-vendored diff-gaussian-rasterization-w-pose module reached from
// /root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141 and from loss.backward() at
// /root/reference/src/mapper.py:329,490,699.
//
// MI355X design (not the CUDA 16x16-block/atomicAdd scheme):
//   * a bin is 8x8 pixels = exactly one wave64; a 256-thread workgroup is four independent waves covering one
//     16x16 reference tile.  Waves never synchronise with each other: no __syncthreads in either kernel.
//   * FORWARD is pixel-parallel (lane = pixel).  The wave stages 64 sorted splats at a time into its private LDS
//     slice (coalesced gather -> ds_write_b128), then walks them with broadcast ds_read_b128; per-pixel
//     accumulators stay in VGPRs.  n_touched is one ballot+popcount+atomic per (wave, splat), not per pixel.
//   * BACKWARD is splat-parallel (lane = splat, loop over the 64 pixels).  For one pixel the transmittance in front
//     of every splat is a wave-wide multiplicative DPP scan and the colour behind it an additive DPP scan of ONE
//     scalar (w_j = dL/dC . rgb_j + dL/dD * depth_j), so per (pixel, 64 splats) there are 2 scans instead of the
//     10 cross-lane reductions a pixel-parallel backward needs, and the 10 per-splat gradient sums accumulate in
//     that lane's registers.  Each (tile, splat) pair then writes its 48-byte partial to a slot owned by the
//     Gaussian: no atomics, bitwise run-to-run deterministic; preprocess_bwd gathers the slots in fixed order.
#include "sgr_common.h"

namespace sgr {

// workgroup -> 16x16 super tile with an XCD-aware remap: hardware places block b on XCD b%8, we hand every XCD a
// contiguous run of super tiles so neighbouring tiles (which share Gaussians) hit the same 4 MiB L2.
__device__ __forceinline__ int super_tile_of_block(int b, int nblocks) {
  int per = (nblocks + 7) >> 3;
  int t = (b & 7) * per + (b >> 3);
  return t;   // may be >= nblocks for the tail: caller checks
}

// ------------------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256) blend_fwd_kernel(
    int H, int W, int gx, int gy, int sgx, int sgy, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ xy, const float4* __restrict__ conic_o,
    const float4* __restrict__ rgbd, const float* __restrict__ bg, float* __restrict__ out_color,
    float* __restrict__ out_depth, float* __restrict__ out_opacity, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_maxc, int32_t* __restrict__ n_touched) {
  __shared__ float4 stage[4][kWave * 3];   // per wave: 64 splats x 48 B
  const int nblocks = sgx * sgy;
  const int st = super_tile_of_block(blockIdx.x, nblocks);
  if (st >= nblocks) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = (st % sgx) * 2 + (wv & 1), ty = (st / sgx) * 2 + (wv >> 1);
  if (tx >= gx || ty >= gy) return;          // whole wave outside the image
  const int tile = ty * gx + tx;
  const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  float4* lds = stage[wv];

  const uint2 rng = ranges[tile];
  const int count = (int)(rng.y - rng.x);

  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  for (int base = 0; base < count; base += kWave) {
    const int n = min(kWave, count - base);
    // gather this chunk: lane j fetches splat j (coalesced index read, then 40 B of geometry)
    if (lane < n) {
      uint32_t g = point_list[rng.x + base + lane];
      float2 m = xy[g];
      float4 co = conic_o[g];
      float4 cd = rgbd[g];
      lds[lane * 3 + 0] = make_float4(m.x, m.y, co.x, co.y);
      lds[lane * 3 + 1] = make_float4(co.z, co.w, cd.w, __uint_as_float(g));
      lds[lane * 3 + 2] = make_float4(cd.x, cd.y, cd.z, 0.f);
    }
    // single wave: LDS writes are visible to the same wave after the implicit lgkmcnt wait; no barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < n; ++j) {
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
      float4 e0 = lds[j * 3 + 0], e1 = lds[j * 3 + 1], e2 = lds[j * 3 + 2];
      AlphaEval a = eval_alpha(e0.x - pxf, e0.y - pyf, e0.z, e0.w, e1.x, e1.y);
      float testT = T * (1.f - a.alpha);
      bool live = !done && a.ok;
      bool term = live && (testT < kTEps);
      bool comp = live && !term;
      done = done || term;
      float w = comp ? a.alpha * T : 0.f;
      C0 = __fmaf_rn(e2.x, w, C0);
      C1 = __fmaf_rn(e2.y, w, C1);
      C2 = __fmaf_rn(e2.z, w, C2);
      D = __fmaf_rn(e1.z, w, D);
      unsigned long long tm = __builtin_amdgcn_ballot_w64(comp && testT > kTouchedT);
      if (tm != 0ull && lane == 0) atomicAdd(&n_touched[__float_as_uint(e1.w)], (int)__popcll(tm));
      if (comp) { T = testT; last = (uint32_t)(base + j + 1); }
    }
    __builtin_amdgcn_wave_barrier();
    if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
  }

  // per-tile bound for the backward: it never has to look past the last contributor of any pixel
  uint32_t mx = last;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  if (lane == 0) tile_maxc[tile] = mx;

  if (inside) {
    const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[hw + pix] = C1 + T * bg[1];
    out_color[2 * hw + pix] = C2 + T * bg[2];
    out_depth[pix] = D;
    out_opacity[pix] = 1.f - T;
  }
}

// ------------------------------------------------------------------------------------------------ backward
__global__ void __launch_bounds__(256) blend_bwd_kernel(
    int H, int W, int gx, int gy, int sgx, int sgy, const uint2* __restrict__ ranges,
    const uint32_t* __restrict__ point_list, const float2* __restrict__ xy, const float4* __restrict__ conic_o,
    const float4* __restrict__ rgbd, const ushort4* __restrict__ rect, const uint32_t* __restrict__ offsets,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const uint32_t* __restrict__ tile_maxc, const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    float4* __restrict__ partials, int64_t cap) {
  const int nblocks = sgx * sgy;
  const int st = super_tile_of_block(blockIdx.x, nblocks);
  if (st >= nblocks) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tx = (st % sgx) * 2 + (wv & 1), ty = (st / sgx) * 2 + (wv >> 1);
  if (tx >= gx || ty >= gy) return;
  const int tile = ty * gx + tx;
  const uint2 rng = ranges[tile];
  const int count = (int)(rng.y - rng.x);
  if (count == 0) return;
  const int eff = min(count, (int)tile_maxc[tile]);

  // pixel state: lane p owns pixel p of the tile; the pixel loop broadcasts it with v_readlane
  const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
  const bool inside = px < W && py < H;
  const size_t pix = (size_t)py * W + px, hw = (size_t)H * W;
  float dCr = 0.f, dCg = 0.f, dCb = 0.f, dD = 0.f, Tc = 1.f, Sc = 0.f;
  int ncont = 0;
  if (inside) {
    dCr = dL_dcolor[pix]; dCg = dL_dcolor[hw + pix]; dCb = dL_dcolor[2 * hw + pix];
    dD = dL_ddepth ? dL_ddepth[pix] : 0.f;
    Tc = final_T[pix];
    ncont = (int)n_contrib[pix];
    // the background term -T_final/(1-alpha_j) * (bg . dL/dC) has the same shape as "colour behind splat j"
    Sc = Tc * (bg[0] * dCr + bg[1] * dCg + bg[2] * dCb);
  }
  const float halfW = 0.5f * (float)W, halfH = 0.5f * (float)H;
  const float tx0 = (float)(tx * kTile), ty0 = (float)(ty * kTile);

  const int nchunks = (eff + kWave - 1) / kWave;
  for (int c = nchunks - 1; c >= 0; --c) {
    // lanes are mapped to splats in REVERSE list order so that "everything behind me" is a prefix scan over lanes
    const int idx = c * kWave + (kWave - 1 - lane);
    const bool valid = idx < eff;
    uint32_t g = 0;
    float mx = 0.f, my = 0.f, A = 0.f, B = 0.f, Cc = 0.f, op = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dep = 0.f;
    if (valid) {
      g = point_list[rng.x + idx];
      float2 m = xy[g];
      float4 co = conic_o[g];
      float4 cd = rgbd[g];
      mx = m.x; my = m.y; A = co.x; B = co.y; Cc = co.z; op = co.w; cr = cd.x; cg = cd.y; cb = cd.z; dep = cd.w;
    }
    float s_g = 0.f, s_gx = 0.f, s_gy = 0.f, s_gxx = 0.f, s_gxy = 0.f, s_gyy = 0.f;
    float a_o = 0.f, a_r = 0.f, a_g = 0.f, a_b = 0.f, a_d = 0.f;
    (void)s_g;

    for (int p = 0; p < kWave; ++p) {
      const int nc = __builtin_amdgcn_readlane(ncont, p);
      if (nc <= c * kWave) continue;                       // this pixel stopped before this chunk (uniform)
      const float pdr = readlane_f(dCr, p), pdg = readlane_f(dCg, p), pdb = readlane_f(dCb, p), pdd = readlane_f(dD, p);
      const float pT = readlane_f(Tc, p), pS = readlane_f(Sc, p);
      const float dx = mx - (tx0 + (float)(p & 7)), dy = my - (ty0 + (float)(p >> 3));
      AlphaEval a = eval_alpha(dx, dy, A, B, Cc, op);
      const bool ok = valid && (idx < nc) && a.ok;
      const float om = ok ? 1.f - a.alpha : 1.f;
      const float P = wave_scan_mul(om);                   // prod over this splat and all behind it (in chunk)
      const float E = wave_shr1(P, 1.f);                   // prod over all strictly behind it
      const float rP = __builtin_amdgcn_rcpf(P);
      const float Tj = pT * rP;                            // transmittance in front of splat j
      const float inv1ma = E * rP;                         // 1 / (1 - alpha_j)
      const float w = __fmaf_rn(pdr, cr, __fmaf_rn(pdg, cg, __fmaf_rn(pdb, cb, pdd * dep)));
      const float aT = ok ? a.alpha * Tj : 0.f;
      const float q = w * aT;
      const float Qi = wave_scan_add(q);                   // inclusive: this splat and all behind it
      const float Sx = (Qi - q) + pS;                      // strictly behind (+ carried chunks + background term)
      const float dL_dalpha = __fmaf_rn(Tj, w, -Sx * inv1ma);
      // carry to the next (nearer) chunk: lane 63 holds the nearest splat of this chunk
      const float nT = readlane_f(Tj, 63), nS = readlane_f(Qi, 63) + pS;
      if (lane == p) { Tc = nT; Sc = nS; }
      if (ok) {
        a_r = __fmaf_rn(aT, pdr, a_r);
        a_g = __fmaf_rn(aT, pdg, a_g);
        a_b = __fmaf_rn(aT, pdb, a_b);
        a_d = __fmaf_rn(aT, pdd, a_d);
        const float gd = a.G * dL_dalpha;                  // dL/dopacity contribution; alpha clamp is straight-through
        a_o += gd;
        const float gg = gd * op;                          // G * dL/dG
        const float gxv = gg * dx, gyv = gg * dy;
        s_gx += gxv; s_gy += gyv;
        s_gxx = __fmaf_rn(gxv, dx, s_gxx);
        s_gxy = __fmaf_rn(gxv, dy, s_gxy);
        s_gyy = __fmaf_rn(gyv, dy, s_gyy);
      }
    }

    if (valid) {
      // slot of this (tile, Gaussian) pair inside the Gaussian's own run of partials
      ushort4 r = rect[g];
      uint64_t slot = (uint64_t)offsets[g] + (uint32_t)((ty - (int)r.y) * ((int)r.z - (int)r.x) + (tx - (int)r.x));
      if ((int64_t)slot < cap) {
        float dmx = (-(A * s_gx) - B * s_gy) * halfW;
        float dmy = (-(Cc * s_gy) - B * s_gx) * halfH;
        partials[slot * 3 + 0] = make_float4(dmx, dmy, -0.5f * s_gxx, -s_gxy);
        partials[slot * 3 + 1] = make_float4(-0.5f * s_gyy, a_o, a_r, a_g);
        partials[slot * 3 + 2] = make_float4(a_b, a_d, 0.f, 0.f);
      }
    }
  }
}

// (tile, Gaussian) pairs that the forward never reached (beyond every pixel's last contributor, or in tiles whose
// pixels all terminated early) still own a slot: zero them so the gather in preprocess_bwd reads defined data.
__global__ void __launch_bounds__(256) zero_partials_kernel(float4* __restrict__ partials, const SavedHeader* __restrict__ hdr) {
  size_t n = (size_t)hdr->sorted_count * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    partials[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_blend_fwd(const SgrSettings& s, const SgrOutputs& out, const Layout& L, char* saved, hipStream_t st) {
  int nblocks = L.sgx * L.sgy;
  int grid = ((nblocks + 7) / 8) * 8;
  ProfScope prof(PK_BLEND_FWD, st);
  hipLaunchKernelGGL(blend_fwd_kernel, dim3(grid), dim3(256), 0, st, s.image_height, s.image_width, L.gx, L.gy, L.sgx,
                     L.sgy, (const uint2*)(saved + L.o_ranges), (const uint32_t*)(saved + L.o_point_list),
                     (const float2*)(saved + L.o_xy), (const float4*)(saved + L.o_conic_o),
                     (const float4*)(saved + L.o_rgbd), s.bg, out.color, out.depth, out.opacity,
                     (float*)(saved + L.o_final_T), (uint32_t*)(saved + L.o_n_contrib),
                     (uint32_t*)(saved + L.o_tile_maxc), out.n_touched);
}

void launch_blend_bwd(const SgrSettings& s, const SgrGradOutputs& go, const Layout& L, const char* saved, char* scratch,
                      hipStream_t st) {
  int nblocks = L.sgx * L.sgy;
  int grid = ((nblocks + 7) / 8) * 8;
  float4* partials = (float4*)(scratch + L.o_partials);
  {
    ProfScope prof(PK_ZERO, st);
    hipLaunchKernelGGL(zero_partials_kernel, dim3(1024), dim3(256), 0, st, partials, (const SavedHeader*)(saved + L.o_hdr));
  }
  ProfScope prof(PK_BLEND_BWD, st);
  hipLaunchKernelGGL(blend_bwd_kernel, dim3(grid), dim3(256), 0, st, s.image_height, s.image_width, L.gx, L.gy, L.sgx,
                     L.sgy, (const uint2*)(saved + L.o_ranges), (const uint32_t*)(saved + L.o_point_list),
                     (const float2*)(saved + L.o_xy), (const float4*)(saved + L.o_conic_o),
                     (const float4*)(saved + L.o_rgbd), (const ushort4*)(saved + L.o_rect),
                     (const uint32_t*)(saved + L.o_offsets), s.bg, (const float*)(saved + L.o_final_T),
                     (const uint32_t*)(saved + L.o_n_contrib), (const uint32_t*)(saved + L.o_tile_maxc), go.dL_dcolor,
                     go.dL_ddepth, partials, L.cap);
}

}  // namespace sgr
