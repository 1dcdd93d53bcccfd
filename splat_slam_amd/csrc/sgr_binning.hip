// Tile binning for gfx950 without a device-wide sort:  K2 tile_scan + K3 scatter  (K1 = preprocess_fwd counts pairs per
// tile, K4 = blend_fwd sorts each tile's short list inside its own wave).  Replaces the cub InclusiveSum +
// duplicateWithKeys + 64-bit DeviceRadixSort + identifyTileRanges chain of the un-vendored CUDA rasterizer
// (/root/reference/.gitmodules:4-6; call site thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141).
//
// Why: at SLAM map sizes the global sort is pure launch latency (6 radix passes x several kernels over ~60 k pairs).
// Per-tile lists are short (tens of splats), LDS is 160 KB/CU: sort them where they are consumed.  The order is
// identical to the reference's (tile, depth-bits) stable sort because keys (depth bits, Gaussian index) are unique.
// Everything is fixed-order (two-level scans, no floating-point atomics): results are bitwise reproducible.
#include "sgr_common.h"

namespace sgr {

// exclusive scan of in[0..n) by ONE 1024-thread block, 4 items per thread per pass; returns the total
__device__ uint32_t block1024_scan(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n, uint32_t* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;   // 16 waves
  uint32_t carry = 0;
  for (int base = 0; base < n; base += 4096) {
    int i0 = base + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k) < n ? in[i0 + k] : 0u;
    uint32_t s4 = v[0] + v[1] + v[2] + v[3];
    uint32_t inc = wave_scan_add_u32(s4);
    if (lane == 63) red[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { uint32_t r = red[w]; pre += (w < wv) ? r : 0u; tot += r; }
    __syncthreads();
    uint32_t run = carry + pre + inc - s4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if ((i0 + k) < n) out[i0 + k] = run; run += v[k]; }
    carry += tot;
  }
  return carry;
}

// K2: three independent scans, one 1024-thread block each: (0) tile starts, (1) block bases of the partial-slot
// offsets, (2) block bases of the visible list.  ranges[t] = (start, start): scatter uses .y as the fill cursor, so
// after K3 it is the end of the tile's run.
__global__ void __launch_bounds__(1024) tile_scan_kernel(int ntiles, int nblocks, int64_t cap,
                                                         const uint32_t* __restrict__ tile_count, uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ block_touched, uint32_t* __restrict__ block_base_t,
                                                         const uint32_t* __restrict__ block_vis, uint32_t* __restrict__ block_base_v,
                                                         uint32_t* __restrict__ tmp, SavedHeader* __restrict__ hdr) {
  __shared__ uint32_t red[16];
  if (blockIdx.x == 0) {
    uint32_t R = block1024_scan(tile_count, tmp, ntiles, red);
    __syncthreads();
    for (int t = threadIdx.x; t < ntiles; t += 1024) { uint32_t s0 = tmp[t]; ranges[t] = make_uint2(s0, s0); }
    if (threadIdx.x == 0) {
      hdr->num_rendered = R;
      hdr->overflow = (int64_t)R > cap ? 1u : 0u;
      hdr->sorted_count = (uint32_t)((int64_t)R > cap ? cap : (int64_t)R);
    }
  } else if (blockIdx.x == 1) {
    (void)block1024_scan(block_touched, block_base_t, nblocks, red);
  } else {
    uint32_t V = block1024_scan(block_vis, block_base_v, nblocks, red);
    if (threadIdx.x == 0) hdr->num_visible = V;
  }
}

// K3: finishes the two-level scans (absolute partial-slot offsets, compact visible list) and scatters one
// (depth bits | Gaussian) key per pair into its tile's run.  Order inside a run is arbitrary here; K4 sorts it.
__global__ void __launch_bounds__(256) scatter_kernel(int N, int gx, int64_t cap, const int32_t* __restrict__ radii,
                                                      const uint32_t* __restrict__ touched, uint32_t* __restrict__ offsets,
                                                      const ushort4* __restrict__ rect, const float4* __restrict__ rgbd,
                                                      const uint32_t* __restrict__ block_base_t,
                                                      const uint32_t* __restrict__ block_base_v, uint32_t* __restrict__ vis_list,
                                                      uint2* __restrict__ ranges, uint64_t* __restrict__ entries) {
  __shared__ uint32_t red[4];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool vis = i < N && radii[i] > 0;
  uint32_t tot;
  uint32_t vpos = block256_exclusive_scan(vis ? 1u : 0u, red, tot);
  if (tot == 0) return;                                   // nothing visible in this block (uniform)
  if (i >= N) return;
  uint32_t off = offsets[i] + block_base_t[blockIdx.x];
  offsets[i] = off;
  if (!vis) return;
  vis_list[block_base_v[blockIdx.x] + vpos] = (uint32_t)i;
  if (touched[i] == 0) return;
  ushort4 r = rect[i];
  uint64_t key = ((uint64_t)__float_as_uint(rgbd[i].w) << 32) | (uint32_t)i;
  const int w = (int)r.z - (int)r.x, cnt = w * ((int)r.w - (int)r.y);
  // returning atomics are latency-bound: keep 4 in flight per thread (most splats cover <= 4 bins)
  for (int k0 = 0; k0 < cnt; k0 += 4) {
    uint32_t pos[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int kk = k0 + k;
      if (kk < cnt) pos[k] = atomicAdd(&ranges[((int)r.y + kk / w) * gx + (int)r.x + kk % w].y, 1u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k0 + k < cnt && (int64_t)pos[k] < cap) entries[pos[k]] = key;
  }
}

void launch_binning(const SgrSettings& s, const SgrOutputs& out, const Layout& L, char* saved, char* scratch, hipStream_t st) {
  const int N = s.num_gaussians;
  {
    ProfScope prof(PK_SCAN, st);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(3), dim3(1024), 0, st, L.ntiles, L.pre_blocks, L.cap,
                       (const uint32_t*)(saved + L.o_tile_count), (uint2*)(saved + L.o_ranges),
                       (const uint32_t*)(saved + L.o_block_touched), (uint32_t*)(saved + L.o_block_base_t),
                       (const uint32_t*)(saved + L.o_block_vis), (uint32_t*)(saved + L.o_block_base_v),
                       (uint32_t*)(saved + L.o_tile_maxc), (SavedHeader*)(saved + L.o_hdr));
  }
  if (N > 0) {
    ProfScope prof(PK_SCATTER, st);
    hipLaunchKernelGGL(scatter_kernel, dim3(L.pre_blocks), dim3(256), 0, st, N, L.gx, L.cap, out.radii,
                       (const uint32_t*)(saved + L.o_touched), (uint32_t*)(saved + L.o_offsets),
                       (const ushort4*)(saved + L.o_rect), (const float4*)(saved + L.o_rgbd),
                       (const uint32_t*)(saved + L.o_block_base_t), (const uint32_t*)(saved + L.o_block_base_v),
                       (uint32_t*)(saved + L.o_vis_list), (uint2*)(saved + L.o_ranges), (uint64_t*)(scratch + L.o_entries));
  }
}

}  // namespace sgr
