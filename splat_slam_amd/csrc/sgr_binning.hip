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
template <typename LOAD>
__device__ uint32_t block1024_scan(LOAD in, uint32_t* __restrict__ out, int n, uint32_t* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;   // 16 waves
  uint32_t carry = 0;
  for (int base = 0; base < n; base += 4096) {
    int i0 = base + threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k) < n ? in(i0 + k) : 0u;
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

// the exactly sized runs of over-full tiles: the overflow list (tile, rank, key) is filed at start(tile) + rank; the first kBucket
// keys of such a tile stay in its bucket (the tile kernels read both places).  `first` / `stride`: this caller's share of the work.
__device__ __forceinline__ void file_overfull_runs(char* saved, char* scratch, const LOff& L, size_t novf, size_t first, size_t stride) {
  const uint2* __restrict__ ranges = (const uint2*)(saved + L.o_ranges);
  uint64_t* __restrict__ entries = (uint64_t*)(scratch + L.o_entries);
  const OvfEntry* __restrict__ ovf = (const OvfEntry*)(scratch + L.o_ovf);
  for (size_t e = first; e < novf; e += stride) {
    const OvfEntry o = ovf[e];
    const uint64_t pos = (uint64_t)(ranges[(size_t)o.tile * kRngStride].x & ~kOverfull) + o.rank;
    if ((int64_t)pos < L.cap) entries[pos] = o.key;
  }
}

// K2 block 3: the order the tile kernels take the view's 16x16 super tiles in -- LONGEST LISTS FIRST (a counting sort of the super
// tiles by the pairs on their four 8x8 tiles, 16 pairs per class).  A tile kernel launch is ~12 rounds of resident waves and every
// wave runs at a fifth of a SIMD: a 250-splat tile that starts in the last round finishes ~100 us after its neighbours, with the
// chip idle around it.  With the long lists dispatched first the launch ends on its shortest ones.  (Scheduling only: no result
// depends on it; the order inside a class is whatever the LDS atomics make it.)
__device__ void order_super_tiles(char* saved, const LOff& L, SavedHeader* hdr) {
  constexpr int kClasses = 128;
  __shared__ uint32_t hist[kClasses];
  unsigned long long* tile_count = (unsigned long long*)(saved + L.o_tile_count);
  uint32_t* order = (uint32_t*)(saved + L.o_tile_order);
  const int nsuper = L.sgx * L.sgy;                      // (counter word w = super tile w: ((gy + 1) / 2) x gxp words, gxp == sgx)
  auto class_of = [&](int w) {
    const unsigned long long c = tile_count[(size_t)w * (kCntSlotWords / 2)];
    const uint32_t pairs = (uint32_t)(c & 0xffffu) + (uint32_t)((c >> 16) & 0xffffu) + (uint32_t)((c >> 32) & 0xffffu) + (uint32_t)(c >> 48);
    const uint32_t k = (pairs + 15u) >> 4;               // 0: no pair at all
    return (int)(kClasses - 1u - (k < (uint32_t)kClasses - 1u ? k : (uint32_t)kClasses - 1u));     // ascending class = descending length
  };
  for (int k = threadIdx.x; k < kClasses; k += 1024) hist[k] = 0u;
  __syncthreads();
  constexpr int kKeep = 4;                               // classes kept in registers between the two passes (4096 super tiles = 1024 x 1024 px)
  int cls[kKeep];
#pragma unroll
  for (int r = 0; r < kKeep; ++r) {
    const int w = (int)threadIdx.x + r * 1024;
    cls[r] = w < nsuper ? class_of(w) : -1;
    if (cls[r] >= 0) atomicAdd(&hist[cls[r]], 1u);
  }
  for (int w = (int)threadIdx.x + kKeep * 1024; w < nsuper; w += 1024) atomicAdd(&hist[class_of(w)], 1u);
  __syncthreads();
  if (threadIdx.x < 64) {                                // exclusive scan of the 128 class counts by one wave, two per lane
    const uint32_t a = hist[2 * threadIdx.x], b = hist[2 * threadIdx.x + 1];
    const uint32_t inc = wave_scan_add_u32(a + b);
    hist[2 * threadIdx.x] = inc - a - b;
    hist[2 * threadIdx.x + 1] = inc - b;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kKeep; ++r) {
    const int w = (int)threadIdx.x + r * 1024;
    if (cls[r] >= 0) order[atomicAdd(&hist[cls[r]], 1u)] = (uint32_t)w;
  }
  for (int w = (int)threadIdx.x + kKeep * 1024; w < nsuper; w += 1024) order[atomicAdd(&hist[class_of(w)], 1u)] = (uint32_t)w;
  // both readers are done with the counters (this block: the loop above; block 0: its flag): leave them clean for the next forward.
  // Block 0 used to, on its critical path; here the stores and the wait hide behind block 0's scan and header.
  __syncthreads();
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(&hdr->counts_read, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store(&hdr->counts_read, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  for (int w = threadIdx.x; w < ((L.gy + 1) / 2) * L.gxp; w += 1024) tile_count[(size_t)w * (kCntSlotWords / 2)] = 0ull;
}

// K2: independent jobs per view, one 1024-thread block each (grid = (3 or 4, views)):
//   (0) tile starts: ranges[t] = (start, end) of the tile's run; for tiles with more than kBucket pairs .x carries kOverfull.
//       When no scatter launch follows (`k3_follows` == 0: the caller's measured longest list fits the buckets) this block also
//       files the overflow list, should there be one after all -- correct for any map, just not parallel;
//   (1) segment bases of the partial-slot offsets;
//   (2) the number of visible Gaussians (and the segments' bases in the compact visible list; the list itself is written by the
//       first blocks of the tile kernel's launch, sgr_blend.hip: nothing before the backward reads it).
//   (3) the launch order of the super tiles (order_super_tiles above) -- where it is used (tile_order_used, sgr_common.h).
// The later of blocks 0 and 1 folds the two pair counts into the header.  A map whose lists stay within the buckets (kBucket = 256 since round 6)
// therefore needs no third binning launch at all.
__global__ void __launch_bounds__(1024) tile_scan_kernel(ViewTab tab, LOff L, int k3_follows) {
  __shared__ uint32_t red[16];
  __shared__ uint32_t red_max[16];
  __shared__ uint32_t s_flag;
  char* saved = tab.saved[blockIdx.y];
  SavedHeader* hdr = (SavedHeader*)(saved + L.o_hdr);
  if (blockIdx.x == 0) {
    uint32_t* tmp = (uint32_t*)(saved + L.o_tile_maxc);      // free until blend_fwd overwrites it
    uint2* ranges = (uint2*)(saved + L.o_ranges);
    unsigned long long* tile_count = (unsigned long long*)(saved + L.o_tile_count);
    auto count_of = [&](int t) {
      const int x = t % L.gx, y = t / L.gx;
      return (uint32_t)(tile_count[tile_counter_word(x, y, L.gxp)] >> tile_counter_shift(x, y)) & 0xffffu;
    };
    // ONE pass, 8 tiles per thread (8192 per trip: a 640x480 view has 4800): the scanned values stay in registers and go straight
    // into `ranges` (the first version scanned into a scratch array and read it back: a second trip through L2 on the critical
    // path of every iteration -- this block is what the tile kernels wait for)
    (void)tmp;
    uint32_t R = 0, over = 0, saturated = 0, longest = 0;
    {
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      for (int base = 0; base < L.ntiles; base += 8192) {
        const int i0 = base + (int)threadIdx.x * 8;
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i0 + k) < L.ntiles ? count_of(i0 + k) : 0u;
        uint32_t s8 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s8 += v[k];
        const uint32_t inc = wave_scan_add_u32(s8);
        if (lane == 63) red[wv] = inc;
        __syncthreads();
        uint32_t pre = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t r = red[w]; pre += (w < wv) ? r : 0u; tot += r; }
        __syncthreads();
        uint32_t run = R + pre + inc - s8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if ((i0 + k) < L.ntiles) {
            const uint32_t c = v[k];
            saturated |= c >= kTileCountLimit ? 1u : 0u;
            longest = c > longest ? c : longest;
            // <= kBucket pairs: K1 already binned them in the tile's bucket, the run [s0, s0+c) only addresses point_list.
            // more: the run is completed below / by scatter_kernel; .y ends at s0 + c as well.
            ranges[(size_t)(i0 + k) * kRngStride] = make_uint2(c <= (uint32_t)kBucket ? run : (run | kOverfull), run + c);
            over += c > (uint32_t)kBucket ? 1u : 0u;
          }
          run += v[k];
        }
        R += tot;
      }
    }
    over = wave_scan_add_u32(over);
    saturated = __syncthreads_or((int)saturated) ? 1u : 0u;
    longest = (uint32_t)wave_max_i32((int)longest);
    if ((threadIdx.x & 63) == 0) red_max[threadIdx.x >> 6] = longest;
    // consumed (the barrier above is behind every thread's counter reads).  Block 3 reads the counters as well and is the one that
    // leaves them clean for the next forward -- once it has seen this flag.  (Blocks are dispatched x-fastest: this block is on the chip
    // before its view's block 3 can be, the wait cannot starve.)
    // (relaxed on both sides: nothing this block WROTE is read over there, and its counter reads have returned -- their values are in
    //  the scan above; a release here would write this block's `ranges` back through L2 first)
    if (tile_order_used(L)) {
      if (threadIdx.x == 64) __hip_atomic_store(&hdr->counts_read, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {                                       // no block 3 in this launch
      // (the order array still gets a valid permutation: a later backward of this forward may come with another hint and read it)
      uint32_t* order = (uint32_t*)(saved + L.o_tile_order);
      for (int w = threadIdx.x; w < ((L.gy + 1) / 2) * L.gxp; w += 1024) { tile_count[(size_t)w * (kCntSlotWords / 2)] = 0ull; order[w] = (uint32_t)w; }
    }
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = over;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      uint32_t mx = 0;
      for (int wv = 0; wv < 16; ++wv) { tot += red[wv]; mx = red_max[wv] > mx ? red_max[wv] : mx; }
      hdr->num_overfull = tot;
      hdr->max_tile_count = mx;
      hdr->ovf_count = hdr->ovf_cursor;          // K1 is done appending; leave the cursor clean for the next forward
      hdr->ovf_cursor = 0u;
      hdr->num_rendered = R;
      // 1: more pairs than the workspace holds (grow it); 2: > kTileCountLimit splats on ONE 8x8 tile (the map has degenerated:
      // a 16-bit tile counter may have carried into its neighbour) -- either way the lists of this view are not to be used
      hdr->overflow = (saturated || hdr->count_saturated) ? 2u : ((int64_t)R > L.cap ? 1u : 0u);
      hdr->count_saturated = 0u;
      hdr->sorted_count = (uint32_t)((int64_t)R > L.cap ? L.cap : (int64_t)R);
      s_flag = tot;
    }
    __syncthreads();                              // (also: every thread's `ranges` are visible to the block)
    if (!k3_follows && s_flag != 0u) {
      const size_t novf = hdr->ovf_count < (uint64_t)L.cap ? hdr->ovf_count : (size_t)L.cap;
      file_overfull_runs(saved, tab.scratch[blockIdx.y], L, novf, threadIdx.x, 1024);
    }
  } else if (blockIdx.x == 1) {
    const uint32_t* bt = (const uint32_t*)(saved + L.o_block_touched);
    const uint32_t slots = block1024_scan([&](int i) { return bt[i]; }, (uint32_t*)(saved + L.o_block_base_t), L.nseg, red);
    if (threadIdx.x == 0) hdr->slot_total = slots;
  } else if (blockIdx.x == 3) {
    order_super_tiles(saved, L, hdr);
    return;
  } else if (blockIdx.x == 2) {
    const uint32_t* bv = (const uint32_t*)(saved + L.o_block_vis);
    uint32_t V = block1024_scan([&](int i) { return bv[i]; }, (uint32_t*)(saved + L.o_block_base_v), L.nseg, red);
    if (threadIdx.x == 0) hdr->num_visible = V;
    return;
  } else {
    return;
  }
  // the later of blocks 0 and 1 folds the two pair counts: what a caller sizes the workspace by is the larger of the pairs
  // binned and the partial slots reserved (bins the exact footprint test dropped keep their slot)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&hdr->k2_tickets, 1u) == 1u) {
      __threadfence();
      hdr->k2_tickets = 0u;
      const uint32_t binned = *(volatile uint32_t*)&hdr->num_rendered, slots = *(volatile uint32_t*)&hdr->slot_total;
      hdr->num_binned = binned;
      if (slots > binned) hdr->num_rendered = slots;
      uint32_t ov = *(volatile uint32_t*)&hdr->overflow;
      if ((int64_t)slots > L.cap && ov == 0u) hdr->overflow = ov = 1u;
      // the sticky pair (see SavedHeader): this forward's verdict outlives the next forward's header
      if (ov != 0u) hdr->overflow_events = hdr->overflow_events + 1u;
      const uint32_t demanded = slots > binned ? slots : binned;
      if (demanded > hdr->max_rendered) hdr->max_rendered = demanded;
    }
  }
}

// K3 (only launched when long lists are expected): the exactly sized runs of over-full tiles, completed in parallel WITHOUT
// atomics: K1 appended every pair whose rank in its tile was >= kBucket to the view's overflow list (tile, rank, key); with the
// tile starts of K2 its place is start(tile) + rank, and the first kBucket keys are copied over from the bucket.  (The first
// version re-scattered all pairs of such tiles with a returning atomic per pair from a loop over each Gaussian's rectangle:
// 1.3 ms per launch on a converged map, where splats cover tens to hundreds of tiles.)  Order inside a run / bucket is
// arbitrary; K4 sorts it.
__global__ void __launch_bounds__(256) scatter_kernel(ViewTab tab, LOff L) {
  const int v = blockIdx.y;
  char* saved = tab.saved[v];
  SavedHeader* hdr = (SavedHeader*)(saved + L.o_hdr);
  if (hdr->num_overfull == 0) return;
  const size_t novf = hdr->ovf_count < (uint64_t)L.cap ? hdr->ovf_count : (size_t)L.cap;
  file_overfull_runs(saved, tab.scratch[v], L, novf, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

// header + per-tile pair counters of every view of the batch in one launch (replaces one memset per view)
__global__ void __launch_bounds__(256) zero_heads_kernel(ViewTab tab, LOff L, size_t nwords) {
  uint32_t* p = (uint32_t*)(tab.saved[blockIdx.y] + L.o_hdr);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

void launch_zero_heads(const ViewTab& tab, int nviews, const LOff& L, size_t zero_bytes, hipStream_t st) {
  size_t nwords = zero_bytes / 4;
  int blocks = (int)((nwords + 255) / 256);
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(zero_heads_kernel, dim3(blocks, nviews), dim3(256), 0, st, tab, L, nwords);
}

void launch_binning(const ViewTab& tab, int nviews, const LOff& L, hipStream_t st) {
  // the scatter launch is only worth its dispatch when lists beyond the buckets are expected: L.mean_hint = the caller's measured
  // longest list (0: unknown -> launch it).  Without it K2's first block files whatever overflowed (correct, not parallel).
  const int k3 = (L.N > 0 && (L.mean_hint == 0 || L.mean_hint > kBucket)) ? 1 : 0;
  {
    ProfScope prof(PK_SCAN, st);
    hipLaunchKernelGGL(tile_scan_kernel, dim3(tile_order_used(L) ? 4 : 3, nviews), dim3(1024), 0, st, tab, L, k3);
  }
  if (k3) {
    ProfScope prof(PK_SCATTER, st);
    hipLaunchKernelGGL(scatter_kernel, dim3((L.nseg + 3) / 4, nviews), dim3(256), 0, st, tab, L);
  }
}

}  // namespace sgr
