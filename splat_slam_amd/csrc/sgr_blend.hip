// Tile blending for gfx950: per-tile list sort + front-to-back alpha compositing (forward) and its gradient (backward).
//
// Replaces renderCUDA (forward/backward) -- and, together with sgr_binning.hip, the global radix sort -- of the
// un-vendored diff-gaussian-rasterization-w-pose module reached from
// /root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:130-141 and from loss.backward() at
// /root/reference/src/mapper.py:329,490,699.
//
// MI355X design (not the CUDA 16x16-block/atomicAdd scheme):
//   * a bin is 8x8 pixels = exactly one wave64 = one workgroup (no __syncthreads in either kernel); the four tiles of a
//     16x16 reference tile run back to back on the same XCD (tile_of_block).
//   * FORWARD first sorts its tile's (depth bits | Gaussian) keys: <= 64 keys in registers (every lane ranks its key
//     against the others, broadcast through LDS), <= 512 keys bitonic in registers, <= 1024 / 4096 in the wave's LDS slice, longer lists in
//     place in HBM (slow path).  Then it is pixel-parallel (lane = pixel): 64 sorted splats at a time are staged into LDS
//     pair-interleaved and walked TWO splats per trip with broadcast ds_read_b128 + packed fp32 (v_pk_*_f32); per-pixel
//     accumulators stay in VGPRs.  n_touched is one ballot+popcount+atomic per (wave, splat), not per pixel.
//   * BACKWARD is splat-parallel (lane = splat).  For one pixel the transmittance in front of every splat is a
//     multiplicative DPP scan over lanes and the colour behind it an additive DPP scan of ONE scalar
//     (w_j = dL/dC . rgb_j + dL/dD * depth_j): 2 scans per (pixel, list) instead of the 10 cross-lane reductions of a
//     pixel-parallel backward; the 10 per-splat gradient sums accumulate in that lane's registers.  Short lists do not
//     waste lanes: with <= 16 (<= 32) splats the wave processes 4 (2) pixels at once, one per 16- (32-)lane group.
//     Each (tile, splat) pair writes its 48-byte partial to a slot owned by the Gaussian: no atomics, bitwise
//     run-to-run deterministic; preprocess_bwd gathers the slots in fixed order.
#include <type_traits>

#include "sgr_common.h"

#ifndef SGR_TILE_WAVES
#define SGR_TILE_WAVES 5      // fused tile kernel: 84 VGPRs; 4 waves per SIMD measured 10 % slower, 6 (80 VGPRs, 4 spilled) no faster
#endif
// The stand-alone backward is the kernel of FRESH maps' drop-in path and of the north-star roofline: lists of ~11, latency hidden by
// resident waves.  Round 6's row-wise loop cost it its sixth wave per SIMD (74 -> 85 VGPRs: light scene 78 -> 83 us despite 5 % fewer
// instructions); with HALF rows per outer iteration of a 64-lane chunk (74.5 instead of 66 instructions per pair there, 81 in round 5)
// it fits 80 registers again -- one value spilled around the 64-lane loop, stored and reloaded once per CHUNK.  The fused kernel
// keeps whole rows: its LDS slice bounds it to 5 waves per SIMD anyway.
#ifndef SGR_BWD_WAVES
#define SGR_BWD_WAVES 6       // stand-alone backward
#endif
#ifndef SGR_BWD_KPR64
#define SGR_BWD_KPR64 2       // stand-alone backward: pairs per outer iteration of a 64-lane chunk (see bwd_chunk2)
#endif

namespace sgr {

// Lists the wave sorts on chip: three builds of the forward kernel, picked per launch from the LONGEST list the caller has
// measured for these cameras (SgrWorkspace.max_list_hint).  "light": up to 512 keys sorted IN REGISTERS (below), only the
// sorted Gaussian indices (4 B each) go to LDS -- 7 KB per wave, 20 waves per CU stay resident -- and it is the build that runs forward
// and backward of a tile in one wave; "mid" / "heavy": 1024 / 4096 64-bit keys bitonic in LDS (11 / 35 KB per wave).
constexpr int kSortLight = 512, kSortMid = 1024, kSortHeavy = 4096;
constexpr bool sort_in_registers(int sort_max) { return sort_max <= kSortLight; }

// Bitonic sort of a[0..n) for ANY n with one wave ("mirror" formulation: every compare-exchange is ascending, so
// virtual +inf padding behind n never moves).  LOAD/STORE abstract LDS vs. device-coherent global memory.
template <typename LD, typename ST, typename SYNC>
__device__ __forceinline__ void wave_sort_any(int n, int lane, LD load, ST store, SYNC sync) {
  int P = 1;
  while (P < n) P <<= 1;
  for (int k = 2, lk = 1; k <= P; k <<= 1, ++lk) {
    for (int j = k >> 1, lj = lk - 1; j > 0; j >>= 1, --lj) {       // j = 1 << lj: shifts, not the integer divisions t / j, t % j
      const bool first = (j == (k >> 1));                           // (half of a 129..256-entry tile's forward was this sort)
      for (int t = lane; t < (P >> 1); t += kWave) {
        int i = ((t >> lj) << (lj + 1)) + (t & (j - 1));            // lower index of the pair
        int l = first ? (i ^ (k - 1)) : (i ^ j);                    // mirror partner on the first step of a merge
        if (l < i) { int tmp = i; i = l; l = tmp; }
        if (l < n) {
          uint64_t a = load(i), b = load(l);
          if (b < a) { store(i, b); store(l, a); }
        }
      }
      sync();
    }
  }
}

// ---- register-blocked bitonic sort of 64 * KPL keys: lane l holds elements l * KPL .. l * KPL + KPL - 1.  A stage of stride j
// compares elements e and e ^ j: for j < KPL both live in the same lane's registers (no data movement at all), for j >= KPL the
// partner is the same register of lane l ^ (j / KPL): one DPP quad_perm (lane ^ 1, lane ^ 2), ds_swizzle (^ 4, 8, 16) or
// ds_bpermute (^ 32) per dword -- no LDS storage, no barrier, no index arithmetic.  256 keys: 15 of the 36 stages stay in
// registers; the LDS-resident network this replaces spent ~40 instructions per stage (index arithmetic + two round trips +
// a fence): half of a 129..256-entry tile's forward.  Padding keys (~0) sort to the end.
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
  else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1f | (M << 10));          // bit mode: xor mask M
  else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ M) << 2, (int)v);
}
template <int KPL, int KK, int J>
struct SortStage {
  static __device__ __forceinline__ void run(uint64_t (&k)[KPL], int lane) {
    if constexpr (J < KPL) {
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        if ((r & J) != 0) continue;
        // block of KK elements ascending iff (e & KK) == 0, e = lane * KPL + r
        // (keys are unique up to the ~0 padding, so "a < b" is "not b < a": ONE compare, the direction is an xor of lane masks)
        const bool down = KK < KPL ? ((r & KK) != 0) : (((lane * KPL) & KK) != 0);
        const uint64_t a = k[r], b = k[r | J];
        const bool sw = (b < a) != down;
        k[r] = sw ? b : a;
        k[r | J] = sw ? a : b;
      }
    } else {
      constexpr int M = J / KPL;
      const bool lower = (lane & M) == 0;
      const bool up = ((lane * KPL) & KK) == 0;
      const bool keep_max = lower != up;
#pragma unroll
      for (int r = 0; r < KPL; ++r) {
        const uint64_t mine = k[r];
        const uint64_t other = ((uint64_t)lane_xor<M>((uint32_t)(mine >> 32), lane) << 32) | lane_xor<M>((uint32_t)mine, lane);
        const bool take = (other < mine) != keep_max;
        k[r] = take ? other : mine;
      }
    }
    if constexpr (J > 1) SortStage<KPL, KK, J / 2>::run(k, lane);
  }
};
template <int KPL, int KK>
struct SortMerge {
  static __device__ __forceinline__ void run(uint64_t (&k)[KPL], int lane) {
    SortStage<KPL, KK, KK / 2>::run(k, lane);
    if constexpr (KK < kWave * KPL) SortMerge<KPL, KK * 2>::run(k, lane);
  }
};
// sorts keys_in[0..count) (count <= 64 * KPL) and leaves the sorted Gaussian indices (low words) in ids[0..64 * KPL)
// (the keys of an over-full tile live in two places: ranks 0..kBucket-1 in the tile's bucket, where K1 binned them, the rest in the
//  tile's exactly sized run, where K3 filed the overflow list -- `run` is addressed by rank as well; nobody copies the bucket part over)
template <int KPL>
__device__ __forceinline__ void wave_sort_registers(const uint64_t* __restrict__ bucket, const uint64_t* __restrict__ run, int count, int lane,
                                                    uint32_t* ids /*LDS*/) {
  uint64_t k[KPL];
#pragma unroll
  for (int r = 0; r < KPL; ++r) {
    const int e = lane * KPL + r;
    k[r] = e < count ? (e < kBucket ? bucket[e] : run[e]) : ~0ull;
  }
  SortMerge<KPL, 2>::run(k, lane);
#pragma unroll
  for (int r = 0; r < KPL; ++r) ids[lane * KPL + r] = (uint32_t)k[r];
}

// The exponent of a splat's footprint is evaluated in base 2 and with its sign flipped: both tile kernels take the conic scaled to
//   A' = A log2(e) / 2,  B' = B log2(e),  C' = C log2(e) / 2
// where they stage / fetch a splat's record (three multiplies per (tile, splat) pair) and evaluate
//   npow = dx (A' dx + B' dy) + (C' dy) dy  =  -log2(e) * power            (five operations where -(A dx^2 + C dy^2) / 2 - B dx dy took seven)
//   G = exp2(-npow)                                                        (the negation is a source modifier of v_exp_f32)
// Which pairs contribute -- the reference's `power > 0 -> skip` and `alpha < 1/255 -> skip` -- is ONE unsigned integer compare of the bit
// patterns, bits(npow) <= bits(nthr) with nthr = log2(255 opacity) >= 0 per splat: a negative npow (power > 0) has its sign bit set and
// is larger than any non-negative float's pattern, and for npow >= 0 the patterns are ordered like the values, so the compare is
// 0 <= npow <= nthr  <=>  power <= 0 and opacity * G >= 1/255 (up to the rounding of the logarithm: the reference's own fp32 expf puts
// the alpha = 1/255 level set no more exactly; parity tests move scenes off that knife edge).  It replaced two floating-point compares
// (+ an and) per (pixel, splat) in both walks.  A splat with 255 * opacity < 1 can never contribute: it is staged as a splat far outside
// any image with nthr = +0.  Forward and backward evaluate the same IEEE operation sequence, so they agree bit for bit on who contributes.
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kStageFloats = 14;       // staged floats per splat (pair-interleaved: 7 float4 per PAIR of splats)
constexpr int kStageBytes = kStageFloats * 4;
constexpr int kStash = 16, kStashStride = 66, kStashOffset = (kStash / 2 + 1) * 2 * kStageBytes;      // (the staging bytes of 16 splats + the pad pair)
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
struct Shape { float mx, my, A, B, C, op, nthr; };
__device__ __forceinline__ Shape stage_shape(float px, float py, float A, float B, float C, float opacity) {
  const float x255 = 255.f * opacity;
  Shape s;
  if (x255 >= 1.f) {
    s.mx = px; s.my = py; s.A = A * (0.5f * kLog2e); s.B = B * kLog2e; s.C = C * (0.5f * kLog2e); s.op = opacity;
    s.nthr = __builtin_amdgcn_logf(x255);        // v_log_f32 = log2 (>= +0 for x255 >= 1)
  } else {                                       // (also the zero splat that pads an odd list: npow is huge, +0 never reaches it)
    s.mx = -1.0e6f; s.my = -1.0e6f; s.A = 1.f; s.B = 0.f; s.C = 1.f; s.op = 0.f; s.nthr = 0.f;
  }
  return s;
}
// bits(npow) <= bits(nthr) as unsigned integers (see above)
__device__ __forceinline__ bool in_footprint(float npow, float nthr) { return __float_as_uint(npow) <= __float_as_uint(nthr); }

// gfx950 has packed fp32 (v_pk_mul/add/fma_f32: two IEEE fp32 results per issue slot); both blend kernels use 2-vectors
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }

// Wave-wide reductions of the forward's epilogue as single-instruction DPP steps (a lane whose DPP source is out of range
// keeps its value).  Same step sequence -- hence the same association of the sum -- as wave_scan_add().
#define SGR_ROW_STEPS(X)                                                                                            \
  X("row_shr:1 row_mask:0xf bank_mask:0xf") X("row_shr:2 row_mask:0xf bank_mask:0xf") X("row_shr:4 row_mask:0xf bank_mask:0xf") \
  X("row_shr:8 row_mask:0xf bank_mask:0xf") X("row_bcast:15 row_mask:0xa bank_mask:0xf") X("row_bcast:31 row_mask:0xc bank_mask:0xf")
#define SGR_MAX1(CTRL) "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 " CTRL "\n\t"
#define SGR_ADD4(CTRL)                                                                                       \
  "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL \
  "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // maximum over the wave, in every lane
  asm(SGR_ROW_STEPS(SGR_MAX1) "s_nop 1" : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Four wave totals at once: the HALVES of the wave trade places (v_permlane32_swap: a.hi <-> b.lo), so one swap + one add folds TWO
// values over the half-wave distance; v_permlane16_swap does the same over the row distance; the last 16 lanes are four DPP row
// rotations of ONE register.  10 instructions where four separate DPP reductions took 24 + 4 readlanes.  Returns, in lane 0 of
// row r (lanes 0, 16, 32, 48): the total of a, c, b, d (in this order).  Fixed association: bitwise reproducible.
__device__ __forceinline__ float wave_sum4_rows(float a, float b, float c, float d) {
  asm("s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1\n\t"
      "v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\ts_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\ts_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return a;
}

// Per-pixel state that only the two blend kernels exchange is stored TILE-major (index tile * 64 + lane): a wave then
// reads / writes one contiguous run instead of eight row pieces of its 8x8 tile.  The loss-gradient code bytes live in
// the caller's [3,H,W] float scratch: tile-major as well whenever that holds 64 bytes per tile (any image but a
// few-pixel one).
__device__ __forceinline__ bool code_bytes_tiled(const LOff& L) {
  return (uint64_t)L.ntiles * 64ull <= 12ull * (uint64_t)L.H * (uint64_t)L.W;
}

// ------------------------------------------------------------------------------------------------ backward
// inclusive scans restricted to groups of GW lanes (GW = 16: one DPP row, 32: two rows, 64: whole wave)
// One scan step as a single VALU op: v_mul_f32_dpp with vdst = src0 = src1.  A lane whose DPP source is out of range
// (bound_ctrl off) or outside row_mask is DISABLED and keeps its value -- exactly the identity a product scan needs
// (the builtin path costs a v_mov_dpp with old = 1.0 plus a v_mul).  s_nop 1 = the 2 wait states a DPP read of a
// freshly written VGPR requires.
#define SGR_MUL_DPP(CTRL) "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 " CTRL "\n\t"
#define SGR_ADD_DPP(CTRL) "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " CTRL "\n\t"
// The backward scans TWO values per lane (its pixel pair): interleaved, each scan's step is the other's wait state, so a
// step of both costs two VALU ops + s_nop 0.
#define SGR_DPP2(OP, CTRL) "v_" OP "_f32_dpp %0, %0, %0 " CTRL "\n\tv_" OP "_f32_dpp %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
#define SGR_SCAN2_ROW(OP)                                                                                     \
  "s_nop 1\n\t" SGR_DPP2(OP, "row_shr:1 row_mask:0xf bank_mask:0xf") SGR_DPP2(OP, "row_shr:2 row_mask:0xf bank_mask:0xf") \
      SGR_DPP2(OP, "row_shr:4 row_mask:0xf bank_mask:0xf") SGR_DPP2(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")
#define SGR_SCAN2(OP, a, b)                                                                                                \
  do {                                                                                                                     \
    if (GW == 16) asm(SGR_SCAN2_ROW(OP) "s_nop 0" : "+v"(a), "+v"(b));                                                      \
    else if (GW == 32) asm(SGR_SCAN2_ROW(OP) SGR_DPP2(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf") "s_nop 0" : "+v"(a), "+v"(b)); \
    else asm(SGR_SCAN2_ROW(OP) SGR_DPP2(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                     \
             SGR_DPP2(OP, "row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 0" : "+v"(a), "+v"(b));                          \
  } while (0)
template <int GW>
__device__ __forceinline__ void group_scan_mul2(float& a, float& b) { SGR_SCAN2("mul", a, b); }
template <int GW>
__device__ __forceinline__ void group_scan_add2(float& a, float& b) { SGR_SCAN2("add", a, b); }
// value of the previous lane inside the group (first lane of a group receives `fill`)
template <int GW>
__device__ __forceinline__ float group_shr1(float v, float fill, int lane) {
  if (GW == 16) return dpp_f<DPP_ROW_SHR1>(fill, v);
  float r = dpp_f<DPP_WAVE_SHR1>(fill, v);
  if (GW == 32) r = (lane == 32) ? fill : r;
  return r;
}
// Groups of 8 / 4 lanes are INTERLEAVED inside their DPP row: the 16 / GW groups of a row own the lanes l16 % (16 / GW), a lane's
// position inside its group is l16 / (16 / GW).  "The previous lane of my group" is then a plain row_shr:(16 / GW) and a scan step
// over distance d inside the group a row_shr:(d * 16 / GW), whose out-of-range lanes (the first d lanes of every group: exactly the
// ones that must keep their value) are disabled by the hardware -- one DPP instruction per step and value.  Contiguous groups
// (lanes 8..15 = the second group) leaked the lower group's tail through row_shr:1 / :2 and paid a copy and a select per step:
// 7 instead of 3 (GW = 8) and 6 instead of 2 (GW = 4) instructions per scan and pixel, 16 of the ~88 of an iteration.
#define SGR_SCAN2_IL8(OP)                                                                                       \
  "s_nop 1\n\t" SGR_DPP2(OP, "row_shr:2 row_mask:0xf bank_mask:0xf") SGR_DPP2(OP, "row_shr:4 row_mask:0xf bank_mask:0xf") \
      SGR_DPP2(OP, "row_shr:8 row_mask:0xf bank_mask:0xf") "s_nop 0"
#define SGR_SCAN2_IL4(OP) \
  "s_nop 1\n\t" SGR_DPP2(OP, "row_shr:4 row_mask:0xf bank_mask:0xf") SGR_DPP2(OP, "row_shr:8 row_mask:0xf bank_mask:0xf") "s_nop 0"
template <>
__device__ __forceinline__ void group_scan_mul2<8>(float& a, float& b) { asm(SGR_SCAN2_IL8("mul") : "+v"(a), "+v"(b)); }
template <>
__device__ __forceinline__ void group_scan_add2<8>(float& a, float& b) { asm(SGR_SCAN2_IL8("add") : "+v"(a), "+v"(b)); }
template <>
__device__ __forceinline__ float group_shr1<8>(float v, float fill, int) { return dpp_f<DPP_ROW_SHR2>(fill, v); }
template <>
__device__ __forceinline__ void group_scan_mul2<4>(float& a, float& b) { asm(SGR_SCAN2_IL4("mul") : "+v"(a), "+v"(b)); }
template <>
__device__ __forceinline__ void group_scan_add2<4>(float& a, float& b) { asm(SGR_SCAN2_IL4("add") : "+v"(a), "+v"(b)); }
template <>
__device__ __forceinline__ float group_shr1<4>(float v, float fill, int) { return dpp_f<DPP_ROW_SHR4>(fill, v); }

// Two uses of "the previous lane of my group" that fold into ONE DPP instruction each (per pixel of the pair):
//   prev_times(): r = prev(p) * m, the group's first lane (no previous lane: its DPP source is out of range, the lane is disabled and
//                 keeps what r held) gets 1 * m -- r is preset to m;
//   prev_plus(): prev(q) + c, the first lane gets 0 + c (bound_ctrl:0 reads an out-of-range source as zero).
// GW = 32 has a first lane in the middle of the wave-wide shift (lane 32): one select puts it right.
#define SGR_PREV2(INS, CTRL, TAIL) "s_nop 1\n\t" INS " %0, %2, %4 " CTRL " row_mask:0xf bank_mask:0xf" TAIL "\n\t" INS " %1, %3, %5 " CTRL " row_mask:0xf bank_mask:0xf" TAIL "\n\ts_nop 1"
// (in place: r enters as m and is its own second source, so no copy of m is made unless the caller still needs it)
#define SGR_PREVT2(CTRL) "s_nop 1\n\tv_mul_f32_dpp %0, %2, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %1, %3, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\ts_nop 1"
template <int GW>
__device__ __forceinline__ v2f prev_times(float p0, float p1, v2f m, int lane) {
  v2f r = m;
  if constexpr (GW == 16) asm(SGR_PREVT2("row_shr:1") : "+v"(r.x), "+v"(r.y) : "v"(p0), "v"(p1));
  else if constexpr (GW == 8) asm(SGR_PREVT2("row_shr:2") : "+v"(r.x), "+v"(r.y) : "v"(p0), "v"(p1));
  else if constexpr (GW == 4) asm(SGR_PREVT2("row_shr:4") : "+v"(r.x), "+v"(r.y) : "v"(p0), "v"(p1));
  else asm(SGR_PREVT2("wave_shr:1") : "+v"(r.x), "+v"(r.y) : "v"(p0), "v"(p1));
  if (GW == 32 && lane == 32) r = m;
  return r;
}
template <int GW>
__device__ __forceinline__ v2f prev_plus(float q0, float q1, v2f c, int lane) {
  v2f r;
  if constexpr (GW == 16) asm(SGR_PREV2("v_add_f32_dpp", "row_shr:1", " bound_ctrl:0") : "=&v"(r.x), "=&v"(r.y) : "v"(q0), "v"(q1), "v"(c.x), "v"(c.y));
  else if constexpr (GW == 8) asm(SGR_PREV2("v_add_f32_dpp", "row_shr:2", " bound_ctrl:0") : "=&v"(r.x), "=&v"(r.y) : "v"(q0), "v"(q1), "v"(c.x), "v"(c.y));
  else if constexpr (GW == 4) asm(SGR_PREV2("v_add_f32_dpp", "row_shr:4", " bound_ctrl:0") : "=&v"(r.x), "=&v"(r.y) : "v"(q0), "v"(q1), "v"(c.x), "v"(c.y));
  else asm(SGR_PREV2("v_add_f32_dpp", "wave_shr:1", " bound_ctrl:0") : "=&v"(r.x), "=&v"(r.y) : "v"(q0), "v"(q1), "v"(c.x), "v"(c.y));
  if (GW == 32 && lane == 32) r = c;
  return r;
}

// The hot loop of the backward, TWO pixels per lane.  blend_bwd is bound by VALU issue (one wave64 op = 4 cycles), and
// gfx950 has packed fp32 (v_pk_mul/add/fma_f32: two IEEE fp32 results per issue slot), so every lane carries its splat
// against a horizontally adjacent pixel PAIR in 2-vectors: the quadratic form, the colour dot product, alpha*T and all
// ten accumulate FMAs issue once per pair; only the DPP scans, exp/rcp and the selects stay per pixel.  Every packed op
// is the same IEEE operation sequence as the forward walk's, so forward and backward still agree bit for bit on which pairs
// contribute.  Branch free: a pair that does not contribute has alpha*T = 0 and G*dL/dalpha = 0.
// (A matrix-core variant -- the 16-lane-group layout is exactly the A/B operand layout of v_mfma_f32_16x16x4_f32, the
// ten sums being two small GEMMs over the pixels -- cut the instruction count by 20 % but ran 8 % slower: fp32 MFMA
// passes contend with the VALU work of the other waves of the SIMD.)

// pixel pair g (0..31) of a tile = the horizontally adjacent pixels 2g, 2g + 1 -- ONE layout of the staged pixel state for
// every group width, so that chunks of different widths can follow each other on the same tile (see tile_backward)

// What a lane needs to know about ITS splat: footprint, colour, depth and the slot of this (tile, Gaussian) pair inside
// the Gaussian's run of partials (0xffffffff: not stored -- beyond the capacity).
struct SplatRec { float mx, my, A, B, C, op, nthr, r, g, b, dep; uint32_t slot; };      // (A, B, C, nthr: stage_shape())
__device__ __forceinline__ uint32_t pair_slot(const char* __restrict__ saved, const LOff& L, uint32_t g, uint32_t rel, uint32_t rect01,
                                              uint32_t rect23, int tx, int ty, int64_t cap) {
  const Rect r = unpack_rect(rect01, rect23);
  const uint64_t s = (uint64_t)abs_offset(saved, L, g, rel) + (uint32_t)((ty - r.y0) * (r.x1 - r.x0) + (tx - r.x0));
  return (int64_t)s < cap ? (uint32_t)s : 0xffffffffu;
}
__device__ __forceinline__ SplatRec splat_from_grec(const GRec* __restrict__ grec, const char* __restrict__ saved, const LOff& L,
                                                    uint32_t g, int tx, int ty, int64_t cap) {
  // everything that depends on g in ONE round trip (the slot is only needed after the loop, its latency is not)
  const float4* rec = (const float4*)(grec + g);
  const float4 m = rec[0], co = rec[1], cd = rec[2];
  const uint32_t rel = ((const uint32_t*)(rec + 3))[1];
  SplatRec s;
  s.slot = pair_slot(saved, L, g, rel, __float_as_uint(m.z), __float_as_uint(m.w), tx, ty, cap);
  const Shape sh = stage_shape(m.x, m.y, co.x, co.y, co.z, co.w);
  s.mx = sh.mx; s.my = sh.my; s.A = sh.A; s.B = sh.B; s.C = sh.C; s.op = sh.op; s.nthr = sh.nthr; s.r = cd.x; s.g = cd.y; s.b = cd.z; s.dep = cd.w;
  return s;
}
// list position -> Gaussian through the index list the forward kernel published (blend_bwd_kernel)
struct SrcPointList {
  const uint32_t* __restrict__ point_list; int64_t begin; const GRec* __restrict__ grec; const char* __restrict__ saved; int tx, ty; int64_t cap;
  __device__ __forceinline__ SplatRec load(int idx, const LOff& L) const { return splat_from_grec(grec, saved, L, point_list[begin + idx], tx, ty, cap); }
};
// fused kernel, list longer than one chunk: the sorted keys are still where the wave sorted them (LDS, or HBM for huge lists)
struct SrcKeys {
  const uint32_t* ids_lds; const uint64_t* keys_lds; const uint64_t* __restrict__ keys_hbm; const GRec* __restrict__ grec; const char* __restrict__ saved; int tx, ty; int64_t cap;
  __device__ __forceinline__ uint32_t gaussian(int idx) const {
    if (ids_lds) return ids_lds[idx];           // light build: the register sort left the sorted indices
    return keys_lds ? (uint32_t)keys_lds[idx] : (uint32_t)__hip_atomic_load(keys_hbm + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ SplatRec load(int idx, const LOff& L) const { return splat_from_grec(grec, saved, L, gaussian(idx), tx, ty, cap); }
};
// The forward walk's staging area: kStageFloats floats per splat, PAIR-interleaved (element e of splats 2p, 2p + 1 side by side at float
// 2 e of pair p's 7 float4): 0 mx  1 my | 2 A'  3 B' | 4 C'  5 opacity | 6 depth  7 Gaussian | 8 r  9 g | 10 b  11 nthr | 12 slot  13 -
__device__ __forceinline__ float* staged_at(float4* lds, uint32_t pos) { return (float*)lds + (pos >> 1) * (2 * kStageFloats) + (pos & 1); }
__device__ __forceinline__ void stage_splat(float4* lds, uint32_t pos, const float4& m, const float4& co, const float4& cd, uint32_t g, uint32_t slot) {
  const Shape sh = stage_shape(m.x, m.y, co.x, co.y, co.z, co.w);
  float* f = staged_at(lds, pos);
  f[0] = sh.mx; f[2] = sh.my; f[4] = sh.A; f[6] = sh.B; f[8] = sh.C; f[10] = sh.op; f[12] = cd.w; f[14] = __uint_as_float(g);
  f[16] = cd.x; f[18] = cd.y; f[20] = cd.z; f[22] = sh.nthr; f[24] = __uint_as_float(slot);
}
// fused kernel, list of one chunk (98 % of the tiles of a SLAM view): the staging area still holds every splat of the tile: no second
// gather from HBM
struct SrcStaged {
  const float* lds;
  __device__ __forceinline__ SplatRec load(int idx, const LOff&) const {
    const float* f = lds + (idx >> 1) * (2 * kStageFloats) + (idx & 1);
    SplatRec s;
    s.mx = f[0]; s.my = f[2]; s.A = f[4]; s.B = f[6]; s.C = f[8]; s.op = f[10]; s.dep = f[12];
    s.r = f[16]; s.g = f[18]; s.b = f[20]; s.nthr = f[22]; s.slot = __float_as_uint(f[24]);
    return s;
  }
};

// One chunk = the list positions [start, start + GW) (those below `end`) against the 64 pixels of the tile, 2*64/GW pixels
// per iteration.  Lanes are mapped to splats in REVERSE list order inside their group, so "everything behind me" is a prefix
// scan.  Chunks run back to front; `carry`: more (nearer) chunks follow, leave (T, S) in front of this chunk per pixel.
// LDS: pixA2[g] = (dCr0,dCr1, dCg0,dCg1 | dCb0,dCb1, dD0,dD1), pixB2[g] = (T0,T1, S0,S1 | -,-, nc0,nc1) for pair g.
//
// Round 6: the loop runs ROW by row of the tile.  Pixel pair g sits in row g / 4 at x = 2 (g % 4): a lane of a 64-wide chunk meets the
// four pairs of a row one after the other (32 lanes: two), so everything that only depends on the row -- dy, B' dy, C' dy^2 and the
// three gradient sums that carry a factor dy (sum G dL/dG dy, ... dx dy, ... dy^2 = dy * or dy^2 * a ROW's sum of G dL/dG (dx)) -- is
// formed once per row, and the lane's dx against its (at most four) pair columns once per CHUNK; the coordinates come from the
// tile's position, not from LDS.  80 -> 66 instructions per iteration at 64 lanes.
template <int GW, typename SRC, bool STASH = false, int KPR64 = 4>
__device__ __forceinline__ void bwd_chunk2(
    int lane, int start, int end, bool carry, int tx, int ty, const float4* pixA2 /*LDS*/, float4* pixB2 /*LDS*/, const SRC& src,
    const LOff& L, float4* __restrict__ partials, const float* stash = nullptr /*LDS: exp2(-npow) of the forward walk, see blend_fwd_kernel*/) {
  constexpr int PP = kWave / GW;                 // pixel pairs the wave works on at once
  constexpr int KPR = GW == 64 ? KPR64 : (GW == 32 ? 2 : 1);   // pairs of one row that ONE lane meets per outer iteration (64 lanes: 4 -- or 2: fewer registers --, 32: 2, narrower: 1)
  constexpr int KST = PP >= 4 ? 0 : PP;          // ... their distance in pair columns
  constexpr int PPO = PP >= 4 ? PP : PP * KPR;   // pairs the wave covers per outer iteration (a whole row = 4 or a part of one)
  constexpr int KCOL = PP >= 4 ? 1 : 4 / PP;     // pair columns of a row that one lane meets at all
  // (opaque to the optimiser: the five instantiations sit in one loop, and hoisting each one's lane arithmetic out of it
  //  cost more live registers than the kernel has at 5 waves per SIMD -- they were spilled to scratch: +37 MB of writes per launch)
  asm volatile("" : "+v"(lane));
  constexpr int GPR = GW < 16 ? 16 / GW : 1;     // groups per DPP row (narrow groups are interleaved inside their row, see above)
  const int sub = GW < 16 ? (lane >> 4) * GPR + (lane & (GPR - 1)) : lane / GW;      // which of them this lane works on
  const int sl = GW < 16 ? (lane & 15) / GPR : lane % GW;                            // its position inside its group
  int idx = start + (GW - 1 - sl);               // list position of this lane's splat
  const bool valid = idx < end;
  float mx = 0.f, my = 0.f, A = 0.f, B = 0.f, Cc = 0.f, op = 0.f, nthr = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, dep = 0.f;
  uint32_t slot = 0xffffffffu;                   // this (tile, Gaussian) pair's slot inside the Gaussian's run of partials
  if (valid) {
    const SplatRec sr = src.load(idx, L);
    mx = sr.mx; my = sr.my; A = sr.A; B = sr.B; Cc = sr.C; op = sr.op; nthr = sr.nthr; cr = sr.r; cg = sr.g; cb = sr.b; dep = sr.dep; slot = sr.slot;
  }
  const float* my_stash = STASH ? stash + (valid ? idx : start) * kStashStride : nullptr;
  if (!valid) idx = 0x7fffffff;                  // (an empty lane is behind every pixel's last contributor: `idx < nc` rejects it)
  v2f s_gx, s_gy, s_gxx, s_gxy, s_gyy, a_o, a_r, a_g, a_b, a_d;       // (64-bit moves: the compiler cleared the twenty halves one by one)
  asm("v_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0\n\tv_mov_b64 %4, 0\n\t"
      "v_mov_b64 %5, 0\n\tv_mov_b64 %6, 0\n\tv_mov_b64 %7, 0\n\tv_mov_b64 %8, 0\n\tv_mov_b64 %9, 0"
      : "=v"(s_gx), "=v"(s_gy), "=v"(s_gxx), "=v"(s_gxy), "=v"(s_gyy), "=v"(a_o), "=v"(a_r), "=v"(a_g), "=v"(a_b), "=v"(a_d));
  // this lane's pair columns: x of the pair's first pixel = 8 tx + 2 (column); both pixels of a pair share the row.  Narrow chunks keep
  // dx = mx - x of their one or two columns in registers for the whole chunk; a 64-wide chunk meets all four columns, which are the same
  // for every lane: their x live in SGPRs (eight VGPRs of dx put the fused kernel over its 96) and dx is ONE packed subtract per pair
  const int k0 = sub & 3, row0 = sub >> 2;
  v2f dxs[KCOL];
  float xs0[KCOL], xs1[KCOL];
#pragma unroll
  for (int h = 0; h < KCOL; ++h) {
    const float xf = (float)(tx * kTile + 2 * (k0 + h * KST));
    if constexpr (GW == kWave) {
      xs0[h] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xf)));
      xs1[h] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xf + 1.f)));
    } else {
      dxs[h] = splat2(mx) - (v2f){xf, xf + 1.f};   // (the forward walk's subtraction: both sides are exact integers up to here)
    }
  }
  // rows none of whose pixels has a contributor inside or behind this chunk need no work (a 64-wide chunk of a long list only: the
  // far end of a list is behind the last contributor of most pixels once a tile saturates); one LDS read + compare per CHUNK
  unsigned long long live = ~0ull;
  if (GW == kWave) {
    const int nc_mine = __float_as_int(((const float*)pixB2)[(lane >> 1) * 8 + 6 + (lane & 1)]);
    live = __builtin_amdgcn_ballot_w64(nc_mine > start);
  }
  const int gp0 = row0 * 4 + k0;                 // this lane's pair in the wave's first outer iteration
  const float y0f = (float)(ty * kTile + row0);
  float yrun = y0f;                              // (whole rows per outer iteration: the row's y is a running sum, no conversion in the loop)

#pragma unroll 1
  for (int o = 0; o < 32 / PPO; ++o, yrun += (float)(PPO / 4)) {
    const int prow = (o * PPO) >> 2;             // the tile row of this outer iteration's pairs (narrow groups: of the group's first row)
    if (GW == kWave && ((live >> (8 * prow)) & 0xffull) == 0ull) continue;
    const float yf = PPO >= 4 ? yrun : y0f + (float)prow;
    const float dy = my - yf;
    const float Bdy = B * dy, Cdy2 = (Cc * dy) * dy;
    v2f r_gg = {0.f, 0.f}, r_gx = {0.f, 0.f};    // a half row's sums of G dL/dG and G dL/dG dx (KPR > 1)
#pragma unroll
    for (int h = 0; h < KPR; ++h) {
      // (the unrolled pairs of a row stay one after the other: left alone, the scheduler moves the LDS reads of all four to the top and
      //  the kernel spills ~200 registers)
      if constexpr (KPR > 1) __builtin_amdgcn_sched_barrier(0);
      const int gp = o * PPO + gp0 + h * KST;             // this lane's pixel pair (same for the whole group)
      const float4 a0 = pixA2[gp * 2], a1 = pixA2[gp * 2 + 1];     // LDS, broadcast inside the group
      const float4 b0 = pixB2[gp * 2];
      const int2 ncp = *(const int2*)((const float*)(pixB2 + gp * 2 + 1) + 2);
      const int nc0 = ncp.x, nc1 = ncp.y;
      v2f dx;
      if constexpr (GW == kWave) {
        if constexpr (KPR == KCOL) dx = splat2(mx) - (v2f){xs0[h], xs1[h]};
        else dx = splat2(mx) - ((o & 1) ? (v2f){xs0[2 + h], xs1[2 + h]} : (v2f){xs0[h], xs1[h]});      // (half rows: a scalar select)
      } else dx = dxs[h];
      v2f G;
      bool ok0 = idx < nc0, ok1 = idx < nc1;
      if constexpr (STASH) {
        const float2 gs = *(const float2*)(my_stash + 2 * gp);     // what the forward walk computed for these two pixels (0: outside the footprint)
        G = (v2f){gs.x, gs.y};
      } else {
        // the forward walk's footprint evaluation on the pair, same operation order
        const v2f u = __builtin_elementwise_fma(splat2(A), dx, splat2(Bdy));
        const v2f npow = __builtin_elementwise_fma(u, dx, splat2(Cdy2));
        G = (v2f){exp2_fast(-npow.x), exp2_fast(-npow.y)};
        ok0 = ok0 && in_footprint(npow.x, nthr);
        ok1 = ok1 && in_footprint(npow.y, nthr);
      }
      v2f og = splat2(op) * G;
      // a pair that does not contribute takes part with opacity * G = 0: alpha = 0 (factor 1 in the product, weight 0 in the sums) and
      // G dL/dG = 0 -- ONE select per pixel masks everything downstream
      og.x = ok0 ? og.x : 0.f;
      og.y = ok1 ? og.y : 0.f;
      static_assert(kAlphaMax == 0.99f, "the literal 0x3f7d70a4 below is 0.99f");
      v2f alpha;                                           // min(0.99, og) (as asm: behind a select fminf() first canonicalises its operand)
      asm("v_min_f32 %0, 0x3f7d70a4, %2\n\tv_min_f32 %1, 0x3f7d70a4, %3" : "=&v"(alpha.x), "=v"(alpha.y) : "v"(og.x), "v"(og.y));
      const v2f one_m = splat2(1.f) - alpha;
      float P0 = one_m.x, P1 = one_m.y;
      group_scan_mul2<GW>(P0, P1);                         // prod over this splat and all behind it (in chunk)
      const v2f rP = {__builtin_amdgcn_rcpf(P0), __builtin_amdgcn_rcpf(P1)};
      const v2f Tj = (v2f){b0.x, b0.y} * rP;               // transmittance in front of splat j
      // 1 / (1 - alpha_j) = (prod over all strictly behind it) / (prod incl. it).  (Scanning the reciprocals 1 / (1 - alpha) instead --
      // the factor is then the scan's INPUT -- saves these two DPP multiplies and costs two copies: the scan works in place.  Measured
      // in the ISA, round 6: 264 instructions per row either way.)
      const v2f inv1ma = prev_times<GW>(P0, P1, rP, lane);
      const v2f dCr = {a0.x, a0.y}, dCg = {a0.z, a0.w}, dCb = {a1.x, a1.y}, dD = {a1.z, a1.w};
      const v2f w = __builtin_elementwise_fma(dCr, splat2(cr), __builtin_elementwise_fma(dCg, splat2(cg),
                    __builtin_elementwise_fma(dCb, splat2(cb), dD * splat2(dep))));
      const v2f aT = alpha * Tj;                           // (Tj is finite: the product only spans contributing splats)
      const v2f q = w * aT;
      float Q0 = q.x, Q1 = q.y;
      group_scan_add2<GW>(Q0, Q1);                         // inclusive: this splat and all behind it
      const v2f Sc = {b0.z, b0.w};
      const v2f Sx = prev_plus<GW>(Q0, Q1, Sc, lane);      // strictly behind (+ carried chunks + background term)
      const v2f dL_dalpha = __builtin_elementwise_fma(Tj, w, -(Sx * inv1ma));
      if (carry && sl == GW - 1) {
        // carry to the next (nearer) chunk: the last lane of the group holds the nearest splat of this chunk
        const v2f S2 = (v2f){Q0, Q1} + Sc;
        pixB2[gp * 2] = make_float4(Tj.x, Tj.y, S2.x, S2.y);
      }
      a_r = __builtin_elementwise_fma(aT, dCr, a_r);
      a_g = __builtin_elementwise_fma(aT, dCg, a_g);
      a_b = __builtin_elementwise_fma(aT, dCb, a_b);
      a_d = __builtin_elementwise_fma(aT, dD, a_d);
      const v2f gg = og * dL_dalpha;                       // G * dL/dG = opacity * G * dL/dalpha (the alpha clamp is straight-through)
      const v2f gxv = gg * dx;
      s_gxx = __builtin_elementwise_fma(gxv, dx, s_gxx);
      if constexpr (KPR > 1) {
        // the sums that carry a factor dy are folded per HALF row (two pairs), whatever part of a row an outer iteration covers: the
        // stand-alone backward runs half rows (SGR_BWD_KPR64), the fused kernel whole rows, and the two must stay bitwise equal
        r_gg = (h & 1) == 0 ? gg : r_gg + gg;
        r_gx = (h & 1) == 0 ? gxv : r_gx + gxv;
        if ((h & 1) == 1) {
          a_o += r_gg;
          s_gx += r_gx;
          s_gy = __builtin_elementwise_fma(r_gg, splat2(dy), s_gy);
          s_gxy = __builtin_elementwise_fma(r_gx, splat2(dy), s_gxy);
          s_gyy = __builtin_elementwise_fma(r_gg, splat2(dy * dy), s_gyy);
        }
        // (pins this pair's accumulations HERE: the optimiser otherwise sinks the sums of all unrolled pairs of the row behind the
        //  last one -- everything they read stays live across the row: 137 VGPRs)
        asm volatile("" : "+v"(a_r), "+v"(a_g), "+v"(a_b), "+v"(a_d), "+v"(s_gxx), "+v"(r_gg), "+v"(r_gx));
      } else {
        a_o += gg;                                         // dL/dopacity = sum G dL/dalpha = (this sum) / opacity, divided once after the loop
        s_gx += gxv;
        s_gy = __builtin_elementwise_fma(gg, splat2(dy), s_gy);
        s_gxy = __builtin_elementwise_fma(gxv, splat2(dy), s_gxy);
        s_gyy = __builtin_elementwise_fma(gg, splat2(dy * dy), s_gyy);
      }
    }
  }
  float t_gx = s_gx.x + s_gx.y, t_gy = s_gy.x + s_gy.y, t_gxx = s_gxx.x + s_gxx.y, t_gxy = s_gxy.x + s_gxy.y,
        t_gyy = s_gyy.x + s_gyy.y, t_o = (a_o.x + a_o.y) * (op > 0.f ? __builtin_amdgcn_rcpf(op) : 0.f), t_r = a_r.x + a_r.y,
        t_g = a_g.x + a_g.y, t_b = a_b.x + a_b.y, t_d = a_d.x + a_d.y;
  // The slot of this (tile, Gaussian) pair, 12 floats: gx gy gxx gxy | o r g b | gyy d - -  (RAW sums: the conic / half-image
  // factors that turn them into dL/dmean2D and dL/dconic are per Gaussian, the dense backward applies them once to the sum over
  // the Gaussian's tiles; the last 8 bytes are padding, never written or read).
  float* __restrict__ out = (float*)partials + (size_t)slot * 12;
  if (GW == kWave) {
    if (slot != 0xffffffffu) {
      *(float4*)(out + 0) = make_float4(t_gx, t_gy, t_gxx, t_gxy);
      *(float4*)(out + 4) = make_float4(t_o, t_r, t_g, t_b);
      *(float2*)(out + 8) = make_float2(t_gyy, t_d);
    }
    return;
  }
  // GW < 64: the 64 / GW groups saw disjoint pixels and their ten sums must meet.  As a butterfly of ds_bpermute shuffles that
  // was 10 x log2(64 / GW) shuffles + adds (88 instructions for a 4-lane chunk: more than its two loop iterations).  Here the
  // values are paired and the HALVES of the wave trade places (v_permlane32_swap: a.hi <-> b.lo), so one swap + one add folds
  // TWO values over the half-wave distance, each total landing in the half that keeps it; v_permlane16_swap does the same over
  // the row distance (a.row1 <-> b.row0, a.row3 <-> b.row2), and the distances inside a 16-lane row are DPP row rotations
  // folded into the add.  5 + 3 registers instead of 10 per level, and every row ends up owning the two or three floats it
  // stores: row 0 (gx gy; gyy), row 1 (gxx gxy), row 2 (o r; d), row 3 (g b).  Fixed order: bitwise reproducible.
  asm volatile("s_nop 1\n\t"
               "v_permlane32_swap_b32 %0, %5\n\tv_permlane32_swap_b32 %1, %6\n\tv_permlane32_swap_b32 %2, %7\n\t"
               "v_permlane32_swap_b32 %3, %8\n\tv_permlane32_swap_b32 %4, %9\n\ts_nop 1\n\t"
               "v_add_f32 %0, %0, %5\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_add_f32 %4, %4, %9\n\t"
               "s_nop 1"
               : "+v"(t_gx), "+v"(t_gxx), "+v"(t_gy), "+v"(t_gxy), "+v"(t_gyy), "+v"(t_o), "+v"(t_g), "+v"(t_r), "+v"(t_b), "+v"(t_d));
  // lower half: t_gx t_gxx t_gy t_gxy t_gyy are complete (for GW = 32); upper half: the same registers hold o g r b d
  if (GW == 32) {
    if (slot != 0xffffffffu) {
      const int up = lane >= 32 ? 1 : 0;
      float* o2 = out + 4 * up;
      *(float2*)(o2 + 0) = make_float2(t_gx, t_gy);       // upper half: o r
      *(float2*)(o2 + 2) = make_float2(t_gxx, t_gxy);     //             g b
      out[8 + up] = t_gyy;                                //             d
    }
    return;
  }
  float u01 = t_gx, u23 = t_gy, u4 = t_gyy, w1 = t_gxx, w3 = t_gxy, w4 = 0.f;
  asm volatile("s_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %3\n\tv_permlane16_swap_b32 %1, %4\n\tv_mov_b32 %5, %2\n\ts_nop 1\n\t"
               "v_permlane16_swap_b32 %2, %5\n\ts_nop 1\n\t"
               "v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %5\n\ts_nop 1"
               : "+v"(u01), "+v"(u23), "+v"(u4), "+v"(w1), "+v"(w3), "+v"(w4));
  // rows: u01 = gx | gxx | o | g,  u23 = gy | gxy | r | b,  u4 = gyy | gyy | d | d   (complete for GW = 16)
  // the groups of a row (interleaved: lane ^ 1 for two, lane ^ 2 then lane ^ 1 for four -- the pairing order of the contiguous layout)
  if (GW == 4)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(u01), "+v"(u23), "+v"(u4));
  if (GW <= 8)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(u01), "+v"(u23), "+v"(u4));
  if ((lane & (GPR - 1)) == 0 && slot != 0xffffffffu) {       // the first group of every row stores that row's floats
    const int row = lane >> 4;
    *(float2*)(out + 2 * row) = make_float2(u01, u23);
    if (!(row & 1)) out[8 + (row >> 1)] = u4;            // row 0: gyy, row 2: d
  }
}

// The backward of one tile given this lane's pixel state (pxA = dL/dC r, g, b and dL/dD; pxB[0] = final transmittance,
// pxB[2] = last contributor as uint bits) and a source for the tile's sorted splats.  Used by blend_bwd_kernel (state read
// back from HBM) and by the fused tile kernel (state still in the forward walk's registers).
template <typename SRC, int KPR64 = 4>
__device__ __forceinline__ void tile_backward(int lane, int eff, int tx, int ty, const float pxA[4], float pxB[3],
                                              const float* __restrict__ bg, float4* pixA /*LDS*/, float4* pixB /*LDS*/, const SRC& src,
                                              const LOff& L, float4* __restrict__ partials, const float* stash = nullptr) {
  // the background term -T_final/(1-alpha_j) * (bg . dL/dC) has the same shape as "colour behind splat j"
  pxB[1] = pxB[0] * (bg[0] * pxA[0] + bg[1] * pxA[1] + bg[2] * pxA[2]);
  // pixel p -> pair p / 2, half p % 2
  {
    float* fa = (float*)pixA + (lane >> 1) * 8 + (lane & 1);
    float* fb = (float*)pixB + (lane >> 1) * 8 + (lane & 1);
    fa[0] = pxA[0]; fa[2] = pxA[1]; fa[4] = pxA[2]; fa[6] = pxA[3];
    fb[0] = pxB[0]; fb[2] = pxB[1]; fb[6] = pxB[2];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // The list is cut into chunks of 64 / 32 / 16 / 8 / 4 splats from the far end: a lane = a splat, and a chunk of width GW works
  // on 64 / GW pixel pairs at once, so a chunk costs 32 * GW / 64 iterations of the loop above whatever part of its lanes is
  // filled -- and an iteration costs nearly the same at every width (85 instructions at 64 lanes, 68 at 8: only the scan depth
  // differs).  So the plan is the binary expansion of the list length (24 splats: 16 + 8 = 12 iterations, not one 32-wide chunk of
  // 16), rounded up to the next width only when fewer lanes stay empty than a chunk's prologue + epilogue (~95 instructions) costs.
  int end = eff;
  while (end > 0) {
    const int gw = end >= 61 ? 64 : (end >= 29 ? 32 : (end >= 13 ? 16 : (end >= 5 ? 8 : 4)));
    const int start = gw >= end ? 0 : end - gw;
    const bool carry = start > 0;
    if constexpr (std::is_same<SRC, SrcStaged>::value) {
      if (stash) {            // (a list of <= kStash splats: chunks of 16 / 8 / 4 only)
        if (gw == 16) bwd_chunk2<16, SRC, true>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials, stash);
        else if (gw == 8) bwd_chunk2<8, SRC, true>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials, stash);
        else bwd_chunk2<4, SRC, true>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials, stash);
        end = start;
        if (carry) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        continue;
      }
    }
    if (gw == 64) bwd_chunk2<64, SRC, false, KPR64>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials);
    else if (gw == 32) bwd_chunk2<32>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials);
    else if (gw == 16) bwd_chunk2<16>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials);
    else if (gw == 8) bwd_chunk2<8>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials);
    else bwd_chunk2<4>(lane, start, end, carry, tx, ty, pixA, pixB, src, L, partials);
    end = start;
    if (carry) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// PACKED: the pixel gradients come as the forward's loss epilogue left them (one code byte per pixel, see blend_fwd)
struct SignGrad { const float* exp_a[kMaxViews]; float w_rgb, w_dep; };
__device__ __forceinline__ float sign_code(uint32_t c, float k) { return (c & 1u) ? k : ((c & 2u) ? -k : 0.f); }

// ------------------------------------------------------------------------------------------------ forward
// FUSED (the mapping iteration: loss in the epilogue, backward in the SAME wave): the pixel state the backward needs --
// final transmittance, last contributor, the L1 loss gradient (sign x per-view constant) -- is still in this lane's
// registers and the tile's sorted splats are still staged in LDS, so the wave goes straight on with tile_backward():
// no final_T / n_contrib / code-byte / index-list round trip through HBM, no second launch, one tile prologue.
// The compact visible list of a view (what the dense backward iterates) rides in the tile kernel's launch: `kCompSegs` segments
// per single-wave block, 16 lanes per segment copy that segment's list (K1) to its place -- segment base = the visible counts of
// the segments in front, which the wave adds up itself (<= a few thousand values).  It used to be a set of blocks of K2, on the
// critical path between K1 and the tile kernels (K2 17.6 -> 13.6 us without them); nothing before the backward reads the list.
constexpr int kCompSegs = 4;
__device__ __forceinline__ int comp_blocks(const LOff& L) { return (((L.nseg + kCompSegs - 1) / kCompSegs) + 7) & ~7; }   // (x8: XCD mapping)
__device__ __forceinline__ void compact_visible_list(char* saved, const LOff& L, int j) {       // one wave
  const uint32_t* __restrict__ bv = (const uint32_t*)(saved + L.o_block_vis);
  const int seg0 = j * kCompSegs, lane = threadIdx.x;
  if (seg0 >= L.nseg) return;
  uint32_t before = 0;
  for (int i = lane; i < seg0; i += kWave) before += bv[i];
  before = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add_u32(before), 63);
  const int ls = lane >> 4, sub = lane & 15;
  const uint32_t c = seg0 + ls < L.nseg ? bv[seg0 + ls] : 0u;
  const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, 0), c1 = (uint32_t)__builtin_amdgcn_readlane((int)c, 16),
                 c2 = (uint32_t)__builtin_amdgcn_readlane((int)c, 32);
  const uint32_t b = before + (ls > 0 ? c0 : 0u) + (ls > 1 ? c1 : 0u) + (ls > 2 ? c2 : 0u);
  const uint32_t* __restrict__ seg_list = (const uint32_t*)(saved + L.o_seg_list);
  uint32_t* __restrict__ vis_list = (uint32_t*)(saved + L.o_vis_list);
  GRec* __restrict__ grec = (GRec*)(saved + L.o_grec);
  for (uint32_t k = sub; k < c; k += 16) {
    const uint32_t i = seg_list[(size_t)(seg0 + ls) * kSeg + k];
    grec[i].vis_pos = b + k;
    vis_list[b + k] = i;
  }
}

// Workgroup = ONE wave = one 8x8 tile.  The four waves of a 16x16 super tile used to share a 256-thread workgroup: they never
// synchronise, but a workgroup's LDS (5 workgroups per CU) is released only when its LAST wave retires, and the four lists of a
// super tile differ in length -- the counters showed 3.7 resident waves per SIMD on average where the registers allow 5.
// tile_of_block(): hardware places block b on XCD b % 8.  Round 6: the view's super tiles are taken in K2's launch order (longest lists
// first: order_super_tiles, sgr_binning.hip) and DEALT to the XCDs round robin -- until then every XCD owned a band of the image (a
// contiguous run of super tiles) and the launch ended when the XCD with the heaviest band did (opaque scene: 0.503 -> 0.484 ms from
// the dealing alone).  The four tiles of a super tile still run back to back on one XCD (they share most of their Gaussians: single
// tiles dealt round robin cost +2.5 %).
// Launch geometry of the tile kernels: grid = (32, rows, views).  x = xcd + 8 * (tile of the super tile); y = rank of the super tile in
// its XCD's share of the launch order (+ `lead` rows in front: blocks that do something else first, compact_visible_list); z = view.
// Dispatch is x-fastest, views one after the other.  (Measured and not kept: the views INTERLEAVED -- rank r of every view on the chip
// before rank r + 1 of any, so that the launch ends on the shortest lists of all views: opaque scene +-0, and with the identity
// order of a light map +4 % -- the same image region of twelve views then runs at the same time, and sparse regions coincide.)
struct TileGrid { int view, lead_index; bool is_lead; int k, wv; };
__device__ __forceinline__ TileGrid tile_grid(int lead_blocks_per_view) {
  TileGrid g;
  const int x = blockIdx.x, lead_rows = (lead_blocks_per_view + 31) >> 5;
  g.view = (int)blockIdx.z;
  g.is_lead = (int)blockIdx.y < lead_rows;
  g.lead_index = (int)blockIdx.y * 32 + (x & 31);
  g.k = ((int)blockIdx.y - lead_rows) * 8 + (x & 7);
  g.wv = (x >> 3) & 3;
  return g;
}
__host__ inline dim3 tile_grid_dim(const LOff& L, int nviews, int lead_blocks_per_view) {
  const int rows = (L.sgx * L.sgy + 7) / 8 + (lead_blocks_per_view + 31) / 32;
  return dim3(32, rows, nviews);
}
__device__ __forceinline__ bool tile_of_block(const TileGrid& g, const LOff& L, const char* saved, int& tx, int& ty) {
  // everything here is wave-uniform and must STAY in SGPRs (the tile's coordinates feed most of the kernel's address arithmetic):
  // the division by the super-tile row length is a multiply-high by a host-made reciprocal -- the compiler's integer division goes
  // through the VALU's float reciprocal and left tx / ty (and everything derived from them) in vector registers, ~35 instructions
  const int sgx = L.sgx, nsuper = L.sgx * L.sgy;
  const int k = g.k, wv = g.wv;
  int st;
  if (tile_order_used(L)) {
    if (k >= nsuper) return false;
    st = __builtin_amdgcn_readfirstlane((int)((const uint32_t*)(saved + L.o_tile_order))[k]);
  } else {
    // a light map (no launch order): every XCD keeps a BAND of the image, as in rounds 1-5 -- neighbouring tiles write neighbouring
    // partial slots of the Gaussians they share, and dealt to different XCDs those 48-byte slots share cache lines across L2s: the
    // dense backward that reads them took +3.5 us (light scene) with nothing gained in the tile kernel
    st = (k & 7) * ((nsuper + 7) >> 3) + (k >> 3);
    if (st >= nsuper) return false;
  }
  // st / sgx  (st * sgx < 2^32; an image of one super-tile column -- W <= 16 -- has no 32-bit reciprocal: 2^32 / 1 + 1 wraps to 1)
  const int row = sgx == 1 ? st : (int)__builtin_amdgcn_readfirstlane((int)__umulhi((uint32_t)st, L.sgx_magic));
  const int col = st - row * sgx;
  tx = __builtin_amdgcn_readfirstlane(col * 2 + (wv & 1));
  ty = __builtin_amdgcn_readfirstlane(row * 2 + (wv >> 1));
  return true;
}

template <int SORT_MAX, bool FUSED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SGR_TILE_WAVES))) blend_fwd_kernel(ViewTab tab, LOff L, const float* __restrict__ bg, LossTab lt,
                                                        LossCoef lc) {
  const TileGrid tg = tile_grid(comp_blocks(L));
  const int vw = tg.view;
  char* saved = tab.saved[vw];
  const int H = L.H, W = L.W, gx = L.gx, gy = L.gy;
  const int64_t cap = L.cap;
  const uint2* __restrict__ ranges = (const uint2*)(saved + L.o_ranges);
  uint64_t* __restrict__ entries = (uint64_t*)(tab.scratch[vw] + L.o_entries);
  uint32_t* __restrict__ point_list = (uint32_t*)(saved + L.o_point_list);
  const GRec* __restrict__ grec = grec_of(saved, L);
  float* __restrict__ out_color = tab.color[vw];
  float* __restrict__ out_depth = tab.depth[vw];
  float* __restrict__ out_opacity = tab.opacity[vw];
  float2* __restrict__ pix_state = (float2*)(saved + L.o_final_T);
  uint32_t* __restrict__ tile_maxc = (uint32_t*)(saved + L.o_tile_maxc);
  int32_t* __restrict__ n_touched = tab.n_touched[vw];
  extern __shared__ __attribute__((aligned(16))) char smem[];   // SORT_MAX sorted ids (4 B) or keys (8 B) + 64 splats x kStageBytes (+ FUSED: 2 x 64 float4 of pixel state)
  constexpr bool REGSORT = sort_in_registers(SORT_MAX);
  constexpr size_t kKeyBytes = REGSORT ? 4 : 8;
  // the first rows of the launch: the view's compact visible list (see above)
  if (tg.is_lead) { if (tg.lead_index < comp_blocks(L)) compact_visible_list(saved, L, tg.lead_index); return; }
  int tx, ty;
  if (!tile_of_block(tg, L, saved, tx, ty)) return;
  const int lane = threadIdx.x;
  constexpr int kLdsSortMax = SORT_MAX;
  char* slice = smem;
  if (tx >= gx || ty >= gy) return;          // whole wave outside the image
  const int tile = ty * gx + tx;
  const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  float4* lds = (float4*)slice;                              // 64 staged splats x kStageBytes
  uint64_t* keys = (uint64_t*)(slice + kWave * kStageBytes);          // mid / heavy build: the keys, sorted in place
  uint32_t* ids = (uint32_t*)(slice + kWave * kStageBytes);           // light build: sorted Gaussian indices (the keys were sorted in registers)
  // G stash (fused kernel, lists of <= kStash splats -- three quarters of a SLAM view's tiles): such a list is staged in the first
  // kStashOffset bytes and never uses the sorted-index area, so the 4.2 KB behind it hold exp2(-npow) of every (splat, pixel) pair the
  // forward walk evaluates (row = list position, kStashStride floats apart: rows two banks apart, conflict-free 8-byte reads of a
  // group's pixel pair), with 0 where the pixel is outside the splat's footprint (in_footprint()).  The backward reads it back instead
  // of re-evaluating the quadratic form, the exponential and the footprint test.  The same bits the backward would recompute: results
  // unchanged.
  float* stash = (float*)(slice + kStashOffset);
  static_assert(!FUSED || kStashOffset + kStash * kStashStride * 4 <= kWave * kStageBytes + SORT_MAX * kKeyBytes, "the stash overlays unused staging + index space");

  // ground truth of the fused loss epilogue: fetched NOW so that the round trip hides behind sorting and blending
  const float* __restrict__ gt_image = lt.gt_image[vw];
  float gt0 = 0.f, gt1 = 0.f, gt2 = 0.f, gtd = 0.f, ea = 1.f, eb = 0.f;
  if (gt_image && inside) {
    // uniform base (SGPR pair) + ONE 32-bit byte offset per lane: the planes are reached by moving the base, not the lane's address
    const uint32_t off = ((uint32_t)py * (uint32_t)W + (uint32_t)px) * 4u;
    const size_t plane = (size_t)H * (size_t)W * 4u;
    const char* g0 = (const char*)gt_image;
    gt0 = *(const float*)(g0 + off); gt1 = *(const float*)(g0 + plane + off); gt2 = *(const float*)(g0 + 2 * plane + off);
    gtd = *(const float*)((const char*)lt.gt_depth[vw] + off);
    ea = lt.exp_a[vw] ? __expf(lt.exp_a[vw][0]) : 1.f;
    eb = lt.exp_b[vw] ? lt.exp_b[vw][0] : 0.f;
  }

  // unsorted keys: the tile's bucket (filled by K1) unless the tile had more than kBucket pairs (then its exact run).
  // The bucket's address does not depend on the tile's range: lane j fetches bucket entry j (kBucket >= one wave) in the
  // same round trip as the range -- the first half of the bucket, which covers 98 % of the tiles of a SLAM view; the
  // rest follows once the count is known (entries behind the tile's count are ignored).
  const uint64_t* __restrict__ bucket = (const uint64_t*)(tab.scratch[vw] + L.o_bucket) + (size_t)tile * kBucket;
  static_assert(kBucket >= kWave, "the first 64 entries of a bucket are fetched one per lane");
  uint64_t key_spec = lane < 32 ? bucket[lane] : ~0ull;
  const uint2 rng = ranges[(size_t)tile * kRngStride];
  const int64_t begin = rng.x & ~kOverfull;
  const int64_t endc = (int64_t)rng.y < cap ? (int64_t)rng.y : cap;
  // A view whose pairs did not fit the workspace (header.overflow, set by K2) is not composited at all: its lists are incomplete --
  // keys that K1 could not append to the overflow list were never filed, so a run may hold whatever the block held before, and a
  // stale "Gaussian index" must not be dereferenced.  The view renders as background; every caller discards it (the gather passes
  // mask it out, the loops re-run it at a larger capacity).  One scalar load per wave.
  const uint32_t view_overflow = ((const SavedHeader*)(saved + L.o_hdr))->overflow;
  const int count = (endc > begin && view_overflow == 0u) ? (int)(endc - begin) : 0;
  const bool overfull = (rng.x & kOverfull) != 0;
  const uint64_t* __restrict__ keys_in = overfull ? entries + begin : bucket;

  // ---- sort this tile's run by (depth bits, Gaussian index) and publish the index list for the backward
  int mode = 0;                               // 0: registers, 1: LDS, 2: global
  if (count <= kWave) {
    // One chunk.  Every lane keeps ITS (unsorted) key and finds the key's position in the sorted list by counting
    // smaller keys (keys are unique), fetches the key's record and files it under that position.  The keys are broadcast through
    // LDS (the sorted-index area, which a one-chunk tile does not use): four per 16-byte read, so a key costs the compare and the
    // add -- two v_readlane per key through SGPRs were half of the loop's instructions on a unit that is the kernel's bottleneck.
    if (count > 32 && lane >= 32 && lane < count) key_spec = bucket[lane];
    const uint64_t key = lane < count ? key_spec : ~0ull;       // (ranks below kBucket are in the bucket whatever the tile's final count)
    const uint32_t g = (uint32_t)key;
    uint32_t rank = 0;
    {
      uint64_t* kl = (uint64_t*)ids;
      kl[lane] = key;                            // (lanes behind the list hold ~0: smaller than nothing)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int j = 0; j < count; j += 4) {
        const uint4 a = *(const uint4*)(kl + j), b = *(const uint4*)(kl + j + 2);
        rank += ((((uint64_t)a.y << 32) | a.x) < key ? 1u : 0u) + ((((uint64_t)a.w << 32) | a.z) < key ? 1u : 0u) +
                ((((uint64_t)b.y << 32) | b.x) < key ? 1u : 0u) + ((((uint64_t)b.w << 32) | b.z) < key ? 1u : 0u);
      }
      __builtin_amdgcn_wave_barrier();           // (the stash may overwrite the keys once the walk starts)
    }
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), co = m, cd = m;
    uint32_t slot = 0xffffffffu;
    if (lane < count) {
      const float4* rec = (const float4*)(grec + g);
      m = rec[0];
      co = rec[1];
      cd = rec[2];
      if (FUSED) slot = pair_slot(saved, L, g, ((const uint32_t*)(rec + 3))[1], __float_as_uint(m.z), __float_as_uint(m.w), tx, ty, cap);
    }
    if (lane >= count) rank = (uint32_t)lane;        // lane == count files the zero splat that pads an odd list
    if (!FUSED && lane < count) point_list[begin + rank] = g;      // (the fused backward reads the staged records instead)
    if (lane <= count) stage_splat(lds, rank, m, co, cd, g, slot);
  } else if (REGSORT && count <= kLdsSortMax) {
    mode = 1;                                 // 65..512 keys: sorted in registers, 2 / 4 / 8 per lane (count > 64: keys_in = the tile's run)
    if (count <= 2 * kWave) wave_sort_registers<2>(bucket, entries + begin, count, lane, ids);
    else if (count <= 4 * kWave) wave_sort_registers<4>(bucket, entries + begin, count, lane, ids);
    else wave_sort_registers<8>(bucket, entries + begin, count, lane, ids);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!FUSED)
      for (int i = lane; i < count; i += kWave) point_list[begin + i] = ids[i];
  } else if (count <= kLdsSortMax) {
    mode = 1;
    for (int i = lane; i < count; i += kWave) keys[i] = i < kBucket ? bucket[i] : keys_in[i];      // (count > 64: keys_in = the tile's run)
    __builtin_amdgcn_wave_barrier();
    wave_sort_any(count, lane, [&](int i) { return keys[i]; }, [&](int i, uint64_t v) { keys[i] = v; },
                  [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); });
    if (!FUSED)
      for (int i = lane; i < count; i += kWave) point_list[begin + i] = (uint32_t)keys[i];
  } else {
    mode = 2;                                 // slow path: in place in HBM through device-coherent accesses
    uint64_t* e = entries + begin;
    for (int i = lane; i < kBucket; i += kWave) __hip_atomic_store(e + i, bucket[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the bucket part joins its run)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();
    wave_sort_any(count, lane,
                  [&](int i) { return __hip_atomic_load(e + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
                  [&](int i, uint64_t v) { __hip_atomic_store(e + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); },
                  [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent"); __builtin_amdgcn_wave_barrier(); });
    if (!FUSED)
      for (int i = lane; i < count; i += kWave)
        point_list[begin + i] = (uint32_t)__hip_atomic_load(e + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  const bool use_stash = FUSED && mode == 0 && count <= kStash && !n_touched;
  float T = 1.f;
  v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f}, Dd = {0.f, 0.f};   // (even, odd) list positions, added at the end
  // Where this pixel's walk ENDS: the list position of the splat at which its transmittance would have fallen below kTEps (that splat
  // is not composited), 0x7fffffff while it has not.  That is all the backward needs to know (`idx < nc`): every splat in front of
  // that position that passes the footprint test WAS composited, so the backward's own footprint test singles out the contributors.
  // Rounds 1-5 tracked the last CONTRIBUTOR instead -- a move and a select per trip for every pixel; a pixel terminates at most once,
  // and which pixels do in a trip is a wave mask the termination logic has anyway: the select now sits behind a scalar branch that a
  // fresh map's tiles never take.
  uint32_t term = inside ? 0x7fffffffu : 0u;
  uint32_t term_last = 0;                     // position of the latest termination in the tile (monotonic; scalar)
  unsigned long long done_m = __builtin_amdgcn_ballot_w64(!inside || (L.dbg & 2048));      // finished pixels, one bit per lane (bit 11 of SGR_DEBUG, EXPERIMENT: no walk, no backward -- what the rest of a tile's wave costs)
  const v2f px2 = splat2(pxf), py2 = splat2(pyf);

  for (int base = 0; base < count; base += kWave) {
    const int n = min(kWave, count - base);
    // gather this chunk: lane j fetches splat j (one 64-byte record: centre | conic, opacity | colour, depth) and stores
    // it PAIR-INTERLEAVED (splats 2p, 2p+1 side by side, 6 float4 per pair) so that the walk reads 2-vectors;
    // an odd chunk is padded with a splat of opacity 0 (never contributes)
    if (mode != 0 && lane <= n) {
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f), co = m, cd = m;
      uint32_t g = 0;
      if (lane < n) {
        if (mode == 1) g = REGSORT ? ids[base + lane] : (uint32_t)keys[base + lane];
        else g = (uint32_t)__hip_atomic_load(entries + begin + base + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float4* rec = (const float4*)(grec + g);
        m = rec[0];
        co = rec[1];
        cd = rec[2];
      }
      stage_splat(lds, (uint32_t)lane, m, co, cd, g, 0xffffffffu);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Two splats per trip, branch free.  blend_fwd is VALU-issue bound like the backward, so the footprint (the same
    // operation sequence, packed), alpha * T and the four accumulate FMAs issue once per PAIR (v_pk_*_f32); exp, the
    // tests and the transmittance chain stay per splat.  The "every pixel finished" exit is polled every 4 splats.
    auto walk = [&](auto count_touched, auto stash_g) {
#pragma clang fp contract(off)      // the two instantiations must round alike (T * (1 - alpha) is not to become an fma in one)
      for (int j = 0; j < n; j += 2) {
        if ((j & 3) == 0 && ~done_m == 0ull) break;
        const float4* e = lds + (j >> 1) * (kStageFloats / 2);
        const float4 q0 = e[0], q1 = e[1], q2 = e[2], q3 = e[3], q4 = e[4], q5 = e[5];
        const v2f dx = (v2f){q0.x, q0.y} - px2, dy = (v2f){q0.z, q0.w} - py2;
        // npow = dx (A' dx + B' dy) + (C' dy) dy  (stage_shape(); the backward evaluates the same operation sequence)
        const v2f u = __builtin_elementwise_fma((v2f){q1.x, q1.y}, dx, (v2f){q1.z, q1.w} * dy);
        const v2f npow = __builtin_elementwise_fma(u, dx, ((v2f){q2.x, q2.y} * dy) * dy);
        const v2f G = {exp2_fast(-npow.x), exp2_fast(-npow.y)};
        // Who contributes is decided on wave masks in SGPRs (every lane of the wave is active here: a ballot is the whole comparison):
        // the and / andn2 / or of the termination logic are scalar instructions.  ONE integer compare per (pixel, splat): in_footprint()
        const unsigned long long ok0 = __builtin_amdgcn_ballot_w64(in_footprint(npow.x, q5.z));
        const unsigned long long ok1 = __builtin_amdgcn_ballot_w64(in_footprint(npow.y, q5.w));
        if (decltype(stash_g)::value) {
          float* sp = stash + j * kStashStride + lane;
          sp[0] = __builtin_amdgcn_inverse_ballot_w64(ok0) ? G.x : 0.f;
          sp[kStashStride] = __builtin_amdgcn_inverse_ballot_w64(ok1) ? G.y : 0.f;
        }
        const v2f og = (v2f){q2.z, q2.w} * G;
        const v2f alpha = {fminf(kAlphaMax, og.x), fminf(kAlphaMax, og.y)};
        const v2f one_m = splat2(1.f) - alpha;
        const float test0 = T * one_m.x;
        const unsigned long long live0 = ok0 & ~done_m, lt0 = __builtin_amdgcn_ballot_w64(test0 < kTEps);
        const unsigned long long comp0_m = live0 & ~lt0;
        done_m |= live0 & lt0;
        const bool comp0 = __builtin_amdgcn_inverse_ballot_w64(comp0_m);
        const float T1 = comp0 ? test0 : T;
        const float test1 = T1 * one_m.y;
        const unsigned long long live1 = ok1 & ~done_m, lt1 = __builtin_amdgcn_ballot_w64(test1 < kTEps);
        const unsigned long long comp1_m = live1 & ~lt1;
        done_m |= live1 & lt1;
        const bool comp1 = __builtin_amdgcn_inverse_ballot_w64(comp1_m);
        v2f w = alpha * (v2f){T, T1};
        w.x = comp0 ? w.x : 0.f;
        w.y = comp1 ? w.y : 0.f;
        Cr = __builtin_elementwise_fma((v2f){q4.x, q4.y}, w, Cr);
        Cg = __builtin_elementwise_fma((v2f){q4.z, q4.w}, w, Cg);
        Cb = __builtin_elementwise_fma((v2f){q5.x, q5.y}, w, Cb);
        Dd = __builtin_elementwise_fma((v2f){q3.x, q3.y}, w, Dd);
        if (decltype(count_touched)::value) {
          unsigned long long tm = comp0_m & __builtin_amdgcn_ballot_w64(test0 > kTouchedT);
          if (tm != 0ull && lane == 0) atomicAdd(&n_touched[__float_as_uint(q3.z)], (int)__popcll(tm));
          tm = comp1_m & __builtin_amdgcn_ballot_w64(test1 > kTouchedT);
          if (tm != 0ull && lane == 0) atomicAdd(&n_touched[__float_as_uint(q3.w)], (int)__popcll(tm));
        }
        T = comp1 ? test1 : T1;
        const unsigned long long end0_m = live0 & lt0, end1_m = live1 & lt1;
        if ((end0_m | end1_m) != 0ull) {
          asm volatile("" ::: "memory");            // (a real branch: speculated, the selects would be back in every trip)
          const uint32_t p0 = (uint32_t)(base + j);
          term = __builtin_amdgcn_inverse_ballot_w64(end0_m) ? p0 : (__builtin_amdgcn_inverse_ballot_w64(end1_m) ? p0 + 1u : term);
          term_last = end1_m != 0ull ? p0 + 1u : p0;
        }
      }
    };
    if (n_touched) walk(std::true_type{}, std::false_type{});
    else if (use_stash) walk(std::false_type{}, std::true_type{});
    else walk(std::false_type{}, std::false_type{});
    __builtin_amdgcn_wave_barrier();
    if (~done_m == 0ull) break;
  }
  const float C0 = Cr.x + Cr.y, C1 = Cg.x + Cg.y, C2 = Cb.x + Cb.y, D = Dd.x + Dd.y;
  const uint32_t last = term;

  // per-tile bound for the backward: when every pixel's walk has ended, nothing behind the latest end was composited anywhere
  const uint32_t mx = (~done_m == 0ull) ? term_last : (uint32_t)count;
  if (lane == 0) tile_maxc[tile] = mx;

  float l_rgb = 0.f, l_dep = 0.f, l_da = 0.f, l_db = 0.f;
  const uint32_t tpix = (uint32_t)tile * 64u + (uint32_t)lane;
  if (!FUSED) pix_state[tpix] = make_float2(T, __uint_as_float(last));       // (lanes outside the image: T = 1, no contributor)
  uint32_t code = 0;
  float pxA[4] = {0.f, 0.f, 0.f, 0.f};        // FUSED: dL/dC (r, g, b) and dL/dD of this pixel, straight from the residuals' signs
  if (inside) {
    const uint32_t pix = (uint32_t)py * (uint32_t)W + (uint32_t)px, hw = (uint32_t)H * (uint32_t)W;   // 32-bit: uniform base + lane offset
    const float I[3] = {C0 + T * bg[0], C1 + T * bg[1], C2 + T * bg[2]};
    if (out_color) {            // (a training iteration that only needs the loss passes no image buffers)
      out_color[pix] = I[0];
      out_color[hw + pix] = I[1];
      out_color[2 * hw + pix] = I[2];
      out_depth[pix] = D;
      out_opacity[pix] = 1.f - T;
    }
    if (gt_image) {
      // fused mapping loss (slam_utils.py:71-105): this pixel's residuals, the gradients the backward consumes, and
      // its share of the four sums (|rgb|, |depth|, d/da, d/db)
      // The L1 gradients are +-constant or 0 per value: dL/dC_c = sign * (w_rgb * e^a), dL/dD = sign * w_dep.  The backward
      // gets ONE code byte per pixel (2 bits per value: 0, 1 = +, 2 = -) in the first H*W bytes of the view's dL_dimage
      // scratch and rebuilds the same floats (16 -> 1 byte per pixel written here and read there).
      const float g[3] = {gt0, gt1, gt2};
      const bool m = (g[0] + g[1] + g[2]) > lc.thr;
      const float k_rgb = lc.w_rgb * ea;           // (the float blend_bwd<true> rebuilds from the code byte)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float r = m ? (ea * I[c] + eb) - g[c] : 0.f;
        l_rgb += fabsf(r);
        const bool pos = r > 0.f, neg = r < 0.f;
        if (!FUSED) code |= (pos ? 1u : (neg ? 2u : 0u)) << (2 * c);
        // dL/dC_c = sign(r) * w * e^a and its share of d/da, d/db: selects and ONE explicit fma, so that the fused and the un-fused
        // instantiation cannot round differently ((w * sgn) * e^a = sgn * (w * e^a) exactly: sgn is 0 or +-1)
        const float dab = pos ? lc.w_rgb : (neg ? -lc.w_rgb : 0.f), dk = pos ? k_rgb : (neg ? -k_rgb : 0.f);
        if (FUSED) pxA[c] = dk;
        l_da = __fmaf_rn(dk, I[c], l_da);
        l_db += dab;
      }
      const float gd = gtd;
      const float rd = (gd > 0.01f) ? D - gd : 0.f;
      l_dep = fabsf(rd);
      if (FUSED) pxA[3] = (rd > 0.f) ? lc.w_dep : ((rd < 0.f) ? -lc.w_dep : 0.f);
      else code |= ((rd > 0.f) ? 1u : ((rd < 0.f) ? 2u : 0u)) << 6;
      if (!FUSED && !code_bytes_tiled(L)) ((uint8_t*)lt.dimage[vw])[pix] = (uint8_t)code;
    }
  }
  if (!FUSED && gt_image && code_bytes_tiled(L)) ((uint8_t*)lt.dimage[vw])[tpix] = (uint8_t)code;
  if (gt_image) {      // uniform per view
    const float tot = wave_sum4_rows(l_rgb, l_dep, l_da, l_db);          // rows: |rgb|, d/da, |depth|, d/db
    const int row = lane >> 4;
    if ((lane & 15) == 0) ((float*)((LossPart*)lt.parts[vw] + tile))[row == 1 ? 2 : (row == 2 ? 1 : row)] = tot;
  }
  if (!FUSED) return;

  // ---- the backward of this tile, right here
  if (count == 0) return;
  const int eff = (L.dbg & 4096) ? 0 : min(count, (int)mx);      // (bit 12, EXPERIMENT: no backward)
  float4* __restrict__ partials = (float4*)(tab.scratch[vw] + L.o_partials);
  // pairs the walk never reached (behind every pixel's last contributor) still own a slot: define it as zero
  float pxB[3] = {T, 0.f, __uint_as_float(last)};
  float4* pixA = (float4*)(slice + (size_t)SORT_MAX * kKeyBytes + kWave * kStageBytes);
  float4* pixB = pixA + kWave;
  if (mode == 0) {
    const SrcStaged src = {(const float*)lds};
    for (int idx = eff + lane; idx < count; idx += kWave) {
      const uint32_t slot = src.load(idx, L).slot;
      if (slot != 0xffffffffu) {
        partials[(size_t)slot * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        partials[(size_t)slot * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        partials[(size_t)slot * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (eff == 0) return;
    tile_backward(lane, eff, tx, ty, pxA, pxB, bg, pixA, pixB, src, L, partials, use_stash ? stash : nullptr);
  } else {
    const SrcKeys src = {mode == 1 && REGSORT ? ids : nullptr, mode == 1 && !REGSORT ? keys : nullptr, entries + begin, grec, saved, tx, ty, cap};
    for (int idx = eff + lane; idx < count; idx += kWave) {
      const uint32_t g = src.gaussian(idx);
      const float4 q0 = ((const float4*)(grec + g))[0];
      const uint32_t slot = pair_slot(saved, L, g, grec[g].offset, __float_as_uint(q0.z), __float_as_uint(q0.w), tx, ty, cap);
      if (slot != 0xffffffffu) {
        partials[(size_t)slot * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        partials[(size_t)slot * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        partials[(size_t)slot * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (eff == 0) return;
    tile_backward(lane, eff, tx, ty, pxA, pxB, bg, pixA, pixB, src, L, partials);
  }
}

template <bool PACKED>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SGR_BWD_WAVES))) blend_bwd_kernel(ViewTab tab, LOff L, const float* __restrict__ bg, SignGrad sg) {
  const TileGrid tg = tile_grid(0);
  const int vw = tg.view;
  const char* saved = tab.saved[vw];
  const int H = L.H, W = L.W, gx = L.gx, gy = L.gy;
  const int64_t cap = L.cap;
  const uint2* __restrict__ ranges = (const uint2*)(saved + L.o_ranges);
  const uint32_t* __restrict__ point_list = (const uint32_t*)(saved + L.o_point_list);
  const GRec* __restrict__ grec = grec_of(saved, L);
  const float2* __restrict__ pix_state = (const float2*)(saved + L.o_final_T);
  const uint32_t* __restrict__ tile_maxc = (const uint32_t*)(saved + L.o_tile_maxc);
  const float* __restrict__ dL_dcolor = tab.dL_dcolor[vw];
  const float* __restrict__ dL_ddepth = tab.dL_ddepth[vw];
  float4* __restrict__ partials = (float4*)(tab.scratch[vw] + L.o_partials);
  __shared__ float4 pixbuf[2][kWave];         // pixel gradients + running (T, S) carries
  int tx, ty;
  if (!tile_of_block(tg, L, saved, tx, ty)) return;
  const int lane = threadIdx.x;
  if (tx >= gx || ty >= gy) return;
  const int tile = ty * gx + tx;
  // this pixel's state first: its six loads do not depend on the tile's list and overlap the range / list round trips
  float pxA[4] = {0.f, 0.f, 0.f, 0.f}, pxB[3] = {1.f, 0.f, 0.f};
  {
    const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
    const bool inside = px < W && py < H;
    const uint32_t pix = (uint32_t)py * (uint32_t)W + (uint32_t)px, hw = (uint32_t)H * (uint32_t)W;   // 32-bit: uniform base + lane offset
    const uint32_t tpix = (uint32_t)tile * 64u + (uint32_t)lane;
    if (PACKED) {
      uint32_t code = 0;
      if (code_bytes_tiled(L)) code = ((const uint8_t*)dL_dcolor)[tpix];       // (outside the image: 0)
      else if (inside) code = ((const uint8_t*)dL_dcolor)[pix];
      const float k_rgb = sg.w_rgb * (sg.exp_a[vw] ? __expf(sg.exp_a[vw][0]) : 1.f);
      pxA[0] = sign_code(code, k_rgb); pxA[1] = sign_code(code >> 2, k_rgb); pxA[2] = sign_code(code >> 4, k_rgb);
      pxA[3] = sign_code(code >> 6, sg.w_dep);
    } else if (inside) {
      pxA[0] = dL_dcolor[pix]; pxA[1] = dL_dcolor[hw + pix]; pxA[2] = dL_dcolor[2 * hw + pix];
      pxA[3] = dL_ddepth ? dL_ddepth[pix] : 0.f;
    }
    const float2 ps = pix_state[tpix];                 // (outside the image the forward left T = 1, no contributor)
    pxB[0] = ps.x;
    pxB[2] = ps.y;
  }
  const uint2 rng = ranges[(size_t)tile * kRngStride];
  const int64_t begin = rng.x & ~kOverfull;
  const int64_t endc = (int64_t)rng.y < cap ? (int64_t)rng.y : cap;
  const int count = endc > begin ? (int)(endc - begin) : 0;
  if (count == 0 || ((const SavedHeader*)(saved + L.o_hdr))->overflow != 0u) return;      // (a truncated view: see blend_fwd_kernel)
  const int eff = min(count, (int)tile_maxc[tile]);

  // pairs the forward never reached (behind every pixel's last contributor) still own a slot: define it as zero
  for (int idx = eff + lane; idx < count; idx += kWave) {
    uint32_t g = point_list[begin + idx];
    const float4 q0 = ((const float4*)(grec + g))[0];
    const uint32_t slot = pair_slot(saved, L, g, grec[g].offset, __float_as_uint(q0.z), __float_as_uint(q0.w), tx, ty, cap);
    if (slot != 0xffffffffu) {
      partials[(size_t)slot * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
      partials[(size_t)slot * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
      partials[(size_t)slot * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (eff == 0) return;
  const SrcPointList src = {point_list, begin, grec, saved, tx, ty, cap};
  tile_backward<SrcPointList, SGR_BWD_KPR64>(lane, eff, tx, ty, pxA, pxB, bg, pixbuf[0], pixbuf[1], src, L, partials);
}

template <int SORT_MAX, bool FUSED>
static void launch_blend_fwd_t(const ViewTab& tab, int nviews, const LOff& L, const float* bg, const LossTab& lt,
                               const LossCoef& lc, hipStream_t st) {
  const dim3 grid = tile_grid_dim(L, nviews, (((L.nseg + kCompSegs - 1) / kCompSegs) + 7) & ~7);
  constexpr size_t lds = (size_t)SORT_MAX * (sort_in_registers(SORT_MAX) ? 4 : 8) + kWave * kStageBytes + (FUSED ? 2 * kWave * 16 : 0);
  static bool attr_set = false;
  if (lds > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute((const void*)blend_fwd_kernel<SORT_MAX, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((blend_fwd_kernel<SORT_MAX, FUSED>), grid, dim3(kWave), lds, st, tab, L, bg, lt, lc);
}

// 0 light / 1 mid / 2 heavy from the longest list the caller has MEASURED for the cameras of the batch (max_list_hint =
// SavedHeader.max_tile_count of a recent forward): a deterministic function of a measurement, where a guess from the
// capacity -- which depends on the probe history of the caller -- flipped the build between two runs of the same scene.
// Without a hint the longest list is taken as 4x the mean the capacity allows for (capacity is ~2x the pair count).
// Tiles beyond the chosen build's key count fall back to the in-HBM sort: correct, slow; the light build (the one that runs
// forward and backward of a tile in ONE wave) is kept while the longest list exceeds its 512 keys by no more than a few
// tiles' worth (x1.5), because fusing pays on long lists too (opaque bench scene 1.36 -> 1.17 ms per iteration).
static int blend_build(const LOff& L) {
  const int64_t longest = L.mean_hint > 0 ? (int64_t)L.mean_hint : 2 * L.cap / (int64_t)(L.ntiles > 0 ? L.ntiles : 1);
  return longest > 3 * kSortMid / 2 ? 2 : (longest > 3 * kSortLight / 2 ? 1 : 0);
}

// lt: per-view loss pointers (gt_image[v] == NULL -> plain render).  With a loss, every 8x8 tile also leaves one
// LossPart in lt.parts[v][tile]; launch_mapping_loss_final adds them up in fixed order.
void launch_blend_fwd(const ViewTab& tab, int nviews, const LOff& L, const float* bg, const LossTab* lt, const LossCoef* lc,
                      hipStream_t st) {
  ProfScope prof(PK_BLEND_FWD, st);
  LossTab none = {};
  LossCoef nocoef = {0.f, 0.f, 0.f};
  const LossTab& t = lt ? *lt : none;
  const LossCoef& c = lc ? *lc : nocoef;
  const int build = blend_build(L);
  if (build == 2) launch_blend_fwd_t<kSortHeavy, false>(tab, nviews, L, bg, t, c, st);
  else if (build == 1) launch_blend_fwd_t<kSortMid, false>(tab, nviews, L, bg, t, c, st);
  else launch_blend_fwd_t<kSortLight, false>(tab, nviews, L, bg, t, c, st);
}

// The fused tile kernel exists in the light build only: with thousands of keys per wave in LDS one or three workgroups fit a
// CU, and a backward that runs at that occupancy is slower than blend_bwd_kernel on its own (measured on the opaque room,
// mean list 93: fused heavy 2.80 ms vs 1.33 + 0.70 ms for the pair) -- long-list maps run the two kernels.
bool blend_can_fuse(const LOff& L) { return blend_build(L) == 0; }

// forward + mapping loss + backward of every tile in ONE launch (needs lt / lc: the loss is what links the two halves)
void launch_blend_fused(const ViewTab& tab, int nviews, const LOff& L, const float* bg, const LossTab& lt, const LossCoef& lc,
                        hipStream_t st) {
  ProfScope prof(PK_BLEND_FUSED, st);
  launch_blend_fwd_t<kSortLight, true>(tab, nviews, L, bg, lt, lc, st);
}

// lt / lc given: the pixel gradients are the code bytes blend_fwd's loss epilogue wrote for these views
void launch_blend_bwd(const ViewTab& tab, int nviews, const LOff& L, const float* bg, const LossTab* lt, const LossCoef* lc,
                      hipStream_t st) {
  const dim3 grid = tile_grid_dim(L, nviews, 0);
  ProfScope prof(PK_BLEND_BWD, st);
  SignGrad sg = {};
  if (lt && lc) {
    for (int v = 0; v < nviews; ++v) sg.exp_a[v] = lt->exp_a[v];
    sg.w_rgb = lc->w_rgb; sg.w_dep = lc->w_dep;
    hipLaunchKernelGGL(blend_bwd_kernel<true>, grid, dim3(kWave), 0, st, tab, L, bg, sg);
  } else {
    hipLaunchKernelGGL(blend_bwd_kernel<false>, grid, dim3(kWave), 0, st, tab, L, bg, sg);
  }
}

}  // namespace sgr
