// C-ABI entry points of the rasterizer + the binning stage (scan / duplicate / sort / ranges).
// Public contract: include/splat_hip.h.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "sgr_common.h"

namespace sgr {

void launch_preprocess_fwd(const SgrSettings&, const SgrInputs&, const SgrOutputs&, const Layout&, char*, hipStream_t);
void launch_preprocess_bwd(const SgrSettings&, const SgrInputs&, const int32_t*, const SgrGradInputs&, const Layout&,
                           const char*, char*, hipStream_t);
void launch_blend_fwd(const SgrSettings&, const SgrOutputs&, const Layout&, char*, hipStream_t);
void launch_blend_bwd(const SgrSettings&, const SgrGradOutputs&, const Layout&, const char*, char*, hipStream_t);

static thread_local char g_err[512] = "";
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return set_error(SGR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// ---- rocprim temp sizes are pure functions of the element count: query once per (N, cap)
static size_t scan_temp_bytes(int N) {
  size_t bytes = 0;
  if (N <= 0) return 256;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)N,
                                rocprim::plus<uint32_t>(), (hipStream_t)0);
  return bytes + 256;
}
static size_t sort_temp_bytes(int64_t cap) {
  size_t bytes = 0;
  if (cap <= 0) return 256;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (size_t)cap, 0u, 64u, (hipStream_t)0);
  return bytes + 256;
}
static Layout make_layout(int N, int H, int W, int64_t cap) {
  return Layout(N, H, W, cap, scan_temp_bytes(N), sort_temp_bytes(cap));
}

// ---- binning kernels
__global__ void finalize_count_kernel(int N, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ touched,
                                      int64_t cap, SavedHeader* __restrict__ hdr) {
  uint64_t R = N > 0 ? (uint64_t)offsets[N - 1] + touched[N - 1] : 0;
  hdr->num_rendered = (uint32_t)R;
  hdr->overflow = (int64_t)R > cap ? 1u : 0u;
  hdr->sorted_count = (uint32_t)((int64_t)R > cap ? cap : (int64_t)R);
}

// one thread per Gaussian emits a (tile | depth) key for every bin of its rectangle, at the slot that the backward
// will later use for that pair's gradient partial
__global__ void __launch_bounds__(256) duplicate_keys_kernel(int N, int gx, const int32_t* __restrict__ radii,
                                                             const uint32_t* __restrict__ offsets,
                                                             const uint32_t* __restrict__ touched,
                                                             const ushort4* __restrict__ rect,
                                                             const float4* __restrict__ rgbd, int64_t cap,
                                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || radii[i] <= 0 || touched[i] == 0) return;
  ushort4 r = rect[i];
  uint64_t off = offsets[i];
  uint32_t dbits = __float_as_uint(rgbd[i].w);
  for (int y = r.y; y < r.w; ++y)
    for (int x = r.x; x < r.z; ++x) {
      if ((int64_t)off < cap) {
        keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
        vals[off] = (uint32_t)i;
      }
      ++off;
    }
}

// async mode sorts the whole capacity: park the unused tail behind every real tile
__global__ void __launch_bounds__(256) fill_sentinel_kernel(const SavedHeader* __restrict__ hdr, int64_t cap,
                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  int64_t first = hdr->sorted_count;
  for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = ~0ull;
    vals[i] = 0;
  }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(const SavedHeader* __restrict__ hdr,
                                                          const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
  int64_t n = hdr->sorted_count;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t t = (uint32_t)(keys[i] >> 32);
    if (i == 0 || (uint32_t)(keys[i - 1] >> 32) != t) ranges[t].x = (uint32_t)i;
    if (i == n - 1 || (uint32_t)(keys[i + 1] >> 32) != t) ranges[t].y = (uint32_t)(i + 1);
  }
}

// ---- event-pair profiler
struct ProfState {
  static constexpr int kRing = 2048;
  uint32_t mask = 0;
  hipEvent_t ev[SGR_PROFILE_KINDS][kRing][2];
  bool created[SGR_PROFILE_KINDS] = {};
  int used[SGR_PROFILE_KINDS] = {};
  double ms[SGR_PROFILE_KINDS] = {};
  int64_t launches[SGR_PROFILE_KINDS] = {};
};
static ProfState g_prof;
static void prof_drain(int k) {
  for (int i = 0; i < g_prof.used[k]; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof.ev[k][i][1]) == hipSuccess &&
        hipEventElapsedTime(&t, g_prof.ev[k][i][0], g_prof.ev[k][i][1]) == hipSuccess) {
      g_prof.ms[k] += t;
      g_prof.launches[k] += 1;
    }
  }
  g_prof.used[k] = 0;
}
void prof_begin(int k, hipStream_t st) {
  if (!(g_prof.mask & (1u << k))) return;
  if (!g_prof.created[k]) {
    for (int i = 0; i < ProfState::kRing; ++i) { (void)hipEventCreate(&g_prof.ev[k][i][0]); (void)hipEventCreate(&g_prof.ev[k][i][1]); }
    g_prof.created[k] = true;
  }
  if (g_prof.used[k] == ProfState::kRing) prof_drain(k);
  (void)hipEventRecord(g_prof.ev[k][g_prof.used[k]][0], st);
}
void prof_end(int k, hipStream_t st) {
  if (!(g_prof.mask & (1u << k))) return;
  (void)hipEventRecord(g_prof.ev[k][g_prof.used[k]][1], st);
  g_prof.used[k] += 1;
}

__global__ void __launch_bounds__(256) stats_kernel(int N, int ntiles, const int32_t* __restrict__ radii,
                                                     const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_maxc,
                                                     unsigned long long* __restrict__ out) {
  unsigned long long v = 0, r = 0, re = 0, ne = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) v += radii[i] > 0;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
    uint32_t c = ranges[t].y - ranges[t].x;
    r += c; re += min(c, tile_maxc[t]); ne += c > 0;
  }
  atomicAdd(&out[0], v); atomicAdd(&out[1], r); atomicAdd(&out[2], re); atomicAdd(&out[3], ne);
}

static int check_common(const SgrSettings* s, const SgrWorkspace* ws, const Layout& L) {
  if (!ws->saved || ws->saved_bytes < L.saved_bytes)
    return set_error(SGR_ERR_WORKSPACE, "saved workspace too small: %zu < %zu", ws->saved_bytes, L.saved_bytes);
  if (!ws->scratch || ws->scratch_bytes < L.scratch_bytes)
    return set_error(SGR_ERR_WORKSPACE, "scratch workspace too small: %zu < %zu", ws->scratch_bytes, L.scratch_bytes);
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->projmatrix_raw || !s->campos)
    return set_error(SGR_ERR_INVALID, "settings: bg/viewmatrix/projmatrix/projmatrix_raw/campos must be device pointers");
  return SGR_OK;
}

}  // namespace sgr

using namespace sgr;

extern "C" {

int sgr_abi_version(void) { return SGR_ABI_VERSION; }
const char* sgr_last_error(void) { return g_err; }

size_t sgr_saved_bytes(int32_t N, int32_t H, int32_t W, int64_t cap) { return make_layout(N, H, W, cap).saved_bytes; }
size_t sgr_scratch_bytes(int32_t N, int32_t H, int32_t W, int64_t cap) { return make_layout(N, H, W, cap).scratch_bytes; }

int sgr_forward(const SgrSettings* s, const SgrInputs* in, const SgrOutputs* out, const SgrWorkspace* ws,
                int64_t* num_rendered_host, void* stream) {
  if (!s || !in || !out || !ws) return set_error(SGR_ERR_INVALID, "null argument");
  const int N = s->num_gaussians, H = s->image_height, W = s->image_width;
  if (N < 0 || H <= 0 || W <= 0 || ws->capacity <= 0) return set_error(SGR_ERR_INVALID, "bad sizes N=%d H=%d W=%d cap=%lld", N, H, W, (long long)ws->capacity);
  if (N > 0) {
    if (!in->means3D || !in->opacities) return set_error(SGR_ERR_INVALID, "means3D and opacities are required");
    if ((in->shs != nullptr) == (in->colors_precomp != nullptr))
      return set_error(SGR_ERR_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    if (((in->scales != nullptr) && (in->rotations != nullptr)) == (in->cov3D_precomp != nullptr) ||
        ((in->scales != nullptr) != (in->rotations != nullptr)))
      return set_error(SGR_ERR_INVALID, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (in->shs && (s->sh_degree < 0 || s->sh_degree > 3 || s->sh_coeffs < (s->sh_degree + 1) * (s->sh_degree + 1)))
      return set_error(SGR_ERR_INVALID, "sh_degree %d needs %d coefficients, shs has %d", s->sh_degree, (s->sh_degree + 1) * (s->sh_degree + 1), s->sh_coeffs);
  }
  if (!out->color || !out->depth || !out->opacity || (N > 0 && (!out->radii || !out->n_touched)))
    return set_error(SGR_ERR_INVALID, "null output pointer");
  Layout L = make_layout(N, H, W, ws->capacity);
  if (int rc = check_common(s, ws, L)) return rc;
  hipStream_t st = (hipStream_t)stream;
  char* saved = (char*)ws->saved;
  char* scratch = (char*)ws->scratch;
  SavedHeader* hdr = (SavedHeader*)(saved + L.o_hdr);
  uint32_t* offsets = (uint32_t*)(saved + L.o_offsets);
  uint32_t* touched = (uint32_t*)(saved + L.o_touched);
  uint64_t* keys_in = (uint64_t*)(scratch + L.o_keys_in);
  uint64_t* keys_out = (uint64_t*)(scratch + L.o_keys_out);
  uint32_t* vals_in = (uint32_t*)(scratch + L.o_vals_in);
  uint32_t* point_list = (uint32_t*)(saved + L.o_point_list);

  HIP_TRY(hipMemsetAsync(saved + L.o_ranges, 0, (size_t)L.ntiles * 8, st));
  if (N > 0) HIP_TRY(hipMemsetAsync(out->n_touched, 0, (size_t)N * 4, st));
  launch_preprocess_fwd(*s, *in, *out, L, saved, st);
  if (N > 0) {
    ProfScope prof(PK_SCAN, st);
    size_t tb = L.scan_tmp_bytes;
    HIP_TRY(rocprim::exclusive_scan(scratch + L.o_scan_tmp, tb, touched, offsets, 0u, (size_t)N, rocprim::plus<uint32_t>(), st));
  }
  hipLaunchKernelGGL(finalize_count_kernel, dim3(1), dim3(1), 0, st, N, offsets, touched, L.cap, hdr);

  int64_t sort_n = L.cap;
  if (num_rendered_host) {
    uint32_t R = 0;
    HIP_TRY(hipMemcpyAsync(&R, &hdr->num_rendered, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *num_rendered_host = R;
    if ((int64_t)R > L.cap) return set_error(SGR_ERR_CAPACITY, "%u (tile, Gaussian) pairs exceed capacity %lld", R, (long long)L.cap);
    sort_n = R;
  }
  if (N > 0 && sort_n > 0) {
    {
      ProfScope prof(PK_DUP, st);
      hipLaunchKernelGGL(duplicate_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, L.gx, out->radii, offsets,
                         touched, (const ushort4*)(saved + L.o_rect), (const float4*)(saved + L.o_rgbd), L.cap, keys_in, vals_in);
      if (!num_rendered_host)
        hipLaunchKernelGGL(fill_sentinel_kernel, dim3(256), dim3(256), 0, st, hdr, L.cap, keys_in, vals_in);
    }
    {
      ProfScope prof(PK_SORT, st);
      size_t tb = L.sort_tmp_bytes;
      HIP_TRY(rocprim::radix_sort_pairs(scratch + L.o_sort_tmp, tb, keys_in, keys_out, vals_in, point_list, (size_t)sort_n, 0u,
                                        (unsigned)(32 + L.tile_bits), st));
    }
    ProfScope prof(PK_RANGES, st);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(256), dim3(256), 0, st, hdr, keys_out, (uint2*)(saved + L.o_ranges));
  }
  launch_blend_fwd(*s, *out, L, saved, st);
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

int sgr_backward(const SgrSettings* s, const SgrInputs* in, const int32_t* radii, const SgrGradOutputs* go,
                 const SgrGradInputs* gi, const SgrWorkspace* ws, void* stream) {
  if (!s || !in || !go || !gi || !ws) return set_error(SGR_ERR_INVALID, "null argument");
  const int N = s->num_gaussians, H = s->image_height, W = s->image_width;
  if (N < 0 || H <= 0 || W <= 0 || ws->capacity <= 0) return set_error(SGR_ERR_INVALID, "bad sizes");
  if (!go->dL_dcolor) return set_error(SGR_ERR_INVALID, "dL_dcolor is required");
  if (N > 0 && !radii) return set_error(SGR_ERR_INVALID, "radii is required");
  if (gi->stat_grad_accum && (!gi->stat_denom || !gi->stat_max_radii))
    return set_error(SGR_ERR_INVALID, "stat_grad_accum needs stat_denom and stat_max_radii");
  Layout L = make_layout(N, H, W, ws->capacity);
  if (int rc = check_common(s, ws, L)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (gi->dL_dtau) HIP_TRY(hipMemsetAsync(gi->dL_dtau, 0, 24, st));
    return SGR_OK;
  }
  launch_blend_bwd(*s, *go, L, (const char*)ws->saved, (char*)ws->scratch, st);
  launch_preprocess_bwd(*s, *in, radii, *gi, L, (const char*)ws->saved, (char*)ws->scratch, st);
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

int sgr_query_stats(const SgrWorkspace* ws, int32_t N, int32_t H, int32_t W, const int32_t* radii, int64_t stats_host[4],
                    void* stream) {
  if (!ws || !ws->saved || !ws->scratch || !stats_host) return set_error(SGR_ERR_INVALID, "null argument");
  Layout L = make_layout(N, H, W, ws->capacity);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* d = (unsigned long long*)ws->scratch;
  HIP_TRY(hipMemsetAsync(d, 0, 32, st));
  hipLaunchKernelGGL(stats_kernel, dim3(64), dim3(256), 0, st, N, L.ntiles, radii, (const uint2*)((char*)ws->saved + L.o_ranges),
                     (const uint32_t*)((char*)ws->saved + L.o_tile_maxc), d);
  unsigned long long h[4];
  HIP_TRY(hipMemcpyAsync(h, d, 32, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < 4; ++i) stats_host[i] = (int64_t)h[i];
  return SGR_OK;
}

int sgr_profile_enable(uint32_t kind_mask) {
  g_prof.mask = kind_mask & ((1u << SGR_PROFILE_KINDS) - 1);
  return SGR_OK;
}

int sgr_profile_read(float ms_host[SGR_PROFILE_KINDS], int64_t launches_host[SGR_PROFILE_KINDS]) {
  for (int k = 0; k < SGR_PROFILE_KINDS; ++k) {
    prof_drain(k);
    if (ms_host) ms_host[k] = (float)g_prof.ms[k];
    if (launches_host) launches_host[k] = g_prof.launches[k];
    g_prof.ms[k] = 0.0;
    g_prof.launches[k] = 0;
  }
  return SGR_OK;
}

int sgr_query(const void* saved, int64_t* num_rendered_host, int32_t* overflow_host, void* stream) {
  if (!saved) return set_error(SGR_ERR_INVALID, "null saved block");
  SavedHeader h;
  HIP_TRY(hipMemcpyAsync(&h, saved, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (num_rendered_host) *num_rendered_host = h.num_rendered;
  if (overflow_host) *overflow_host = (int32_t)h.overflow;
  return SGR_OK;
}

}  // extern "C"
