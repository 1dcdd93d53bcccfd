// C-ABI entry points of the rasterizer.  Public contract: include/splat_hip.h.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <hip/hip_runtime.h>

#include "sgr_common.h"

namespace sgr {

void launch_preprocess_fwd(const SgrSettings&, const SgrInputs&, const SgrOutputs&, const Layout&, char*, hipStream_t);
void launch_preprocess_bwd(const SgrSettings&, const SgrInputs&, const int32_t*, const SgrGradInputs&, const Layout&,
                           const char*, char*, hipStream_t);
void launch_binning(const SgrSettings&, const SgrOutputs&, const Layout&, char*, char*, hipStream_t);
void launch_blend_fwd(const SgrSettings&, const SgrOutputs&, const Layout&, char*, char*, hipStream_t);
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

static Layout make_layout(int N, int H, int W, int64_t cap) { return Layout(N, H, W, cap); }

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
  if (N < 0 || H <= 0 || W <= 0 || ws->capacity <= 0 || ws->capacity > 0xffffffffll)
    return set_error(SGR_ERR_INVALID, "bad sizes N=%d H=%d W=%d cap=%lld", N, H, W, (long long)ws->capacity);
  if (H > 65535 * kTile || W > 65535 * kTile) return set_error(SGR_ERR_INVALID, "image too large");
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

  // header + per-tile pair counters: the only memset of the forward
  HIP_TRY(hipMemsetAsync(saved + L.o_hdr, 0, L.zero_bytes, st));
  launch_preprocess_fwd(*s, *in, *out, L, saved, st);                 // K1: project, footprint, count pairs per tile
  launch_binning(*s, *out, L, saved, scratch, st);                    // K2: tile/block scans   K3: scatter keys
  if (num_rendered_host) {
    uint32_t R = 0;
    HIP_TRY(hipMemcpyAsync(&R, &hdr->num_rendered, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *num_rendered_host = R;
    if ((int64_t)R > L.cap) return set_error(SGR_ERR_CAPACITY, "%u (tile, Gaussian) pairs exceed capacity %lld", R, (long long)L.cap);
  }
  launch_blend_fwd(*s, *out, L, saved, scratch, st);                  // K4: per-tile sort + compositing
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

int sgr_map_views(int32_t num_views, const SgrMapView* views, const SgrInputs* in, const SgrGradInputs* grads, float alpha,
                  float rgb_boundary_threshold, int32_t forward_only, void* stream) {
  if (num_views < 0 || (num_views > 0 && (!views || !in)) || (!forward_only && !grads))
    return set_error(SGR_ERR_INVALID, "map_views: null argument");
  for (int v = 0; v < num_views; ++v) {
    const SgrMapView& mv = views[v];
    if (int rc = sgr_forward(&mv.settings, in, &mv.out, &mv.ws, nullptr, stream)) return rc;
    if (forward_only) continue;
    if (int rc = sgr_mapping_loss(mv.settings.image_height, mv.settings.image_width, mv.out.color, mv.out.depth, mv.gt_image,
                                  mv.gt_depth, mv.exposure_a, mv.exposure_b, alpha, rgb_boundary_threshold, 1.0f, mv.loss,
                                  mv.dL_dimage, mv.dL_ddepth, mv.dL_dexposure, mv.dL_dexposure ? mv.dL_dexposure + 1 : nullptr,
                                  mv.loss_scratch, mv.loss_scratch_bytes, stream))
      return rc;
    SgrGradOutputs go = {mv.dL_dimage, mv.dL_ddepth};
    SgrGradInputs gi = *grads;
    gi.dL_dtau = mv.dL_dtau;
    if (int rc = sgr_backward(&mv.settings, in, mv.out.radii, &go, &gi, &mv.ws, stream)) return rc;
  }
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
