// C-ABI entry points of the rasterizer.  Public contract: include/splat_hip.h.
// Every stage is ONE launch for a whole batch of views (ViewTab by value, blockIdx.y = view); the single-view entry
// points are batches of one.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime.h>

#include "sgr_common.h"

namespace sgr {

void launch_preprocess_fwd(const ViewTab&, int, const LOff&, const Common&, const SgrInputs&, hipStream_t);
void launch_preprocess_bwd(const ViewTab&, int, const LOff&, const Common&, const SgrInputs&, const SgrGradInputs&, const FusedAdam*,
                           hipStream_t, bool mapping_loop = false);
void launch_binning(const ViewTab&, int, const LOff&, hipStream_t);
void launch_zero_heads(const ViewTab&, int, const LOff&, size_t, hipStream_t);
void launch_blend_fwd(const ViewTab&, int, const LOff&, const float*, const LossTab*, const LossCoef*, hipStream_t);
void launch_blend_bwd(const ViewTab&, int, const LOff&, const float*, const LossTab*, const LossCoef*, hipStream_t);
void launch_blend_fused(const ViewTab&, int, const LOff&, const float*, const LossTab&, const LossCoef&, hipStream_t);
bool blend_can_fuse(const LOff&);

// run-time options (sgr_set_option)
static int g_opt[SGR_OPT_COUNT] = {1, 0, 0};

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

int debug_flags() {
  static int f = -1;
  if (f < 0) { const char* e = getenv("SGR_DEBUG"); f = e ? atoi(e) : 0; }
  return f | (g_opt[SGR_OPT_SEGMENT_TEST] ? 0 : 4);
}

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
    uint32_t c = ranges[(size_t)t * kRngStride].y - (ranges[(size_t)t * kRngStride].x & ~kOverfull);
    r += c; re += min(c, tile_maxc[t]); ne += c > 0;
  }
  atomicAdd(&out[0], v); atomicAdd(&out[1], r); atomicAdd(&out[2], re); atomicAdd(&out[3], ne);
}

// walked list length (min(pairs, last contributor)) of every tile into 8 bins: 0, 1-4, 5-8, 9-16, 17-32, 33-64, 65-256, >256
__global__ void __launch_bounds__(256) hist_kernel(int ntiles, const uint2* __restrict__ ranges, const uint32_t* __restrict__ tile_maxc,
                                                    unsigned long long* __restrict__ out) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntiles; t += gridDim.x * blockDim.x) {
    uint32_t c = ranges[(size_t)t * kRngStride].y - (ranges[(size_t)t * kRngStride].x & ~kOverfull);
    c = min(c, tile_maxc[t]);
    int b = c == 0 ? 0 : c <= 4 ? 1 : c <= 8 ? 2 : c <= 16 ? 3 : c <= 32 ? 4 : c <= 64 ? 5 : c <= 256 ? 6 : 7;
    atomicAdd(&out[b], 1ull);
  }
}


__global__ void __launch_bounds__(256) depth_keys_kernel(int N, const GRec* __restrict__ grec, const int32_t* __restrict__ radii,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) out[i] = radii[i] > 0 ? grec[i].depth : 0.f;
}

static int check_settings(const SgrSettings* s, const SgrInputs* in) {
  const int N = s->num_gaussians, H = s->image_height, W = s->image_width;
  if (N < 0 || H <= 0 || W <= 0) return set_error(SGR_ERR_INVALID, "bad sizes N=%d H=%d W=%d", N, H, W);
  if (H > 65535 * kTile || W > 65535 * kTile) return set_error(SGR_ERR_INVALID, "image too large");
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->projmatrix_raw || !s->campos)
    return set_error(SGR_ERR_INVALID, "settings: bg/viewmatrix/projmatrix/projmatrix_raw/campos must be device pointers");
  if (N > 0) {
    if (!in->means3D || !in->opacities) return set_error(SGR_ERR_INVALID, "means3D and opacities are required");
    if ((in->shs != nullptr) == (in->colors_precomp != nullptr))
      return set_error(SGR_ERR_INVALID, "Please provide excatly one of either SHs or precomputed colors!");
    if (((in->scales != nullptr) && (in->rotations != nullptr)) == (in->cov3D_precomp != nullptr) ||
        ((in->scales != nullptr) != (in->rotations != nullptr)))
      return set_error(SGR_ERR_INVALID, "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    if (in->shs && (s->sh_degree < 0 || s->sh_degree > 3 || s->sh_coeffs < (s->sh_degree + 1) * (s->sh_degree + 1)))
      return set_error(SGR_ERR_INVALID, "sh_degree %d needs %d coefficients, shs has %d", s->sh_degree,
                       (s->sh_degree + 1) * (s->sh_degree + 1), s->sh_coeffs);
  }
  return SGR_OK;
}

static int check_workspace(const SgrWorkspace* ws, const Layout& L) {
  if (ws->capacity <= 0 || ws->capacity > 0xffffffffll) return set_error(SGR_ERR_INVALID, "bad capacity %lld", (long long)ws->capacity);
  if (!ws->saved || ws->saved_bytes < L.saved_bytes)
    return set_error(SGR_ERR_WORKSPACE, "saved workspace too small: %zu < %zu", ws->saved_bytes, L.saved_bytes);
  if (!ws->scratch || ws->scratch_bytes < L.scratch_bytes)
    return set_error(SGR_ERR_WORKSPACE, "scratch workspace too small: %zu < %zu", ws->scratch_bytes, L.scratch_bytes);
  return SGR_OK;
}

static Common make_common(const SgrSettings* s) {
  Common c;
  c.deg = s->sh_degree; c.M = s->sh_coeffs; c.tanfovx = s->tanfovx; c.tanfovy = s->tanfovy; c.mod = s->scale_modifier;
  c.bg = s->bg;
  c.upstream_pose_jac = g_opt[SGR_OPT_UPSTREAM_POSE_JACOBIAN];
  return c;
}

static void tab_set_view(ViewTab& t, int v, const SgrSettings* s, const SgrOutputs* out, const SgrWorkspace* ws) {
  t.viewmatrix[v] = s->viewmatrix; t.projmatrix[v] = s->projmatrix; t.campos[v] = s->campos; t.projraw[v] = s->projmatrix_raw;
  t.saved[v] = (char*)ws->saved; t.scratch[v] = (char*)ws->scratch;
  if (out) {
    t.color[v] = out->color; t.depth[v] = out->depth; t.opacity[v] = out->opacity; t.radii[v] = out->radii;
    t.n_touched[v] = out->n_touched;
  }
}

// forward of a batch that shares N, H, W, capacity and the view-independent settings
static int forward_batch(const ViewTab& tab, int nviews, const Layout& L, const Common& cm, const SgrInputs& in, hipStream_t st,
                         bool counters_clean = false, int max_list_hint = 0) {
  LOff d = L.dev();
  d.set_hint(max_list_hint);
  // header + per-tile pair counters; tile_scan re-zeroes the counters after reading them, so only blocks that never
  // went through a forward (or the caller does not vouch for) need this launch
  if (!counters_clean) launch_zero_heads(tab, nviews, d, L.zero_bytes, st);
  launch_preprocess_fwd(tab, nviews, d, cm, in, st);     // K1: project, footprint, count pairs per tile
  launch_binning(tab, nviews, d, st);                    // K2: tile/block scans   K3: scatter keys
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
  if (int rc = check_settings(s, in)) return rc;
  const int N = s->num_gaussians;
  if (!out->color || !out->depth || !out->opacity || (N > 0 && (!out->radii || !out->n_touched)))
    return set_error(SGR_ERR_INVALID, "null output pointer");
  Layout L = make_layout(N, s->image_height, s->image_width, ws->capacity);
  if (int rc = check_workspace(ws, L)) return rc;
  hipStream_t st = (hipStream_t)stream;
  ViewTab tab = {};
  tab_set_view(tab, 0, s, out, ws);
  Common cm = make_common(s);
  if (int rc = forward_batch(tab, 1, L, cm, *in, st, ws->counters_clean != 0, ws->max_list_hint)) return rc;
  if (num_rendered_host) {
    uint32_t h2[2] = {0u, 0u};        // num_rendered, overflow
    HIP_TRY(hipMemcpyAsync(h2, (char*)ws->saved + L.o_hdr, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint32_t R = h2[0];
    *num_rendered_host = R;
    if (h2[1] == 2u)
      return set_error(SGR_ERR_INVALID, "more than %u splats on one 8x8 tile: the map has degenerated (16-bit tile counters)", kTileCountLimit);
    if ((int64_t)R > L.cap) return set_error(SGR_ERR_CAPACITY, "%u (tile, Gaussian) pairs exceed capacity %lld", R, (long long)L.cap);
  }
  LOff d1 = L.dev();
  d1.set_hint(ws->max_list_hint);
  launch_blend_fwd(tab, 1, d1, s->bg, nullptr, nullptr, st);          // K4: per-tile sort + compositing
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

int sgr_backward(const SgrSettings* s, const SgrInputs* in, const int32_t* radii, const SgrGradOutputs* go,
                 const SgrGradInputs* gi, const SgrWorkspace* ws, void* stream) {
  if (!s || !in || !go || !gi || !ws) return set_error(SGR_ERR_INVALID, "null argument");
  if (int rc = check_settings(s, in)) return rc;
  const int N = s->num_gaussians;
  if (!go->dL_dcolor) return set_error(SGR_ERR_INVALID, "dL_dcolor is required");
  if (N > 0 && !radii) return set_error(SGR_ERR_INVALID, "radii is required");
  if (gi->stat_grad_accum && (!gi->stat_denom || !gi->stat_max_radii))
    return set_error(SGR_ERR_INVALID, "stat_grad_accum needs stat_denom and stat_max_radii");
  Layout L = make_layout(N, s->image_height, s->image_width, ws->capacity);
  if (int rc = check_workspace(ws, L)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    if (gi->dL_dtau) HIP_TRY(hipMemsetAsync(gi->dL_dtau, 0, 24, st));
    return SGR_OK;
  }
  ViewTab tab = {};
  tab_set_view(tab, 0, s, nullptr, ws);
  tab.radii[0] = const_cast<int32_t*>(radii);
  tab.dL_dcolor[0] = go->dL_dcolor; tab.dL_ddepth[0] = go->dL_ddepth; tab.dL_dtau[0] = gi->dL_dtau;
  LOff d = L.dev();
  Common cm = make_common(s);
  launch_blend_bwd(tab, 1, d, s->bg, nullptr, nullptr, st);
  launch_preprocess_bwd(tab, 1, d, cm, *in, *gi, nullptr, st);
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

}  // extern "C"

__global__ void __launch_bounds__(256) densify_stats_kernel(int64_t n, const float* __restrict__ m2, const int32_t* __restrict__ radii,
                                                            float* __restrict__ accum, float* __restrict__ denom, float* __restrict__ maxr) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = radii[i];
  if (r <= 0) return;
  const F3 g = ld3(m2 + 3 * i);
  accum[i] += sqrtf(g.x * g.x + g.y * g.y);
  denom[i] += 1.f;
  maxr[i] = fmaxf(maxr[i], (float)r);
}

extern "C" {

int sgr_densify_stats(int64_t n, const float* dL_dmeans2D, const int32_t* radii, float* stat_grad_accum, float* stat_denom,
                      float* stat_max_radii, void* stream) {
  if (n < 0 || (n > 0 && (!dL_dmeans2D || !radii || !stat_grad_accum || !stat_denom || !stat_max_radii)))
    return set_error(SGR_ERR_INVALID, "densify_stats: null argument");
  if (n > 0)
    hipLaunchKernelGGL(densify_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, dL_dmeans2D, radii,
                       stat_grad_accum, stat_denom, stat_max_radii);
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

int sgr_backward_views(int32_t num_views, const SgrBackwardView* views, const SgrInputs* in, const SgrGradInputs* grad_in, void* stream) {
  if (num_views < 0 || (num_views > 0 && (!views || !in || !grad_in))) return set_error(SGR_ERR_INVALID, "backward_views: null argument");
  if (num_views == 0) return SGR_OK;
  hipStream_t st = (hipStream_t)stream;
  const SgrBackwardView& f = views[0];
  bool uniform = in->shs && in->scales && in->rotations && !in->colors_precomp && !in->cov3D_precomp && f.settings.sh_degree == 0 &&
                 f.settings.num_gaussians > 0;
  for (int v = 0; v < num_views; ++v) {
    const SgrBackwardView& m = views[v];
    if (int rc = check_settings(&m.settings, in)) return rc;
    if (!m.dL_dcolor || (m.settings.num_gaussians > 0 && !m.radii)) return set_error(SGR_ERR_INVALID, "backward_views: dL_dcolor and radii are required");
    uniform = uniform && m.settings.num_gaussians == f.settings.num_gaussians && m.settings.image_height == f.settings.image_height &&
              m.settings.image_width == f.settings.image_width && m.settings.tanfovx == f.settings.tanfovx &&
              m.settings.tanfovy == f.settings.tanfovy && m.settings.scale_modifier == f.settings.scale_modifier &&
              m.settings.sh_degree == f.settings.sh_degree && m.settings.sh_coeffs == f.settings.sh_coeffs &&
              m.settings.bg == f.settings.bg && m.ws.capacity == f.ws.capacity;
    for (int u = 0; u < v; ++u) uniform = uniform && views[u].ws.scratch != m.ws.scratch && views[u].ws.saved != m.ws.saved;
  }
  SgrGradInputs g = *grad_in;
  g.dL_dmeans2D = nullptr;
  g.dL_dtau = nullptr;
  if (!uniform) {
    // one view after the other: the first call defines every element (unless the caller accumulates), the others add
    for (int v = 0; v < num_views; ++v) {
      const SgrBackwardView& m = views[v];
      const int N = m.settings.num_gaussians;
      Layout L = make_layout(N, m.settings.image_height, m.settings.image_width, m.ws.capacity);
      if (int rc = check_workspace(&m.ws, L)) return rc;
      if (N == 0) {
        if (m.dL_dtau) HIP_TRY(hipMemsetAsync(m.dL_dtau, 0, 24, st));
        continue;
      }
      ViewTab tab = {};
      tab_set_view(tab, 0, &m.settings, nullptr, &m.ws);
      tab.radii[0] = const_cast<int32_t*>(m.radii);
      tab.dL_dcolor[0] = m.dL_dcolor; tab.dL_ddepth[0] = m.dL_ddepth; tab.dL_dtau[0] = m.dL_dtau; tab.dL_dmeans2D[0] = m.dL_dmeans2D;
      SgrGradInputs gv = g;
      gv.accumulate = (grad_in->accumulate || v > 0) ? 1 : 0;
      Common cm = make_common(&m.settings);
      LOff d = L.dev();
      launch_blend_bwd(tab, 1, d, m.settings.bg, nullptr, nullptr, st);
      launch_preprocess_bwd(tab, 1, d, cm, *in, gv, nullptr, st);
    }
    HIP_TRY(hipGetLastError());
    return SGR_OK;
  }
  Layout L = make_layout(f.settings.num_gaussians, f.settings.image_height, f.settings.image_width, f.ws.capacity);
  LOff d = L.dev();
  Common cm = make_common(&f.settings);
  for (int base = 0; base < num_views; base += kMaxViews) {
    const int nv = num_views - base < kMaxViews ? num_views - base : kMaxViews;
    ViewTab tab = {};
    for (int v = 0; v < nv; ++v) {
      const SgrBackwardView& m = views[base + v];
      if (int rc = check_workspace(&m.ws, L)) return rc;
      tab_set_view(tab, v, &m.settings, nullptr, &m.ws);
      tab.radii[v] = const_cast<int32_t*>(m.radii);
      tab.dL_dcolor[v] = m.dL_dcolor; tab.dL_ddepth[v] = m.dL_ddepth; tab.dL_dtau[v] = m.dL_dtau; tab.dL_dmeans2D[v] = m.dL_dmeans2D;
    }
    SgrGradInputs gv = g;
    gv.accumulate = (grad_in->accumulate || base > 0) ? 1 : 0;
    launch_blend_bwd(tab, nv, d, f.settings.bg, nullptr, nullptr, st);
    launch_preprocess_bwd(tab, nv, d, cm, *in, gv, nullptr, st);
  }
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

}  // extern "C"

// fused: optimiser tail to run inside the gather pass; *fused_done tells whether it did (uniform single-chunk batches only)
// store_sinks: the caller promised its gradient sinks STORE semantics (SgrMapStep.grads_clean == -3: every row is overwritten, the
// sinks need not be zero on entry).  Only the fused gather pass can store; whenever a batch does not take it (pose gradients asked
// for, heterogeneous views, more than kMaxViews) the accumulating passes run instead -- onto sinks that are zeroed HERE first, so
// that the promise holds on every path (it used to add this iteration's sums to the previous iteration's leftovers).
static int map_views_impl(int32_t num_views, const SgrMapView* views, const SgrInputs* in, const SgrGradInputs* grads, float alpha,
                          float rgb_boundary_threshold, int32_t forward_only, FusedAdam* fused, bool* fused_done,
                          void* stream, bool store_sinks = false) {
  if (fused_done) *fused_done = false;
  if (num_views < 0 || (num_views > 0 && (!views || !in)) || (!forward_only && !grads))
    return set_error(SGR_ERR_INVALID, "map_views: null argument");
  if (num_views == 0) return SGR_OK;
  hipStream_t st = (hipStream_t)stream;
  // batched execution needs one Layout (same N, H, W, capacity), the same view-independent settings and private scratch
  bool uniform = true;
  const SgrMapView& f = views[0];
  for (int v = 0; v < num_views; ++v) {
    const SgrMapView& m = views[v];
    if (int rc = check_settings(&m.settings, in)) return rc;
    uniform = uniform && m.settings.num_gaussians == f.settings.num_gaussians && m.settings.image_height == f.settings.image_height &&
              m.settings.image_width == f.settings.image_width && m.settings.tanfovx == f.settings.tanfovx &&
              m.settings.tanfovy == f.settings.tanfovy && m.settings.scale_modifier == f.settings.scale_modifier &&
              m.settings.sh_degree == f.settings.sh_degree && m.settings.sh_coeffs == f.settings.sh_coeffs &&
              m.settings.bg == f.settings.bg &&
              m.ws.capacity == f.ws.capacity;
    // several views accumulate through fixed-size gradient records: the reference's default inputs only
    if (num_views > 1) uniform = uniform && in->shs && m.settings.sh_degree == 0 && in->scales && in->rotations;
    for (int u = 0; u < v; ++u) uniform = uniform && views[u].ws.scratch != m.ws.scratch && views[u].ws.saved != m.ws.saved;
    if (!forward_only && grads && !grads->accumulate && num_views > 1)
      return set_error(SGR_ERR_INVALID, "map_views: several views need accumulate != 0");
  }
  if (store_sinks && !forward_only && grads && !(fused && uniform && f.settings.num_gaussians > 0 && num_views <= kMaxViews)) {
    const size_t n = (size_t)f.settings.num_gaussians, m = (size_t)(f.settings.sh_coeffs > 0 ? f.settings.sh_coeffs : 1);
    if (grads->dL_dmeans3D) HIP_TRY(hipMemsetAsync(grads->dL_dmeans3D, 0, n * 12, st));
    if (grads->dL_dshs) HIP_TRY(hipMemsetAsync(grads->dL_dshs, 0, n * m * 12, st));
    if (grads->dL_dopacities) HIP_TRY(hipMemsetAsync(grads->dL_dopacities, 0, n * 4, st));
    if (grads->dL_dscales) HIP_TRY(hipMemsetAsync(grads->dL_dscales, 0, n * 12, st));
    if (grads->dL_drotations) HIP_TRY(hipMemsetAsync(grads->dL_drotations, 0, n * 16, st));
  }
  if (!uniform || f.settings.num_gaussians == 0) {       // heterogeneous views: one after the other (shared scratch is fine)
    for (int v = 0; v < num_views; ++v) {
      const SgrMapView& mv = views[v];
      if (!mv.out.color || !mv.out.depth || !mv.out.opacity || !mv.out.n_touched)
        return set_error(SGR_ERR_INVALID, "map_views: outputs may only be omitted in uniform batches");
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
  Layout L = make_layout(f.settings.num_gaussians, f.settings.image_height, f.settings.image_width, f.ws.capacity);
  LOff d = L.dev();
  int hint = 0;
  for (int v = 0; v < num_views; ++v) hint = views[v].ws.max_list_hint > hint ? views[v].ws.max_list_hint : hint;
  d.set_hint(hint);
  Common cm = make_common(&f.settings);
  const int HW = f.settings.image_height * f.settings.image_width;
  for (int base = 0; base < num_views; base += kMaxViews) {
    const int nv = num_views - base < kMaxViews ? num_views - base : kMaxViews;
    ViewTab tab = {};
    LossTab lt = {};
    bool all_clean = true;
    for (int v = 0; v < nv; ++v) {
      const SgrMapView& m = views[base + v];
      all_clean = all_clean && m.ws.counters_clean != 0;
      if (int rc = check_workspace(&m.ws, L)) return rc;
      const bool no_images = !m.out.color && !m.out.depth && !m.out.opacity && !forward_only;   // loss-only iteration
      if ((!no_images && (!m.out.color || !m.out.depth || !m.out.opacity)) || !m.out.radii)
        return set_error(SGR_ERR_INVALID, "map_views: null output pointer");
      tab_set_view(tab, v, &m.settings, &m.out, &m.ws);
      if (!forward_only) {
        if (!m.gt_image || !m.gt_depth || !m.dL_dimage || !m.dL_ddepth || !m.loss_scratch ||
            m.loss_scratch_bytes < (size_t)L.ntiles * sizeof(LossPart))
          return set_error(SGR_ERR_INVALID, "map_views: loss buffers missing (loss_scratch needs 16 B per 8x8 tile = %zu)",
                           (size_t)L.ntiles * sizeof(LossPart));
        tab.dL_dcolor[v] = m.dL_dimage; tab.dL_ddepth[v] = m.dL_ddepth; tab.dL_dtau[v] = m.dL_dtau;
        // (image / depth are unused by the fused epilogue: it has the pixel in registers)
        lt.image[v] = m.out.color; lt.depth[v] = m.out.depth; lt.gt_image[v] = m.gt_image; lt.gt_depth[v] = m.gt_depth;
        lt.exp_a[v] = m.exposure_a; lt.exp_b[v] = m.exposure_b; lt.loss[v] = m.loss; lt.dimage[v] = m.dL_dimage;
        lt.ddepth[v] = m.dL_ddepth; lt.da[v] = m.dL_dexposure; lt.db[v] = m.dL_dexposure ? m.dL_dexposure + 1 : nullptr;
        lt.parts[v] = m.loss_scratch;
      }
    }
    if (int rc = forward_batch(tab, nv, L, cm, *in, st, all_clean, d.mean_hint)) return rc;
    if (forward_only) {
      launch_blend_fwd(tab, nv, d, f.settings.bg, nullptr, nullptr, st);
      continue;
    }
    // the mapping loss rides in the compositing epilogue (no second pass over the images); only its tiny fixed-order
    // reduction is a separate launch
    LossCoef lc = {alpha / (3.f * (float)HW), (1.f - alpha) / (float)HW, rgb_boundary_threshold};
    // forward, loss and backward of a tile run in the same wave (one launch) unless the option is off (profiling the two
    // halves separately, bitwise A/B tests)
    const bool fused_blend = g_opt[SGR_OPT_FUSED_BLEND] != 0 && blend_can_fuse(d);
    if (fused_blend) launch_blend_fused(tab, nv, d, f.settings.bg, lt, lc, st);
    else launch_blend_fwd(tab, nv, d, f.settings.bg, &lt, &lc, st);
    const bool fuse = fused && num_views <= kMaxViews;
    if (fuse) {        // the loss sums (and the exposure step that consumes them) ride in the optimiser launch
      fused->tail_views = nv; fused->tail_nparts = L.ntiles;
      fused->tail_inv_rgb = 1.f / (3.f * (float)HW); fused->tail_inv_dep = 1.f / (float)HW; fused->tail_alpha = alpha;
      for (int v = 0; v < nv; ++v) {
        fused->tail_parts[v] = lt.parts[v]; fused->tail_loss[v] = lt.loss[v]; fused->tail_da[v] = lt.da[v]; fused->tail_db[v] = lt.db[v];
      }
    } else {
      launch_mapping_loss_final(lt, nv, HW, L.ntiles, alpha, st);
    }
    if (!fused_blend) launch_blend_bwd(tab, nv, d, f.settings.bg, &lt, &lc, st);
    launch_preprocess_bwd(tab, nv, d, cm, *in, *grads, fuse ? fused : nullptr, st, /*mapping_loop=*/true);
    if (fuse && fused_done) *fused_done = true;
  }
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

extern "C" {

int sgr_map_views(int32_t num_views, const SgrMapView* views, const SgrInputs* in, const SgrGradInputs* grads, float alpha,
                  float rgb_boundary_threshold, int32_t forward_only, void* stream) {
  return map_views_impl(num_views, views, in, grads, alpha, rgb_boundary_threshold, forward_only, nullptr, nullptr, stream);
}

}  // extern "C"

// One iteration.  When the Adam step directly follows the views of a uniform single-chunk batch and the gradient sinks
// are the optimiser's own gradient buffers, gather + Adam + next activations run as ONE pass (gather_adam_kernel).
static int map_step_impl(const SgrMapStep* p, bool skip_activate, bool grads_clean, bool allow_fuse, bool* fused_out, void* stream) {
  if (fused_out) *fused_out = false;
  if (!p) return set_error(SGR_ERR_INVALID, "map_step: null argument");
  // an optimiser-only step (no views: the second half of a multi-GPU iteration, after the all-reduce) writes the
  // activations of the UPDATED parameters in the Adam pass itself instead of activating the old ones first
  const bool adam_only = p->num_views == 0 && p->adam_groups;
  if (!skip_activate && !adam_only && (p->scaling || p->rotation || p->opacity))
    if (int rc = sgr_activate(p->num_gaussians, p->scaling, p->rotation, p->opacity, p->scales_out, p->rot_out, p->opac_out, stream)) return rc;
  FusedAdam fa;
  bool try_fuse = false;
  if (allow_fuse && p->adam_groups && p->num_views > 0 && p->num_views <= kMaxViews && !p->forward_only && p->grads && p->in && p->views) {
    const SgrGradInputs& g = *p->grads;
    const SgrAdamGroup* G = p->adam_groups;
    try_fuse = g.accumulate && g.dL_dmeans3D == G[0].grad && g.dL_dshs == G[1].grad && g.dL_dopacities == G[2].grad &&
               g.dL_dscales == G[3].grad && g.dL_drotations == G[4].grad && !g.dL_dmeans2D && !g.dL_dcolors_precomp &&
               !g.dL_dcov3D_precomp && p->in->shs && !p->in->colors_precomp && p->in->scales && p->in->rotations &&
               p->num_gaussians == p->views[0].settings.num_gaussians;
    for (int v = 0; v < p->num_views && try_fuse; ++v)
      try_fuse = !p->views[v].dL_dtau && p->views[v].settings.sh_degree == 0 && p->views[v].settings.sh_coeffs == 1;
    if (try_fuse) {
      if (int rc = make_fused_adam(p->num_gaussians, G, p->beta1, p->beta2, p->eps, p->iso_weight, &fa)) return rc;
      fa.grads_clean = grads_clean ? 1 : 0;
      fa.s_out = p->scales_out; fa.r_out = p->rot_out; fa.o_out = p->opac_out;
      fa.stat_accum = g.stat_grad_accum;
      fa.stat_denom = g.stat_grad_accum ? g.stat_denom : nullptr;
      fa.stat_maxr = g.stat_grad_accum ? g.stat_max_radii : nullptr;
      if (p->exp_rows > 0) {
        if (!p->exp_param || !p->exp_grad || !p->exp_avg || !p->exp_avg_sq || !p->exp_step || !p->exp_active || p->exp_row_width <= 0)
          return set_error(SGR_ERR_INVALID, "map_step: exposure block incomplete");
        fa.exp_rows = p->exp_rows; fa.exp_width = p->exp_row_width; fa.exp_param = p->exp_param; fa.exp_grad = p->exp_grad;
        fa.exp_avg = p->exp_avg; fa.exp_avg_sq = p->exp_avg_sq; fa.exp_step = p->exp_step; fa.exp_active = p->exp_active;
        fa.exp_lr = p->exp_lr; fa.exp_b1 = p->exp_beta1; fa.exp_b2 = p->exp_beta2; fa.exp_eps = p->exp_eps;
      }
    }
  }
  // grads_clean == -2 / -3: no optimiser step in this call, but the gather pass of the fused form (with its riders: loss sums
  // and the exposure step) adds the views' gradients to the sinks / stores them there -- the first half of a multi-GPU iteration
  if ((p->grads_clean == -2 || p->grads_clean == -3) && !p->adam_groups && p->num_views > 0 && p->num_views <= kMaxViews && !p->forward_only && p->grads &&
      p->in && p->views) {
    const SgrGradInputs& g = *p->grads;
    try_fuse = g.accumulate && g.dL_dmeans3D && g.dL_dshs && g.dL_dopacities && g.dL_dscales && g.dL_drotations &&
               !g.dL_dmeans2D && !g.dL_dcolors_precomp && !g.dL_dcov3D_precomp && p->in->shs && !p->in->colors_precomp &&
               p->in->scales && p->in->rotations && p->num_gaussians == p->views[0].settings.num_gaussians;
    for (int v = 0; v < p->num_views && try_fuse; ++v)
      try_fuse = !p->views[v].dL_dtau && p->views[v].settings.sh_degree == 0 && p->views[v].settings.sh_coeffs == 1;
    if (try_fuse) {
      fa = FusedAdam{};
      fa.gather_only = p->grads_clean == -3 ? 2 : 1;     // (-3: the sums are stored for every Gaussian, the sinks need no zeroing)
      fa.G.g[0].grad = g.dL_dmeans3D; fa.G.g[1].grad = g.dL_dshs; fa.G.g[2].grad = g.dL_dopacities;
      fa.G.g[3].grad = g.dL_dscales; fa.G.g[4].grad = g.dL_drotations;
      fa.stat_accum = g.stat_grad_accum;
      fa.stat_denom = g.stat_grad_accum ? g.stat_denom : nullptr;
      fa.stat_maxr = g.stat_grad_accum ? g.stat_max_radii : nullptr;
      if (p->exp_rows > 0) {
        if (!p->exp_param || !p->exp_grad || !p->exp_avg || !p->exp_avg_sq || !p->exp_step || !p->exp_active || p->exp_row_width <= 0)
          return set_error(SGR_ERR_INVALID, "map_step: exposure block incomplete");
        fa.exp_rows = p->exp_rows; fa.exp_width = p->exp_row_width; fa.exp_param = p->exp_param; fa.exp_grad = p->exp_grad;
        fa.exp_avg = p->exp_avg; fa.exp_avg_sq = p->exp_avg_sq; fa.exp_step = p->exp_step; fa.exp_active = p->exp_active;
        fa.exp_lr = p->exp_lr; fa.exp_b1 = p->exp_beta1; fa.exp_b2 = p->exp_beta2; fa.exp_eps = p->exp_eps;
      }
    }
  }
  bool fused = false;
  if (p->num_views > 0)
    if (int rc = map_views_impl(p->num_views, p->views, p->in, p->grads, p->alpha, p->rgb_boundary_threshold, p->forward_only,
                                try_fuse ? &fa : nullptr, &fused, stream, p->grads_clean == -3 && !p->adam_groups))
      return rc;
  if (p->adam_groups && !fused)
    if (int rc = gaussian_adam_step_act(p->num_gaussians, p->adam_groups, p->beta1, p->beta2, p->eps, p->iso_weight,
                                        adam_only && p->scaling ? p->scales_out : nullptr,
                                        adam_only && p->rotation ? p->rot_out : nullptr,
                                        adam_only && p->opacity ? p->opac_out : nullptr, stream))
      return rc;
  if (p->exp_rows > 0 && !fused)
    if (int rc = sgr_masked_adam(p->exp_rows, p->exp_row_width, p->exp_param, p->exp_grad, p->exp_avg, p->exp_avg_sq, p->exp_step,
                                 p->exp_active, p->exp_lr, p->exp_beta1, p->exp_beta2, p->exp_eps, stream))
      return rc;
  if (fused_out) *fused_out = fused;
  return SGR_OK;
}

extern "C" {

int sgr_map_step(const SgrMapStep* p, void* stream) {
  if (!p) return set_error(SGR_ERR_INVALID, "map_step: null argument");
  return map_step_impl(p, false, p->grads_clean > 0, p->grads_clean >= 0, nullptr, stream);
}

int sgr_map_run(const SgrMapRun* r, void* stream) {
  if (!r || r->num_iters < 0 || r->num_window < 0 || r->picks_per_iter < 0 || (r->num_window > 0 && !r->window) ||
      (r->picks_per_iter > 0 && (!r->pool || !r->picks || r->pool_size <= 0)))
    return set_error(SGR_ERR_INVALID, "map_run: bad argument");
  const int nv = r->num_window + r->picks_per_iter;
  if (nv > 64) return set_error(SGR_ERR_INVALID, "map_run: more than 64 views per iteration");
  SgrMapView views[64];
  for (int v = 0; v < r->num_window; ++v) views[v] = r->window[v];
  std::vector<char> pool_used((size_t)(r->pool_size > 0 ? r->pool_size : 0), 0);   // a block is clean after its first forward
  std::vector<char> slot_used((size_t)(r->picks_per_iter > 0 ? r->picks_per_iter : 0), 0);
  SgrMapStep st = r->step;
  st.views = views;
  st.num_views = nv;
  st.adam_groups = r->adam_groups;
  const int32_t exp_rows = st.exp_rows;
  bool prev_fused = false;
  for (int it = 0; it < r->num_iters; ++it) {
    for (int j = 0; j < r->picks_per_iter; ++j) {
      const int32_t k = r->picks[(size_t)it * r->picks_per_iter + j];
      if (k < 0 || k >= r->pool_size) return set_error(SGR_ERR_INVALID, "map_run: pick %d outside the pool", k);
      views[r->num_window + j] = r->pool[k];
      if (r->pick_ws) {                   // shared workspace slots: pick j renders in slot j whichever camera it is
        views[r->num_window + j].ws = r->pick_ws[j];
        if (slot_used[j]) views[r->num_window + j].ws.counters_clean = 1;
        slot_used[j] = 1;
      } else if (pool_used[k]) {
        views[r->num_window + j].ws.counters_clean = 1;
      }
      pool_used[k] = 1;
    }
    if (it == 1)
      for (int v = 0; v < r->num_window; ++v) views[v].ws.counters_clean = 1;
    if (r->n_touched_last_only) {       // the per-Gaussian "touched" counters only matter after the run (mapper.py:494-498)
      const bool last = it == r->num_iters - 1;
      for (int v = 0; v < r->num_window; ++v) views[v].out.n_touched = last ? r->window[v].out.n_touched : nullptr;
      if (!last)
        for (int j = 0; j < r->picks_per_iter; ++j) views[r->num_window + j].out.n_touched = nullptr;
    }
    if (r->adam_groups) {
      if (r->lr0) r->adam_groups[0].lr = r->lr0[it];
      for (int g = 0; g < 5; ++g)
        if (!r->adam_groups[g].skip) r->adam_groups[g].step += 1;
    }
    if (r->pool_exp_row && r->picks_per_iter > 0 && exp_rows > 0) {
      const int32_t row = r->pool_exp_row[r->picks[(size_t)it * r->picks_per_iter]];
      const size_t o = (size_t)(row < 0 ? 0 : row) * (size_t)st.exp_row_width;
      st.exp_rows = row < 0 ? 0 : 1;
      st.exp_param = r->step.exp_param + o; st.exp_grad = r->step.exp_grad + o;
      st.exp_avg = r->step.exp_avg + o; st.exp_avg_sq = r->step.exp_avg_sq + o;
      st.exp_step = r->step.exp_step + (row < 0 ? 0 : row);
    }
    // after a fused tail the activations of the updated parameters are already written and the sinks are clean
    bool fused = false;
    if (int rc = map_step_impl(&st, prev_fused, it == 0 ? r->step.grads_clean > 0 : r->adam_groups != nullptr, r->step.grads_clean >= 0,
                               &fused, stream))
      return rc;
    prev_fused = fused;
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

int sgr_query_depth_keys(const SgrWorkspace* ws, int32_t N, int32_t H, int32_t W, const int32_t* radii, float* depth_out, void* stream) {
  if (!ws || !ws->saved || !radii || !depth_out) return set_error(SGR_ERR_INVALID, "null argument");
  Layout L = make_layout(N, H, W, ws->capacity);
  if (N > 0)
    hipLaunchKernelGGL(depth_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N,
                       (const GRec*)((const char*)ws->saved + L.o_grec), radii, depth_out);
  HIP_TRY(hipGetLastError());
  return SGR_OK;
}

int sgr_query_list_histogram(const SgrWorkspace* ws, int32_t N, int32_t H, int32_t W, int64_t hist_host[8], void* stream) {
  if (!ws || !ws->saved || !ws->scratch || !hist_host) return set_error(SGR_ERR_INVALID, "null argument");
  Layout L = make_layout(N, H, W, ws->capacity);
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* d = (unsigned long long*)ws->scratch;
  HIP_TRY(hipMemsetAsync(d, 0, 64, st));
  hipLaunchKernelGGL(hist_kernel, dim3(32), dim3(256), 0, st, L.ntiles, (const uint2*)((char*)ws->saved + L.o_ranges),
                     (const uint32_t*)((char*)ws->saved + L.o_tile_maxc), d);
  unsigned long long h[8];
  HIP_TRY(hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < 8; ++i) hist_host[i] = (int64_t)h[i];
  return SGR_OK;
}

int sgr_set_option(int32_t option, int32_t value) {
  if (option < 0 || option >= SGR_OPT_COUNT) return set_error(SGR_ERR_INVALID, "unknown option %d", option);
  g_opt[option] = value;
  return SGR_OK;
}

int sgr_get_option(int32_t option) { return option >= 0 && option < SGR_OPT_COUNT ? g_opt[option] : -1; }

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

int sgr_header_to_host(const void* saved, void* pinned_host64, void* stream) {
  if (!saved || !pinned_host64) return set_error(SGR_ERR_INVALID, "null argument");
  HIP_TRY(hipMemcpyAsync(pinned_host64, saved, sizeof(SavedHeader), hipMemcpyDeviceToHost, (hipStream_t)stream));
  return SGR_OK;
}

int sgr_query_header(const void* saved, uint32_t words_host[16], void* stream) {
  if (!saved || !words_host) return set_error(SGR_ERR_INVALID, "null argument");
  static_assert(sizeof(SavedHeader) == 64, "header is 16 words");
  HIP_TRY(hipMemcpyAsync(words_host, saved, sizeof(SavedHeader), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
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
