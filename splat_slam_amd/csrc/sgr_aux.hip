// The smaller C-ABI entry points around the rasterizer (include/splat_hip.h):
//   sgr_mapping_loss  fused fwd+bwd of get_loss_mapping        /root/reference/thirdparty/monogs/utils/slam_utils.py:71-105
//   sgr_adam_step     torch.optim.Adam step on a flat slab     /root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:264-313
//   sknn_dist2        simple_knn distCUDA2                     /root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:194-200
//   se3_*             lietorch SE3 ops of the mapping path     /root/reference/thirdparty/glorie_slam/depth_video.py:327-330
// All are HBM-streaming or tiny; each is a single pass with fixed-order reductions (bitwise reproducible).
#include <cmath>

#include <cstdint>

#include "sgr_common.h"

namespace sgr {
int set_error(int code, const char* fmt, ...);

// ------------------------------------------------------------------------------------------------ mapping loss

__global__ void __launch_bounds__(256) mapping_loss_kernel(LossTab tab, int HW, float w_rgb, float w_dep, float thr) {
  const int vw = blockIdx.y;
  const float* __restrict__ image = tab.image[vw];
  const float* __restrict__ depth = tab.depth[vw];
  const float* __restrict__ gt_image = tab.gt_image[vw];
  const float* __restrict__ gt_depth = tab.gt_depth[vw];
  float* __restrict__ dimage = tab.dimage[vw];
  float* __restrict__ ddepth = tab.ddepth[vw];
  LossPart* __restrict__ parts = (LossPart*)tab.parts[vw];
  const float ea = tab.exp_a[vw] ? __expf(tab.exp_a[vw][0]) : 1.f;
  const float eb = tab.exp_b[vw] ? tab.exp_b[vw][0] : 0.f;
  LossPart acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float g0 = gt_image[p], g1 = gt_image[HW + p], g2 = gt_image[2 * HW + p];
    bool m = (g0 + g1 + g2) > thr;
    float gt[3] = {g0, g1, g2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float I = image[c * HW + p];
      float r = m ? (ea * I + eb) - gt[c] : 0.f;
      acc.rgb += fabsf(r);
      float sgn = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f);
      float dab = w_rgb * sgn;            // dL/d(image_ab)
      if (dimage) dimage[c * HW + p] = dab * ea;
      acc.da += dab * ea * I;
      acc.db += dab;
    }
    float gd = gt_depth[p];
    bool md = gd > 0.01f;
    float rd = md ? depth[p] - gd : 0.f;
    acc.dep += fabsf(rd);
    if (ddepth) ddepth[p] = w_dep * ((rd > 0.f) ? 1.f : ((rd < 0.f) ? -1.f : 0.f));
  }
  __shared__ LossPart red[4];
  float v[4] = {acc.rgb, acc.dep, acc.da, acc.db};
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = {v[0], v[1], v[2], v[3]};
  __syncthreads();
  if (threadIdx.x == 0) {
    LossPart t = red[0];
    for (int w = 1; w < 4; ++w) { t.rgb += red[w].rgb; t.dep += red[w].dep; t.da += red[w].da; t.db += red[w].db; }
    parts[blockIdx.x] = t;
  }
}

// second stage: one 256-thread block adds a view's partials in a fixed order (thread-strided, then DPP + LDS)
__device__ __forceinline__ void loss_final_view(const LossPart* __restrict__ parts, int nparts, float inv_rgb, float inv_dep,
                                                float alpha, float* loss, float* da, float* db, LossPart* red /*LDS[4]*/) {
#pragma clang fp contract(off)
  LossPart t = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nparts; i += 256) {
    LossPart p = parts[i];
    t.rgb += p.rgb; t.dep += p.dep; t.da += p.da; t.db += p.db;
  }
  float v[4] = {t.rgb, t.dep, t.da, t.db};
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = {v[0], v[1], v[2], v[3]};
  __syncthreads();
  if (threadIdx.x == 0) {
    LossPart s = red[0];
    for (int w = 1; w < 4; ++w) { s.rgb += red[w].rgb; s.dep += red[w].dep; s.da += red[w].da; s.db += red[w].db; }
    if (loss) loss[0] = alpha * (s.rgb * inv_rgb) + (1.f - alpha) * (s.dep * inv_dep);
    if (da) da[0] = s.da;
    if (db) db[0] = s.db;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) mapping_loss_final_kernel(LossTab tab, int nparts, float inv_rgb, float inv_dep, float alpha) {
  __shared__ LossPart red[4];
  const int vw = blockIdx.x;
  loss_final_view((const LossPart*)tab.parts[vw], nparts, inv_rgb, inv_dep, alpha, tab.loss[vw], tab.da[vw], tab.db[vw], red);
}

static int loss_blocks(int HW) {
  int blocks = (HW + 255) / 256;
  return blocks > 1024 ? 1024 : blocks;
}

void launch_mapping_loss_final(const LossTab& tab, int nviews, int HW, int nparts, float alpha, hipStream_t st) {
  float inv_rgb = 1.f / (3.f * (float)HW), inv_dep = 1.f / (float)HW;
  hipLaunchKernelGGL(mapping_loss_final_kernel, dim3(nviews), dim3(256), 0, st, tab, nparts, inv_rgb, inv_dep, alpha);
}

void launch_mapping_loss(const LossTab& tab, int nviews, int HW, float alpha, float thr, float upstream, hipStream_t st) {
  int blocks = loss_blocks(HW);
  float inv_rgb = 1.f / (3.f * (float)HW), inv_dep = 1.f / (float)HW;
  hipLaunchKernelGGL(mapping_loss_kernel, dim3(blocks, nviews), dim3(256), 0, st, tab, HW, upstream * alpha * inv_rgb,
                     upstream * (1.f - alpha) * inv_dep, thr);
  hipLaunchKernelGGL(mapping_loss_final_kernel, dim3(nviews), dim3(256), 0, st, tab, blocks, inv_rgb, inv_dep, alpha);
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void __launch_bounds__(256) adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float b1, float b2,
                                                   float eps, float step_size, float bc2_sqrt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i];
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);                 // lerp_
    vi = vi * b2 + (1.f - b2) * gi * gi;              // mul_ + addcmul_
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

// up to kAdamMulti small tensors in ONE launch (block = tensor): the keyframe optimiser of the drop-in path holds ~20 one-element
// exposure parameters, for which torch's multi-tensor Adam issues ~12 launches and a per-tensor launch would issue 20
constexpr int kAdamMulti = 48;
struct AdamMulti {
  float* p[kAdamMulti]; const float* g[kAdamMulti]; float* m[kAdamMulti]; float* v[kAdamMulti];
  int32_t n[kAdamMulti]; float step_size[kAdamMulti], bc2_sqrt[kAdamMulti];
};
__global__ void __launch_bounds__(256) adam_multi_kernel(AdamMulti t, float b1, float b2, float eps) {
  const int k = blockIdx.x;
  float* __restrict__ p = t.p[k]; const float* __restrict__ g = t.g[k]; float* __restrict__ m = t.m[k]; float* __restrict__ v = t.v[k];
  const float step_size = t.step_size[k], bc2_sqrt = t.bc2_sqrt[k];
  for (int i = threadIdx.x; i < t.n[k]; i += blockDim.x) {
    float gi = g[i];
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

__device__ __forceinline__ void masked_adam_row(int r, int width, float* __restrict__ p, const float* __restrict__ g,
                                                float* __restrict__ m, float* __restrict__ v, int32_t* __restrict__ step,
                                                float lr, float b1, float b2, float eps) {
#pragma clang fp contract(off)
  int st = step[r] + 1;
  step[r] = st;
  float bc1 = 1.f - powf(b1, (float)st), bc2 = 1.f - powf(b2, (float)st);
  float step_size = lr / bc1, bc2s = sqrtf(bc2);
  for (int k = 0; k < width; ++k) {
    int i = r * width + k;
    float gi = g[i], mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void masked_adam_kernel(int rows, int width, float* __restrict__ p, const float* __restrict__ g,
                                   float* __restrict__ m, float* __restrict__ v, int32_t* __restrict__ step,
                                   const int32_t* __restrict__ active, float lr, float b1, float b2, float eps) {
#pragma clang fp contract(off)
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows || !active[r]) return;
  masked_adam_row(r, width, p, g, m, v, step, lr, b1, b2, eps);
}

// ------------------------------------------------------------------------------------------------ fused map step
__global__ void __launch_bounds__(256) activate_kernel(int64_t n, const float* __restrict__ scaling,
                                                       const float* __restrict__ rotation, const float* __restrict__ opacity,
                                                       float* __restrict__ s_out, float* __restrict__ r_out,
                                                       float* __restrict__ o_out) {
#pragma clang fp contract(off)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (s_out) {
    const F3 sc = ld3(scaling + 3 * i);
    st3(s_out + 3 * i, F3{expf(sc.x), expf(sc.y), expf(sc.z)});
  }
  if (r_out) {
    float4 q = *(const float4*)(rotation + 4 * i);
    float nn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    *(float4*)(r_out + 4 * i) = make_float4(q.x / nn, q.y / nn, q.z / nn, q.w / nn);
  }
  if (o_out) o_out[i] = 1.f / (1.f + expf(-opacity[i]));
}

#define adam_update(p, g, m, v, GRP, c)                      \
  do {                                                       \
    m = m + (g - m) * (1.f - c.b1);                          \
    v = v * c.b2 + (1.f - c.b2) * g * g;                     \
    float denom_ = sqrtf(v) / c.bc2_sqrt[GRP] + c.eps;       \
    p = p - c.step_size[GRP] * (m / denom_);                 \
  } while (0)

// MODE 0: gradient = sink (read, then zeroed)            -- sgr_gaussian_adam_step
// MODE 1: gradient = sink + gathered (sink zeroed)        -- fused tail, sinks may hold earlier contributions
// MODE 2: gradient = gathered (sinks known to be zero)    -- fused tail, steady state: the sinks are never touched
template <int MODE>
__device__ __forceinline__ float take_grad(float* __restrict__ sink, int64_t j, float gathered) {
  if (MODE == 2) return gathered;
  float g = sink[j];
  sink[j] = 0.f;
  return MODE == 1 ? g + gathered : g;
}

template <int MODE>
__device__ __forceinline__ void gaussian_adam_one(int64_t i, const AdamGroups& G, const AdamConst& c, float iso_coef,
                                                  const float* e /*[14] gathered or zeros*/, float* __restrict__ s_out,
                                                  float* __restrict__ r_out, float* __restrict__ o_out) {
  // every product and sum below is rounded on its own: the three instantiations (and the separate activate_kernel)
  // must agree bit for bit, which fused multiply-adds chosen per instantiation would not guarantee
#pragma clang fp contract(off)
  // xyz and f_dc: identity activations.  (3-float rows move as one 12-byte access per array: three dword accesses with a
  // 12-byte lane stride use a third of every cache line they touch)
#pragma unroll
  for (int grp = 0; grp < 2; ++grp) {
    const SgrAdamGroup& A = G.g[grp];
    if (i < G.r0[grp] || i >= G.r1[grp]) continue;
    float g[3];
    if (MODE == 2) {
      g[0] = e[3 * grp]; g[1] = e[3 * grp + 1]; g[2] = e[3 * grp + 2];
    } else {
      F3 gs = ld3(A.grad + 3 * i);
      st3(A.grad + 3 * i, F3{0.f, 0.f, 0.f});
      g[0] = gs.x; g[1] = gs.y; g[2] = gs.z;
      if (MODE == 1) { g[0] = g[0] + e[3 * grp]; g[1] = g[1] + e[3 * grp + 1]; g[2] = g[2] + e[3 * grp + 2]; }
    }
    if (A.skip) continue;
    F3 p3 = ld3(A.param + 3 * i), m3 = ld3(A.exp_avg + 3 * i), v3 = ld3(A.exp_avg_sq + 3 * i);
    float p[3] = {p3.x, p3.y, p3.z}, m[3] = {m3.x, m3.y, m3.z}, v[3] = {v3.x, v3.y, v3.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) adam_update(p[k], g[k], m[k], v[k], grp, c);
    st3(A.param + 3 * i, F3{p[0], p[1], p[2]});
    st3(A.exp_avg + 3 * i, F3{m[0], m[1], m[2]});
    st3(A.exp_avg_sq + 3 * i, F3{v[0], v[1], v[2]});
  }
  if (i >= G.r0[2] && i < G.r1[2]) {   // opacity: sigmoid
    const SgrAdamGroup& A = G.g[2];
    float p = A.param[i];
    float sg = 1.f / (1.f + expf(-p));
    float g = take_grad<MODE>(A.grad, i, e[6]) * sg * (1.f - sg);
    if (!A.skip) {
      float m = A.exp_avg[i], v = A.exp_avg_sq[i];
      adam_update(p, g, m, v, 2, c);
      A.param[i] = p; A.exp_avg[i] = m; A.exp_avg_sq[i] = v;
    }
    if (o_out) o_out[i] = 1.f / (1.f + expf(-p));
  }
  if (i >= G.r0[3] && i < G.r1[3]) {   // scaling: exp, plus d/ds of iso_weight * mean_{N,3} |s - mean_3(s)|
    const SgrAdamGroup& A = G.g[3];
    const F3 p3 = ld3(A.param + 3 * i);
    float p[3] = {p3.x, p3.y, p3.z}, s[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = expf(p[k]);
    float mean = (s[0] + s[1] + s[2]) / 3.f;
    float sg[3], ssum = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { float d = s[k] - mean; sg[k] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); ssum += sg[k]; }
    float gin[3];
    if (MODE == 2) {
      gin[0] = e[7]; gin[1] = e[8]; gin[2] = e[9];
    } else {
      F3 gs = ld3(A.grad + 3 * i);
      st3(A.grad + 3 * i, F3{0.f, 0.f, 0.f});
      gin[0] = gs.x; gin[1] = gs.y; gin[2] = gs.z;
      if (MODE == 1) { gin[0] = gin[0] + e[7]; gin[1] = gin[1] + e[8]; gin[2] = gin[2] + e[9]; }
    }
    if (!A.skip) {
      F3 m3 = ld3(A.exp_avg + 3 * i), v3 = ld3(A.exp_avg_sq + 3 * i);
      float m[3] = {m3.x, m3.y, m3.z}, v[3] = {v3.x, v3.y, v3.z};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float gs = gin[k] + iso_coef * (sg[k] - ssum / 3.f);
        float g = gs * s[k];
        adam_update(p[k], g, m[k], v[k], 3, c);
      }
      st3(A.param + 3 * i, F3{p[0], p[1], p[2]});
      st3(A.exp_avg + 3 * i, F3{m[0], m[1], m[2]});
      st3(A.exp_avg_sq + 3 * i, F3{v[0], v[1], v[2]});
    }
    if (s_out) st3(s_out + 3 * i, F3{expf(p[0]), expf(p[1]), expf(p[2])});
  }
  if (i >= G.r0[4] && i < G.r1[4]) {   // rotation: x / max(|x|, 1e-12)
    const SgrAdamGroup& A = G.g[4];
    float4 x = *(const float4*)(A.param + 4 * i);
    float4 gy;
    if (MODE == 2) {
      gy = make_float4(e[10], e[11], e[12], e[13]);
    } else {
      gy = *(const float4*)(A.grad + 4 * i);
      *(float4*)(A.grad + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) { gy.x += e[10]; gy.y += e[11]; gy.z += e[12]; gy.w += e[13]; }
    }
    float nrm = sqrtf(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
    float gx[4];
    if (nrm > 1e-12f) {
      float inv = 1.f / nrm;
      float y[4] = {x.x * inv, x.y * inv, x.z * inv, x.w * inv};
      float dot = y[0] * gy.x + y[1] * gy.y + y[2] * gy.z + y[3] * gy.w;
      gx[0] = (gy.x - y[0] * dot) * inv; gx[1] = (gy.y - y[1] * dot) * inv;
      gx[2] = (gy.z - y[2] * dot) * inv; gx[3] = (gy.w - y[3] * dot) * inv;
    } else {
      gx[0] = gy.x * 1e12f; gx[1] = gy.y * 1e12f; gx[2] = gy.z * 1e12f; gx[3] = gy.w * 1e12f;
    }
    float pp[4] = {x.x, x.y, x.z, x.w};
    if (!A.skip) {
      float4 m = *(const float4*)(A.exp_avg + 4 * i), v = *(const float4*)(A.exp_avg_sq + 4 * i);
      float mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) adam_update(pp[k], gx[k], mm[k], vv[k], 4, c);
      *(float4*)(A.param + 4 * i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *(float4*)(A.exp_avg + 4 * i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *(float4*)(A.exp_avg_sq + 4 * i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    if (r_out) {      // same expression as activate_kernel
      float nn = fmaxf(sqrtf(pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2] + pp[3] * pp[3]), 1e-12f);
      *(float4*)(r_out + 4 * i) = make_float4(pp[0] / nn, pp[1] / nn, pp[2] / nn, pp[3] / nn);
    }
  }
}

__global__ void __launch_bounds__(256) gaussian_adam_kernel(int64_t n, AdamGroups G, AdamConst c, float iso_coef,
                                                            float* __restrict__ s_out, float* __restrict__ r_out,
                                                            float* __restrict__ o_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float zero[14] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  gaussian_adam_one<0>(i, G, c, iso_coef, zero, s_out, r_out, o_out);
}

// The single-GPU tail of a mapping iteration in one pass: thread = Gaussian; adds up its gradient records over the
// views of the batch in view order (exactly grad_gather_kernel's sum), then the Adam step of all five groups, then the
// activations the next iteration renders with.  MODE 3 (multi-GPU) stops after the sum: it is added to the gradient
// buffers (what grad_gather_kernel does with accumulate = 1), the riders still do the loss sums + exposure step; MODE 4
// stores it instead (for every Gaussian), so that the flat buffer the ranks exchange is never zeroed in between.
template <int MODE>
__global__ void __launch_bounds__(256) gather_adam_kernel(ViewTab tab, int nviews, LOff L, FusedAdam fa) {
  if ((int)blockIdx.x < fa.tail_views) {      // rider blocks (scheduled first): one per view
    // fixed-order sum of the view's per-tile loss parts, then the exposure (keyframe) Adam step of the ONE slab row this
    // view's exposure gradient lives in (every active row belongs to a window keyframe, and those are rendered every
    // iteration: mapper.py:426-447, 1096-1111) -- no dependency on any other block
    __shared__ LossPart red[4];
    const int v = blockIdx.x;
    loss_final_view((const LossPart*)fa.tail_parts[v], fa.tail_nparts, fa.tail_inv_rgb, fa.tail_inv_dep, fa.tail_alpha,
                    fa.tail_loss[v], fa.tail_da[v], fa.tail_db[v], red);
    // (a view whose forward ran out of pair capacity rendered truncated lists: it takes no part in this step, see below)
    if (threadIdx.x == 0 && fa.exp_rows > 0 && fa.tail_da[v] && ((const SavedHeader*)tab.saved[v])->overflow == 0u) {
      const ptrdiff_t d = fa.tail_da[v] - fa.exp_grad;
      if (d >= 0 && d < (ptrdiff_t)fa.exp_rows * fa.exp_width && d % fa.exp_width == 0) {
        const int r = (int)(d / fa.exp_width);
        if (fa.exp_active[r])
          masked_adam_row(r, fa.exp_width, fa.exp_param, fa.exp_grad, fa.exp_avg, fa.exp_avg_sq, fa.exp_step, fa.exp_lr,
                          fa.exp_b1, fa.exp_b2, fa.exp_eps);
      }
    }
    return;
  }
  // per-view pointers into LDS: the gather below indexes them per LANE (a by-value kernarg table cannot be)
  __shared__ const char* s_scratch[kMaxViews];
  __shared__ const int32_t* s_radii[kMaxViews];
  __shared__ uint32_t s_over[kMaxViews];
#pragma unroll
  for (int u = 0; u < kMaxViews; ++u)
    if ((int)threadIdx.x == u && u < nviews) {
      s_scratch[u] = tab.scratch[u];
      s_radii[u] = tab.radii[u];
      s_over[u] = ((const SavedHeader*)tab.saved[u])->overflow;
    }
  __syncthreads();
  // Capacity overflow (R > cap): the view's lists were truncated and some of its partial slots were never written.  Such a
  // view contributes NOTHING to this step -- no gradient, no densification statistics, no exposure update -- instead of
  // feeding uninitialised sums into the Adam moments; the host sees header.overflow at its next check and grows the capacity.
  uint32_t truncated = 0u;
  for (int u = 0; u < nviews; ++u) truncated |= (s_over[u] != 0u ? 1u : 0u) << u;
  const int i = ((int)blockIdx.x - fa.tail_views) * blockDim.x + threadIdx.x;
  if (i >= L.N) return;
  float a[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) a[k] = 0.f;
  float st_norm = 0.f, st_cnt = 0.f, st_maxr = 0.f;
  // K1 left one word per Gaussian with the views of the batch that see it.  Every lane walks ITS OWN set bits in
  // ascending view order (the fixed summation order of grad_gather_kernel): a wave makes max-over-lanes(popcount) record
  // round trips -- about 3 for a SLAM batch -- instead of one per view of the batch.
  uint32_t seen = 0u;
  for (int p = 0; p < min(L.k1_parts, nviews); ++p) seen |= ((const uint32_t*)(tab.saved[0] + L.o_vismask))[(size_t)p * L.N + i];
  seen &= ~truncated;
  const bool any = seen != 0u;
  while (seen) {
    const int v = __builtin_ctz(seen);
    seen &= seen - 1u;
    const float4* rec = (const float4*)(s_scratch[v] + L.o_gradrec) + (size_t)i * 4;
    const int r = s_radii[v][i];
    float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    a[0] += r0.x; a[1] += r0.y; a[2] += r0.z; a[3] += r0.w; a[4] += r1.x; a[5] += r1.y; a[6] += r1.z; a[7] += r1.w;
    a[8] += r2.x; a[9] += r2.y; a[10] += r2.z; a[11] += r2.w; a[12] += r3.x; a[13] += r3.y;
    st_norm += sqrtf(r3.z * r3.z + r3.w * r3.w);
    st_cnt += 1.f;
    st_maxr = fmaxf(st_maxr, (float)r);
  }
  if (any && fa.stat_accum) {
    fa.stat_accum[i] += st_norm;
    fa.stat_denom[i] += st_cnt;
    fa.stat_maxr[i] = fmaxf(fa.stat_maxr[i], st_maxr);
  }
  if (MODE == 3) {          // multi-GPU iteration: the sums go into the flat gradient buffer the ranks all-reduce
    if (!any) return;
    float* g0 = fa.G.g[0].grad + 3 * (size_t)i; float* g1 = fa.G.g[1].grad + 3 * (size_t)i; float* g3 = fa.G.g[3].grad + 3 * (size_t)i;
    float* g4 = fa.G.g[4].grad + 4 * (size_t)i;
    g0[0] += a[0]; g0[1] += a[1]; g0[2] += a[2];
    g1[0] += a[3]; g1[1] += a[4]; g1[2] += a[5];
    fa.G.g[2].grad[i] += a[6];
    g3[0] += a[7]; g3[1] += a[8]; g3[2] += a[9];
    g4[0] += a[10]; g4[1] += a[11]; g4[2] += a[12]; g4[3] += a[13];
    return;
  }
  if (MODE == 4) {          // ... STORED, for every Gaussian (zeros where no view sees it): nobody has to zero the buffer in between
    st3(fa.G.g[0].grad + 3 * (size_t)i, F3{a[0], a[1], a[2]});
    st3(fa.G.g[1].grad + 3 * (size_t)i, F3{a[3], a[4], a[5]});
    fa.G.g[2].grad[i] = a[6];
    st3(fa.G.g[3].grad + 3 * (size_t)i, F3{a[7], a[8], a[9]});
    *(float4*)(fa.G.g[4].grad + 4 * (size_t)i) = make_float4(a[10], a[11], a[12], a[13]);
    return;
  }
  gaussian_adam_one<MODE >= 3 ? 1 : MODE>(i, fa.G, fa.c, fa.iso_coef, a, fa.s_out, fa.r_out, fa.o_out);
}

// the Adam step of all five groups; with output pointers also the activations the next forward renders with (the
// multi-GPU iteration: all-reduce, then this ONE pass instead of Adam + activate)
int gaussian_adam_step_act(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps, float iso_weight,
                           float* s_out, float* r_out, float* o_out, void* stream) {
  if (n < 0 || !groups) return set_error(SGR_ERR_INVALID, "gaussian_adam: bad argument");
  if (n == 0) return SGR_OK;
  FusedAdam fa;
  if (int rc = make_fused_adam(n, groups, beta1, beta2, eps, iso_weight, &fa)) return rc;
  hipLaunchKernelGGL(gaussian_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, fa.G, fa.c,
                     fa.iso_coef, s_out, r_out, o_out);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "gaussian_adam launch failed");
}

void launch_gather_adam(const ViewTab& tab, int nviews, const LOff& L, const FusedAdam& fa, hipStream_t st) {
  if (L.N <= 0) return;
  const int grid = L.pre_blocks + fa.tail_views;
  if (fa.gather_only == 2)
    hipLaunchKernelGGL(gather_adam_kernel<4>, dim3(grid), dim3(256), 0, st, tab, nviews, L, fa);
  else if (fa.gather_only)
    hipLaunchKernelGGL(gather_adam_kernel<3>, dim3(grid), dim3(256), 0, st, tab, nviews, L, fa);
  else if (fa.grads_clean)
    hipLaunchKernelGGL(gather_adam_kernel<2>, dim3(grid), dim3(256), 0, st, tab, nviews, L, fa);
  else
    hipLaunchKernelGGL(gather_adam_kernel<1>, dim3(grid), dim3(256), 0, st, tab, nviews, L, fa);
}

// ------------------------------------------------------------------------------------------------ 3-NN
struct Top3 { float a, b, c; };
__device__ __forceinline__ void top3_push(Top3& t, float d) {
  if (d < t.c) {
    if (d < t.b) {
      t.c = t.b;
      if (d < t.a) { t.b = t.a; t.a = d; } else { t.b = d; }
    } else {
      t.c = d;
    }
  }
}

// grid = (query blocks, splits): each block scans one slice of the candidates through LDS for 256 queries
__global__ void __launch_bounds__(256) knn_partial_kernel(int n, const float* __restrict__ xyz, int per_split,
                                                          Top3* __restrict__ part) {
  __shared__ float sx[256], sy[256], sz[256];
  int i = blockIdx.x * 256 + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (i < n) { qx = xyz[3 * i]; qy = xyz[3 * i + 1]; qz = xyz[3 * i + 2]; }
  Top3 best = {INFINITY, INFINITY, INFINITY};
  int j0 = blockIdx.y * per_split, j1 = min(n, j0 + per_split);
  for (int base = j0; base < j1; base += 256) {
    int j = base + threadIdx.x;
    if (j < j1) { sx[threadIdx.x] = xyz[3 * j]; sy[threadIdx.x] = xyz[3 * j + 1]; sz[threadIdx.x] = xyz[3 * j + 2]; }
    __syncthreads();
    int cnt = min(256, j1 - base);
    for (int k = 0; k < cnt; ++k) {
      float dx = sx[k] - qx, dy = sy[k] - qy, dz = sz[k] - qz;
      float d = dx * dx + dy * dy + dz * dz;
      if (base + k != i) top3_push(best, d);
    }
    __syncthreads();
  }
  if (i < n) part[(size_t)blockIdx.y * n + i] = best;
}

__global__ void __launch_bounds__(256) knn_merge_kernel(int n, int splits, const Top3* __restrict__ part,
                                                        float* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Top3 best = {INFINITY, INFINITY, INFINITY};
  for (int s = 0; s < splits; ++s) {
    Top3 t = part[(size_t)s * n + i];
    top3_push(best, t.a); top3_push(best, t.b); top3_push(best, t.c);
  }
  // fewer than 4 points: missing neighbours count as 0 like an unfilled best-list would not -- keep finite
  float a = isinf(best.a) ? 0.f : best.a, b = isinf(best.b) ? 0.f : best.b, c = isinf(best.c) ? 0.f : best.c;
  out[i] = (a + b + c) / 3.f;
}

static int knn_splits(int n) {
  int qb = (n + 255) / 256;
  int s = (2048 + qb - 1) / qb;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  int max_s = (n + 255) / 256;      // at least one 256-chunk per split
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

// ---- large inputs: exact 3-NN through a uniform grid ------------------------------------------------------------------
// Brute force is the right tool for what the reference feeds distCUDA2 (the 6-13 k seeds of one keyframe: 40 us) and the
// wrong one for a whole map (300 k: 36 ms; 1.5 M: 0.9 s).  From kKnnGridFrom points on: bounding box -> cell size for ~8
// points per cell if the box were filled (a surface-like map ends up with a few dozen per occupied cell) -> counting sort
// into cells (order inside a cell is arbitrary, the result does not depend on it) -> every point searches the (2r+1)^3
// cells around its own, r = 1, 2, ... until its third-nearest distance is no larger than the distance to the searched
// box's faces (shrunk by the rounding of the cell coordinates), which proves that no unvisited point is nearer.
constexpr int kKnnGridFrom = 16384;
struct KnnGrid { float lo[3]; float h, inv_h, eps; int g[3]; int cells; };
__device__ __forceinline__ uint32_t knn_ordered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float knn_unordered(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
static size_t knn_max_cells(int n) { return (size_t)n / 2 + 1024; }

// bounding box of the points: per-lane running min / max -> wave (DPP-free xor shuffles) -> block through LDS -> ONE set of six
// atomics per block and at most 256 blocks.  (The first version sent six atomics per WAVE of 1024 blocks to the same six words:
// 24 576 serialised returning atomics = 282 us for 300 k points, a quarter of the whole distCUDA2.)
__global__ void __launch_bounds__(256) knn_bounds_kernel(int n, const float* __restrict__ xyz, uint32_t* __restrict__ mm /* min3, max3 (ordered) */) {
  __shared__ float red[4][6];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const F3 p = ld3(xyz + 3 * (size_t)i);
    lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
    lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
    lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    if (k < 3) atomicMin(&mm[k], knn_ordered(fminf(fminf(red[0][k], red[1][k]), fminf(red[2][k], red[3][k]))));
    else atomicMax(&mm[k], knn_ordered(fmaxf(fmaxf(red[0][k], red[1][k]), fmaxf(red[2][k], red[3][k]))));
  }
}

__global__ void knn_grid_setup_kernel(int n, int max_cells, const uint32_t* __restrict__ mm, KnnGrid* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  KnnGrid G;
  float ext[3], amax = 0.f;
  for (int k = 0; k < 3; ++k) {
    const float lo = knn_unordered(mm[k]), hi = knn_unordered(mm[3 + k]);
    G.lo[k] = lo;
    ext[k] = fmaxf(hi - lo, 0.f);
    amax = fmaxf(amax, fmaxf(fabsf(lo), fabsf(hi)));
  }
  const float emax = fmaxf(ext[0], fmaxf(ext[1], ext[2]));
  // volume of the box with degenerate sides lifted to 1/1024 of the longest one (planar / linear maps)
  float vol = 1.f;
  for (int k = 0; k < 3; ++k) vol *= fmaxf(ext[k], emax * (1.f / 1024.f));
  float h = emax > 0.f ? cbrtf(vol * 8.f / (float)n) : 1.f;
  h = fmaxf(h, emax * (1.f / 1024.f));
  if (!(h > 0.f) || !isfinite(h)) h = 1.f;
  for (int iter = 0; iter < 32; ++iter) {
    long long cells = 1;
    for (int k = 0; k < 3; ++k) { G.g[k] = min(1024, max(1, (int)floorf(ext[k] / h) + 1)); cells *= G.g[k]; }
    if (cells <= (long long)max_cells) break;
    h *= 1.26f;                              // (cells shrink by ~2 per step)
  }
  G.cells = G.g[0] * G.g[1] * G.g[2];
  G.h = h; G.inv_h = 1.f / h;
  G.eps = 8e-6f * (amax + emax) + 1e-30f;    // what fp32 rounding can move a cell coordinate by, in length units
  *out = G;
}

__device__ __forceinline__ int knn_cell_coord(float v, float lo, float inv_h, int g) { return min(g - 1, max(0, (int)floorf((v - lo) * inv_h))); }

__global__ void __launch_bounds__(256) knn_count_kernel(int n, const float* __restrict__ xyz, const KnnGrid* __restrict__ Gp,
                                                        uint32_t* __restrict__ count, uint32_t* __restrict__ cell_of) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KnnGrid G = *Gp;
  const int cx = knn_cell_coord(xyz[3 * i], G.lo[0], G.inv_h, G.g[0]), cy = knn_cell_coord(xyz[3 * i + 1], G.lo[1], G.inv_h, G.g[1]),
            cz = knn_cell_coord(xyz[3 * i + 2], G.lo[2], G.inv_h, G.g[2]);
  const uint32_t c = (uint32_t)((cz * G.g[1] + cy) * G.g[0] + cx);
  cell_of[i] = c;
  atomicAdd(&count[c], 1u);
}

// exclusive scan of count[0..cells) by one 1024-thread block (cells <= n/2 + 1024: ~50 passes at 300 k points)
__global__ void __launch_bounds__(1024) knn_scan_kernel(const KnnGrid* __restrict__ Gp, const uint32_t* __restrict__ count,
                                                        uint32_t* __restrict__ start, uint32_t* __restrict__ cursor) {
  __shared__ uint32_t red[16];
  const int cells = Gp->cells, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int base = 0; base < cells; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const uint32_t v = i < cells ? count[i] : 0u;
    const uint32_t inc = wave_scan_add_u32(v);
    if (lane == 63) red[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint32_t r = red[w]; pre += w < wv ? r : 0u; tot += r; }
    __syncthreads();
    if (i < cells) { start[i] = carry + pre + inc - v; cursor[i] = carry + pre + inc - v; }
    carry += tot;
  }
}

__global__ void __launch_bounds__(256) knn_fill_kernel(int n, const float* __restrict__ xyz, const uint32_t* __restrict__ cell_of,
                                                       uint32_t* __restrict__ cursor, float* __restrict__ sorted_xyz,
                                                       uint32_t* __restrict__ sorted_idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t pos = atomicAdd(&cursor[cell_of[i]], 1u);
  sorted_xyz[3 * pos] = xyz[3 * i]; sorted_xyz[3 * pos + 1] = xyz[3 * i + 1]; sorted_xyz[3 * pos + 2] = xyz[3 * i + 2];
  sorted_idx[pos] = (uint32_t)i;
}

// thread = point in CELL order (neighbouring lanes search the same cells)
__global__ void __launch_bounds__(256) knn_query_kernel(int n, const KnnGrid* __restrict__ Gp, const uint32_t* __restrict__ start,
                                                        const uint32_t* __restrict__ count, const float* __restrict__ sorted_xyz,
                                                        const uint32_t* __restrict__ sorted_idx, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const KnnGrid G = *Gp;
  const float q[3] = {sorted_xyz[3 * t], sorted_xyz[3 * t + 1], sorted_xyz[3 * t + 2]};
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = knn_cell_coord(q[k], G.lo[k], G.inv_h, G.g[k]);
  Top3 best;
  for (int r = 1;; ++r) {
    best = {INFINITY, INFINITY, INFINITY};
    const int x0 = max(0, c[0] - r), x1 = min(G.g[0] - 1, c[0] + r), y0 = max(0, c[1] - r), y1 = min(G.g[1] - 1, c[1] + r),
              z0 = max(0, c[2] - r), z1 = min(G.g[2] - 1, c[2] + r);
    for (int cz = z0; cz <= z1; ++cz)
      for (int cy = y0; cy <= y1; ++cy) {
        const int row = (cz * G.g[1] + cy) * G.g[0];
        const uint32_t j0 = start[row + x0], j1 = start[row + x1] + count[row + x1];     // cells of one row are consecutive
        for (uint32_t j = j0; j < j1; ++j) {
          const float dx = sorted_xyz[3 * j] - q[0], dy = sorted_xyz[3 * j + 1] - q[1], dz = sorted_xyz[3 * j + 2] - q[2];
          const float d = dx * dx + dy * dy + dz * dz;
          if ((int)j != t) top3_push(best, d);
        }
      }
    // faces of the searched box that are not faces of the grid: anything beyond them is unvisited
    float bound = INFINITY;
    const int lo_c[3] = {x0, y0, z0}, hi_c[3] = {x1, y1, z1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (lo_c[k] > 0) bound = fminf(bound, q[k] - (G.lo[k] + (float)lo_c[k] * G.h));
      if (hi_c[k] < G.g[k] - 1) bound = fminf(bound, (G.lo[k] + (float)(hi_c[k] + 1) * G.h) - q[k]);
    }
    if (isinf(bound)) break;                         // the whole grid has been searched
    const float safe = bound - G.eps;
    if (safe > 0.f && best.c <= safe * safe) break;
  }
  const float a = isinf(best.a) ? 0.f : best.a, b = isinf(best.b) ? 0.f : best.b, cc = isinf(best.c) ? 0.f : best.c;
  out[sorted_idx[t]] = (a + b + cc) / 3.f;
}

static size_t knn_grid_scratch_bytes(int n) {
  const size_t mc = knn_max_cells(n);
  return 256 + 256 + 3 * align_up(mc * 4) + align_up((size_t)n * 4) + align_up((size_t)n * 12) + align_up((size_t)n * 4);
}

// ------------------------------------------------------------------------------------------------ SE3
struct Pose { float t[3]; float q[4]; };   // q = (x, y, z, w)
__device__ __forceinline__ Pose load_pose(const float* p) { return {{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }
__device__ __forceinline__ void store_pose(float* p, const Pose& a) {
  p[0] = a.t[0]; p[1] = a.t[1]; p[2] = a.t[2]; p[3] = a.q[0]; p[4] = a.q[1]; p[5] = a.q[2]; p[6] = a.q[3];
}
__device__ __forceinline__ void quat_rotate(const float q[4], const float v[3], float out[3]) {
  // v + 2w (u x v) + 2 u x (u x v),  u = (x,y,z)
  float ux = q[0], uy = q[1], uz = q[2], w = q[3];
  float cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
  float dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
  out[0] = v[0] + 2.f * (w * cx + dx);
  out[1] = v[1] + 2.f * (w * cy + dy);
  out[2] = v[2] + 2.f * (w * cz + dz);
}
__device__ __forceinline__ void quat_mul(const float a[4], const float b[4], float o[4]) {
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

enum Se3Op { OP_EXP, OP_LOG, OP_INV, OP_MUL, OP_ACT, OP_ADJT, OP_MATRIX };

template <int OP>
__global__ void __launch_bounds__(256) se3_kernel(int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (OP == OP_EXP) {
    const float* tau = a + 6 * i;
    float rho[3] = {tau[0], tau[1], tau[2]}, th[3] = {tau[3], tau[4], tau[5]};
    float t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
    float ang = sqrtf(t2);
    float A, Bc, Cc, imag, real;   // sin(a)/a, (1-cos a)/a^2, (a - sin a)/a^3, sin(a/2)/a, cos(a/2)
    if (ang < 1e-4f) {
      A = 1.f - t2 / 6.f; Bc = 0.5f - t2 / 24.f; Cc = 1.f / 6.f - t2 / 120.f;
      imag = 0.5f - t2 / 48.f; real = 1.f - t2 / 8.f;
    } else {
      float s = sinf(ang), c = cosf(ang);
      A = s / ang; Bc = (1.f - c) / t2; Cc = (ang - s) / (t2 * ang);
      imag = sinf(0.5f * ang) / ang; real = cosf(0.5f * ang);
    }
    (void)A;
    // t = V rho,  V = I + B [th]x + C [th]x^2
    float c1[3] = {th[1] * rho[2] - th[2] * rho[1], th[2] * rho[0] - th[0] * rho[2], th[0] * rho[1] - th[1] * rho[0]};
    float c2[3] = {th[1] * c1[2] - th[2] * c1[1], th[2] * c1[0] - th[0] * c1[2], th[0] * c1[1] - th[1] * c1[0]};
    Pose p;
    for (int k = 0; k < 3; ++k) p.t[k] = rho[k] + Bc * c1[k] + Cc * c2[k];
    p.q[0] = imag * th[0]; p.q[1] = imag * th[1]; p.q[2] = imag * th[2]; p.q[3] = real;
    store_pose(out + 7 * i, p);
  } else if (OP == OP_LOG) {
    Pose p = load_pose(a + 7 * i);
    float n2 = p.q[0] * p.q[0] + p.q[1] * p.q[1] + p.q[2] * p.q[2];
    float nn = sqrtf(n2), w = p.q[3];
    float k;   // theta = k * (x,y,z)
    if (nn < 1e-6f) {
      k = 2.f / w - 2.f * n2 / (3.f * w * w * w);
    } else {
      // atan(n/w) with the branch that keeps the angle in (-pi, pi]
      float half = (fabsf(w) < 1e-12f) ? (w >= 0.f ? 1.f : -1.f) * 1.57079632679f : atanf(nn / w);
      k = 2.f * half / nn;
    }
    float th[3] = {k * p.q[0], k * p.q[1], k * p.q[2]};
    float t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2], ang = sqrtf(t2);
    // rho = V^-1 t,  V^-1 = I - 1/2 [th]x + D [th]x^2,  D = (1 - (a/2) cot(a/2)) / a^2
    float D;
    if (ang < 1e-4f) D = 1.f / 12.f + t2 / 720.f;
    else { float h = 0.5f * ang; D = (1.f - h * cosf(h) / sinf(h)) / t2; }
    float c1[3] = {th[1] * p.t[2] - th[2] * p.t[1], th[2] * p.t[0] - th[0] * p.t[2], th[0] * p.t[1] - th[1] * p.t[0]};
    float c2[3] = {th[1] * c1[2] - th[2] * c1[1], th[2] * c1[0] - th[0] * c1[2], th[0] * c1[1] - th[1] * c1[0]};
    float* o = out + 6 * i;
    for (int k2 = 0; k2 < 3; ++k2) { o[k2] = p.t[k2] - 0.5f * c1[k2] + D * c2[k2]; o[3 + k2] = th[k2]; }
  } else if (OP == OP_INV) {
    Pose p = load_pose(a + 7 * i), r;
    r.q[0] = -p.q[0]; r.q[1] = -p.q[1]; r.q[2] = -p.q[2]; r.q[3] = p.q[3];
    float rt[3];
    quat_rotate(r.q, p.t, rt);
    r.t[0] = -rt[0]; r.t[1] = -rt[1]; r.t[2] = -rt[2];
    store_pose(out + 7 * i, r);
  } else if (OP == OP_MUL) {
    Pose x = load_pose(a + 7 * i), y = load_pose(b + 7 * i), r;
    quat_mul(x.q, y.q, r.q);
    float rt[3];
    quat_rotate(x.q, y.t, rt);
    r.t[0] = x.t[0] + rt[0]; r.t[1] = x.t[1] + rt[1]; r.t[2] = x.t[2] + rt[2];
    store_pose(out + 7 * i, r);
  } else if (OP == OP_ACT) {
    Pose x = load_pose(a + 7 * i);
    float v[3] = {b[3 * i], b[3 * i + 1], b[3 * i + 2]}, r[3];
    quat_rotate(x.q, v, r);
    out[3 * i] = r[0] + x.t[0]; out[3 * i + 1] = r[1] + x.t[1]; out[3 * i + 2] = r[2] + x.t[2];
  } else if (OP == OP_ADJT) {
    Pose x = load_pose(a + 7 * i);
    const float* v = b + 6 * i;
    float ar[3] = {v[0], v[1], v[2]}, at[3] = {v[3], v[4], v[5]};
    float qi[4] = {-x.q[0], -x.q[1], -x.q[2], x.q[3]};
    float txa[3] = {x.t[1] * ar[2] - x.t[2] * ar[1], x.t[2] * ar[0] - x.t[0] * ar[2], x.t[0] * ar[1] - x.t[1] * ar[0]};
    float m[3] = {at[0] - txa[0], at[1] - txa[1], at[2] - txa[2]};
    float o1[3], o2[3];
    quat_rotate(qi, ar, o1);
    quat_rotate(qi, m, o2);
    float* o = out + 6 * i;
    o[0] = o1[0]; o[1] = o1[1]; o[2] = o1[2]; o[3] = o2[0]; o[4] = o2[1]; o[5] = o2[2];
  } else {   // OP_MATRIX
    Pose x = load_pose(a + 7 * i);
    float ex[3] = {1.f, 0.f, 0.f}, ey[3] = {0.f, 1.f, 0.f}, ez[3] = {0.f, 0.f, 1.f}, c0[3], c1[3], c2[3];
    quat_rotate(x.q, ex, c0); quat_rotate(x.q, ey, c1); quat_rotate(x.q, ez, c2);
    float* o = out + 16 * i;
    for (int r = 0; r < 3; ++r) { o[4 * r] = c0[r]; o[4 * r + 1] = c1[r]; o[4 * r + 2] = c2[r]; o[4 * r + 3] = x.t[r]; }
    o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
  }
}

template <int OP>
static int se3_launch(int64_t n, const float* a, const float* b, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!a || !out))) return set_error(SGR_ERR_INVALID, "se3: null argument");
  if (n == 0) return SGR_OK;
  hipLaunchKernelGGL(se3_kernel<OP>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, a, b, out);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "se3 launch failed");
}

int make_fused_adam(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps, float iso_weight, FusedAdam* out) {
  FusedAdam& fa = *out;
  fa = FusedAdam{};
  fa.c.b1 = beta1; fa.c.b2 = beta2; fa.c.eps = eps;
  for (int k = 0; k < 5; ++k) {
    fa.G.g[k] = groups[k];
    fa.G.r0[k] = 0;
    fa.G.r1[k] = INT64_MAX;
    const SgrAdamGroup& g = fa.G.g[k];
    if (!g.grad || !g.param || (!g.skip && (!g.exp_avg || !g.exp_avg_sq || g.step < 1)))
      return set_error(SGR_ERR_INVALID, "gaussian_adam: null pointer / bad step in group %d", k);
    int64_t st = g.step < 1 ? 1 : g.step;
    double bc1 = 1.0 - std::pow((double)beta1, (double)st), bc2 = 1.0 - std::pow((double)beta2, (double)st);
    fa.c.bc2_sqrt[k] = (float)std::sqrt(bc2);
    fa.c.step_size[k] = (float)((double)g.lr / bc1);
  }
  fa.iso_coef = iso_weight / (3.f * (float)n);
  return SGR_OK;
}

}  // namespace sgr

using namespace sgr;

extern "C" {

int sgr_mapping_loss(int32_t H, int32_t W, const float* image, const float* depth, const float* gt_image,
                     const float* gt_depth, const float* exposure_a, const float* exposure_b, float alpha,
                     float rgb_boundary_threshold, float upstream, float* loss, float* dL_dimage, float* dL_ddepth,
                     float* dL_dexp_a, float* dL_dexp_b, void* scratch, size_t scratch_bytes, void* stream) {
  if (H <= 0 || W <= 0 || !image || !depth || !gt_image || !gt_depth) return set_error(SGR_ERR_INVALID, "mapping_loss: null/size");
  const int HW = H * W;
  if (!scratch || scratch_bytes < (size_t)loss_blocks(HW) * sizeof(LossPart))
    return set_error(SGR_ERR_WORKSPACE, "mapping_loss scratch too small (need %zu)", (size_t)1024 * sizeof(LossPart));
  LossTab tab = {};
  tab.image[0] = image; tab.depth[0] = depth; tab.gt_image[0] = gt_image; tab.gt_depth[0] = gt_depth;
  tab.exp_a[0] = exposure_a; tab.exp_b[0] = exposure_b; tab.loss[0] = loss; tab.dimage[0] = dL_dimage;
  tab.ddepth[0] = dL_ddepth; tab.da[0] = dL_dexp_a; tab.db[0] = dL_dexp_b; tab.parts[0] = scratch;
  launch_mapping_loss(tab, 1, HW, alpha, rgb_boundary_threshold, upstream, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "mapping_loss launch failed");
}

int sgr_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                  float beta2, float eps, int64_t step, void* stream) {
  if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return set_error(SGR_ERR_INVALID, "adam: bad argument");
  if (n == 0) return SGR_OK;
  double bc1 = 1.0 - std::pow((double)beta1, (double)step);
  double bc2 = 1.0 - std::pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float bc2_sqrt = (float)std::sqrt(bc2);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq,
                     beta1, beta2, eps, step_size, bc2_sqrt);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "adam launch failed");
}

int sgr_adam_step_multi(int32_t count, const SgrAdamTensor* tensors, float lr, float beta1, float beta2, float eps, void* stream) {
  if (count < 0 || (count > 0 && !tensors)) return set_error(SGR_ERR_INVALID, "adam_multi: bad argument");
  for (int base = 0; base < count; base += kAdamMulti) {
    const int nb = count - base < kAdamMulti ? count - base : kAdamMulti;
    AdamMulti t = {};
    for (int k = 0; k < nb; ++k) {
      const SgrAdamTensor& a = tensors[base + k];
      if (a.n < 0 || a.n > (1 << 20) || a.step < 1 || (a.n > 0 && (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq)))
        return set_error(SGR_ERR_INVALID, "adam_multi: bad tensor %d", base + k);
      const double bc1 = 1.0 - std::pow((double)beta1, (double)a.step), bc2 = 1.0 - std::pow((double)beta2, (double)a.step);
      t.p[k] = a.param; t.g[k] = a.grad; t.m[k] = a.exp_avg; t.v[k] = a.exp_avg_sq; t.n[k] = (int32_t)a.n;
      t.step_size[k] = (float)((double)lr / bc1); t.bc2_sqrt[k] = (float)std::sqrt(bc2);
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, t, beta1, beta2, eps);
  }
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "adam_multi launch failed");
}

int sgr_activate(int64_t n, const float* scaling, const float* rotation, const float* opacity, float* scales_out,
                 float* rot_out, float* opac_out, void* stream) {
  if (n < 0 || (scales_out && !scaling) || (rot_out && !rotation) || (opac_out && !opacity))
    return set_error(SGR_ERR_INVALID, "activate: null input for a requested output");
  if (n == 0) return SGR_OK;
  hipLaunchKernelGGL(activate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, scaling, rotation,
                     opacity, scales_out, rot_out, opac_out);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "activate launch failed");
}

int sgr_gaussian_adam_step(int64_t n, const SgrAdamGroup groups[5], float beta1, float beta2, float eps, float iso_weight,
                           void* stream) {
  return gaussian_adam_step_act(n, groups, beta1, beta2, eps, iso_weight, nullptr, nullptr, nullptr, stream);
}

int sgr_gaussian_adam_shard(int64_t n_total, const SgrAdamGroup groups[5], const int64_t row0[5], const int64_t row1[5],
                            float beta1, float beta2, float eps, float iso_weight, void* stream) {
  if (n_total < 0 || !groups || !row0 || !row1) return set_error(SGR_ERR_INVALID, "gaussian_adam_shard: bad argument");
  SgrAdamGroup g[5];
  int64_t lo = INT64_MAX, hi = 0;
  for (int k = 0; k < 5; ++k) {
    g[k] = groups[k];
    if (row0[k] < 0 || row1[k] > n_total) return set_error(SGR_ERR_INVALID, "gaussian_adam_shard: rows of group %d outside [0, n]", k);
    if (row1[k] <= row0[k]) { g[k].skip = 1; if (!g[k].param) g[k].param = (float*)8; if (!g[k].grad) g[k].grad = (float*)8; continue; }
    lo = row0[k] < lo ? row0[k] : lo;
    hi = row1[k] > hi ? row1[k] : hi;
  }
  if (hi <= lo) return SGR_OK;
  FusedAdam fa;
  if (int rc = make_fused_adam(n_total, g, beta1, beta2, eps, iso_weight, &fa)) return rc;
  for (int k = 0; k < 5; ++k) { fa.G.r0[k] = row0[k] - lo; fa.G.r1[k] = (row1[k] > row0[k] ? row1[k] : row0[k]) - lo; }
  // thread t works on Gaussian lo + t: shift every base pointer instead of the index
  const int w[5] = {3, 3, 1, 3, 4};
  for (int k = 0; k < 5; ++k) {
    SgrAdamGroup& A = fa.G.g[k];
    A.param += lo * w[k]; A.grad += lo * w[k];
    if (A.exp_avg) A.exp_avg += lo * w[k];
    if (A.exp_avg_sq) A.exp_avg_sq += lo * w[k];
  }
  hipLaunchKernelGGL(gaussian_adam_kernel, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hi - lo, fa.G,
                     fa.c, fa.iso_coef, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "gaussian_adam_shard launch failed");
}

int sgr_masked_adam(int32_t rows, int32_t row_width, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                    int32_t* step, const int32_t* active, float lr, float beta1, float beta2, float eps, void* stream) {
  if (rows < 0 || row_width <= 0 || (rows > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !step || !active)))
    return set_error(SGR_ERR_INVALID, "masked_adam: bad argument");
  if (rows == 0) return SGR_OK;
  hipLaunchKernelGGL(masked_adam_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, rows, row_width, param, grad,
                     exp_avg, exp_avg_sq, step, active, lr, beta1, beta2, eps);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "masked_adam launch failed");
}

size_t sknn_scratch_bytes(int32_t n) {
  if (n <= 0) return 256;
  if (n >= kKnnGridFrom) return knn_grid_scratch_bytes(n);
  return (size_t)knn_splits(n) * (size_t)n * sizeof(Top3) + 256;
}

int sknn_dist2(const float* xyz, int32_t n, float* mean_dist2, void* scratch, size_t scratch_bytes, void* stream) {
  if (n < 0 || (n > 0 && (!xyz || !mean_dist2))) return set_error(SGR_ERR_INVALID, "sknn: null argument");
  if (n == 0) return SGR_OK;
  if (!scratch || scratch_bytes < sknn_scratch_bytes(n)) return set_error(SGR_ERR_WORKSPACE, "sknn scratch too small");
  int qb = (n + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (n >= kKnnGridFrom) {
    const size_t mc = knn_max_cells(n);
    char* base = (char*)scratch;
    uint32_t* mm = (uint32_t*)base;                      base += 256;
    KnnGrid* G = (KnnGrid*)base;                         base += 256;
    uint32_t* count = (uint32_t*)base;                   base += align_up(mc * 4);
    uint32_t* start = (uint32_t*)base;                   base += align_up(mc * 4);
    uint32_t* cursor = (uint32_t*)base;                  base += align_up(mc * 4);
    uint32_t* cell_of = (uint32_t*)base;                 base += align_up((size_t)n * 4);
    float* sorted_xyz = (float*)base;                    base += align_up((size_t)n * 12);
    uint32_t* sorted_idx = (uint32_t*)base;
    if (hipMemsetAsync(mm, 0xff, 12, st) != hipSuccess || hipMemsetAsync(mm + 3, 0, 12, st) != hipSuccess ||
        hipMemsetAsync(count, 0, mc * 4, st) != hipSuccess)
      return set_error(SGR_ERR_HIP, "sknn memset failed");
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(min(qb, 256)), dim3(256), 0, st, n, xyz, mm);
    hipLaunchKernelGGL(knn_grid_setup_kernel, dim3(1), dim3(64), 0, st, n, (int)mc, mm, G);
    hipLaunchKernelGGL(knn_count_kernel, dim3(qb), dim3(256), 0, st, n, xyz, G, count, cell_of);
    hipLaunchKernelGGL(knn_scan_kernel, dim3(1), dim3(1024), 0, st, G, count, start, cursor);
    hipLaunchKernelGGL(knn_fill_kernel, dim3(qb), dim3(256), 0, st, n, xyz, cell_of, cursor, sorted_xyz, sorted_idx);
    hipLaunchKernelGGL(knn_query_kernel, dim3(qb), dim3(256), 0, st, n, G, start, count, sorted_xyz, sorted_idx, mean_dist2);
    return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "sknn launch failed");
  }
  int splits = knn_splits(n);
  int per = ((n + splits - 1) / splits + 255) / 256 * 256;
  hipLaunchKernelGGL(knn_partial_kernel, dim3(qb, splits), dim3(256), 0, st, n, xyz, per, (Top3*)scratch);
  hipLaunchKernelGGL(knn_merge_kernel, dim3(qb), dim3(256), 0, st, n, splits, (const Top3*)scratch, mean_dist2);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "sknn launch failed");
}

int se3_exp(const float* tau, int64_t n, float* pose_out, void* stream) { return se3_launch<OP_EXP>(n, tau, nullptr, pose_out, stream); }
int se3_log(const float* pose, int64_t n, float* tau_out, void* stream) { return se3_launch<OP_LOG>(n, pose, nullptr, tau_out, stream); }
int se3_inv(const float* pose, int64_t n, float* pose_out, void* stream) { return se3_launch<OP_INV>(n, pose, nullptr, pose_out, stream); }
int se3_mul(const float* a, const float* b, int64_t n, float* pose_out, void* stream) { return se3_launch<OP_MUL>(n, a, b, pose_out, stream); }
int se3_act(const float* pose, const float* pts, int64_t n, float* pts_out, void* stream) { return se3_launch<OP_ACT>(n, pose, pts, pts_out, stream); }
int se3_adjT(const float* pose, const float* a, int64_t n, float* out, void* stream) { return se3_launch<OP_ADJT>(n, pose, a, out, stream); }
int se3_matrix(const float* pose, int64_t n, float* mat_out, void* stream) { return se3_launch<OP_MATRIX>(n, pose, nullptr, mat_out, stream); }

}  // extern "C"
