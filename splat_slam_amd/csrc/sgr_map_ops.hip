// Map surgery on the device for gfx950: moving the Gaussians anchored to a keyframe the tracker re-estimated, and
// row compaction / gathering of the per-Gaussian tensors for densify / prune.
//
// Replaces the torch-op chains of Mapper.update_mapping_points (/root/reference/src/mapper.py:154-255) and of
// GaussianModel.prune_points / _prune_optimizer / cat_tensors_to_optimizer
// (/root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:519-597): ~20 elementwise / index kernels per
// call become one pass over the Gaussians, respectively one scan plus one gather pass over ALL per-Gaussian tensors.
#include <cstring>

#include "sgr_common.h"

namespace sgr {
int set_error(int code, const char* fmt, ...);

struct DeformArgs {
  float w2c_old[16], c2w_old[16], transform[16];   // row-major 4x4
  float K[9];
  float tq[4];                                      // rotation of `transform` as (w, x, y, z)
  int frame_idx, rigid, H, W;
};

__device__ __forceinline__ void mat4_apply(const float* M, const float p[3], float out[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) out[r] = M[4 * r] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
}

// thread = Gaussian.  EVERY rotation leaves normalised (the reference writes get_rotation -- the activated tensor -- back
// as the parameter, mapper.py:246-250); Gaussians anchored to `frame_idx` are additionally rescaled along the old camera's
// ray by the depth change at their pixel (:218-226), moved by the pose change (:235-237), rotated (:246-250) and have
// log(rescale) added to their log-scale (:253-255).
__global__ void __launch_bounds__(256) deform_kernel(int64_t n, const int32_t* __restrict__ kf_ids, DeformArgs a,
                                                     const float* __restrict__ depth_new, const float* __restrict__ depth_old,
                                                     float* __restrict__ xyz, float* __restrict__ rot, float* __restrict__ scaling) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 q = *(const float4*)(rot + 4 * i);
  {
    const float nn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    q = make_float4(q.x / nn, q.y / nn, q.z / nn, q.w / nn);
  }
  if (kf_ids[i] == a.frame_idx) {
    const F3 p3 = ld3(xyz + 3 * i);
    float p[3] = {p3.x, p3.y, p3.z};
    if (!a.rigid) {
      float pc[3];
      mat4_apply(a.w2c_old, p, pc);
      const float ph0 = a.K[0] * pc[0] + a.K[1] * pc[1] + a.K[2] * pc[2];
      const float ph1 = a.K[3] * pc[0] + a.K[4] * pc[1] + a.K[5] * pc[2];
      const float ph2 = a.K[6] * pc[0] + a.K[7] * pc[1] + a.K[8] * pc[2];
      // .long() truncates toward zero; clamp into the image like torch.clamp (NaN / inf land on a border pixel)
      const float uf = ph0 / ph2, vf = ph1 / ph2;
      long long u = (uf == uf) ? (long long)fminf(fmaxf(uf, -1e18f), 1e18f) : 0, v = (vf == vf) ? (long long)fminf(fmaxf(vf, -1e18f), 1e18f) : 0;
      u = u < 0 ? 0 : (u > a.W - 1 ? a.W - 1 : u);
      v = v < 0 ? 0 : (v > a.H - 1 ? a.H - 1 : v);
      const float d_new = depth_new[v * a.W + u], d_old = depth_old[v * a.W + u];
      float rescale = 1.f + 1.f / pc[2] * (d_new - d_old);
      if (d_new == 0.f || d_old == 0.f) rescale = 1.f;
      if (rescale <= 0.f) rescale = 1.f;
      pc[0] = rescale * pc[0]; pc[1] = rescale * pc[1]; pc[2] = rescale * pc[2];
      mat4_apply(a.c2w_old, pc, p);
      const F3 s3 = ld3(scaling + 3 * i);
      const float lr = logf(rescale);
      st3(scaling + 3 * i, F3{s3.x + lr, s3.y + lr, s3.z + lr});
    }
    float pn[3];
    mat4_apply(a.transform, p, pn);
    st3(xyz + 3 * i, F3{pn[0], pn[1], pn[2]});
    // quaternion_multiply(tq, q), (w, x, y, z) layout (general_utils.py:160-175)
    const float w1 = a.tq[0], x1 = a.tq[1], y1 = a.tq[2], z1 = a.tq[3], w2 = q.x, x2 = q.y, y2 = q.z, z2 = q.w;
    q = make_float4(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                    w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2);
  }
  *(float4*)(rot + 4 * i) = q;
}

// ---- row compaction: out[k] = in[src[k]] for every tensor of a table (rows of 4-byte elements)
struct RowTab {
  const uint32_t* in[32];
  uint32_t* out[32];
  int32_t row_words[32];
  int32_t count;
};

// keep mask -> ascending list of kept row indices (deterministic: block scans + an ordered second pass)
__global__ void __launch_bounds__(256) keep_count_kernel(int64_t n, const uint8_t* __restrict__ keep, uint32_t* __restrict__ block_tot) {
  __shared__ uint32_t red[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t tot;
  (void)block256_exclusive_scan((i < n && keep[i]) ? 1u : 0u, red, tot);
  if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) keep_scan_kernel(int nblocks, uint32_t* __restrict__ block_tot, int64_t* __restrict__ total) {
  __shared__ uint32_t part[1024];
  // single block, sequential chunks: nblocks <= a few thousand
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    const int b = base + threadIdx.x;
    const uint32_t v = b < nblocks ? block_tot[b] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = carry;
      for (int k = 0; k < 1024; ++k) { uint32_t t = part[k]; part[k] = run; run += t; }
      carry = run;
    }
    __syncthreads();
    if (b < nblocks) block_tot[b] = part[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = (int64_t)carry;
}
__global__ void __launch_bounds__(256) keep_list_kernel(int64_t n, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ block_base,
                                                        int32_t* __restrict__ src) {
  __shared__ uint32_t red[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool k = i < n && keep[i];
  uint32_t tot;
  const uint32_t ex = block256_exclusive_scan(k ? 1u : 0u, red, tot);
  if (k) src[block_base[blockIdx.x] + ex] = (int32_t)i;
}
// grid = (row blocks, tensors): thread = one output row of one tensor
__global__ void __launch_bounds__(256) gather_rows_kernel(int64_t m, const int32_t* __restrict__ src, RowTab tab) {
  const int t = blockIdx.y;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= m) return;
  const int w = tab.row_words[t];
  const uint32_t* in = tab.in[t] + (int64_t)src[k] * w;
  uint32_t* out = tab.out[t] + k * w;
  for (int j = 0; j < w; ++j) out[j] = in[j];
}

}  // namespace sgr

using namespace sgr;

extern "C" {

int sgr_deform_points(int64_t n, const int32_t* unique_kfIDs, const SgrDeformFrame* f, float* xyz, float* rotation,
                      float* scaling, void* stream) {
  if (n < 0 || !f || (n > 0 && (!unique_kfIDs || !xyz || !rotation || !scaling)))
    return set_error(SGR_ERR_INVALID, "deform_points: null argument");
  if (!f->rigid && (!f->depth_new || !f->depth_old || f->height <= 0 || f->width <= 0))
    return set_error(SGR_ERR_INVALID, "deform_points: depth maps are required unless rigid");
  if (n == 0) return SGR_OK;
  DeformArgs a;
  memcpy(a.w2c_old, f->w2c_old, sizeof(a.w2c_old));
  memcpy(a.c2w_old, f->c2w_old, sizeof(a.c2w_old));
  memcpy(a.transform, f->transform, sizeof(a.transform));
  memcpy(a.K, f->intrinsics, sizeof(a.K));
  memcpy(a.tq, f->quat_wxyz, sizeof(a.tq));
  a.frame_idx = f->frame_idx; a.rigid = f->rigid; a.H = f->height; a.W = f->width;
  hipLaunchKernelGGL(deform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, unique_kfIDs, a,
                     f->depth_new, f->depth_old, xyz, rotation, scaling);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "deform launch failed");
}

size_t sgr_compact_scratch_bytes(int64_t n) { return (size_t)((n + 255) / 256 + 1) * 4 + 256; }

int sgr_keep_list(int64_t n, const uint8_t* keep, int32_t* src_rows, int64_t* count_device, void* scratch, size_t scratch_bytes,
                  void* stream) {
  if (n < 0 || (n > 0 && (!keep || !src_rows)) || !count_device) return set_error(SGR_ERR_INVALID, "keep_list: null argument");
  if (!scratch || scratch_bytes < sgr_compact_scratch_bytes(n)) return set_error(SGR_ERR_WORKSPACE, "keep_list: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n + 255) / 256);
  uint32_t* bt = (uint32_t*)scratch;
  if (nb > 0) hipLaunchKernelGGL(keep_count_kernel, dim3(nb), dim3(256), 0, st, n, keep, bt);
  hipLaunchKernelGGL(keep_scan_kernel, dim3(1), dim3(1024), 0, st, nb, bt, count_device);
  if (nb > 0) hipLaunchKernelGGL(keep_list_kernel, dim3(nb), dim3(256), 0, st, n, keep, bt, src_rows);
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "keep_list launch failed");
}

int sgr_gather_rows(int64_t m, const int32_t* src_rows, int32_t num_tensors, const SgrRowTensor* tensors, void* stream) {
  if (m < 0 || num_tensors < 0 || (num_tensors > 0 && !tensors) || (m > 0 && !src_rows))
    return set_error(SGR_ERR_INVALID, "gather_rows: null argument");
  if (m == 0 || num_tensors == 0) return SGR_OK;
  for (int base = 0; base < num_tensors; base += 32) {
    RowTab tab;
    tab.count = num_tensors - base < 32 ? num_tensors - base : 32;
    for (int t = 0; t < tab.count; ++t) {
      const SgrRowTensor& r = tensors[base + t];
      if (!r.in || !r.out || r.row_bytes <= 0 || (r.row_bytes & 3))
        return set_error(SGR_ERR_INVALID, "gather_rows: tensor %d needs pointers and a row size that is a multiple of 4", base + t);
      tab.in[t] = (const uint32_t*)r.in; tab.out[t] = (uint32_t*)r.out; tab.row_words[t] = r.row_bytes / 4;
    }
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((m + 255) / 256), tab.count), dim3(256), 0, (hipStream_t)stream, m,
                       src_rows, tab);
  }
  return hipGetLastError() == hipSuccess ? SGR_OK : set_error(SGR_ERR_HIP, "gather_rows launch failed");
}

}  // extern "C"
