// Does an fp32 MFMA stream cost VALU issue time on gfx950?  Every wave runs ITER iterations of NV dependent-free packed
// FMAs (the tile backward's kind of work) plus NM v_mfma_f32_4x4x1_16b_f32 (outer-product accumulates) per iteration; 5 waves
// per SIMD, all CUs.  Prints time per iteration for NM = 0, 3, 6, 12: if the matrix pipe is fed on its own issue port the
// times are equal.    hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_coissue scripts/micro/mfma_coissue.hip && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NM>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5))) k(float* out, int iters, float seed) {
  v2f a[12];
  for (int i = 0; i < 12; ++i) a[i] = (v2f){seed + i + threadIdx.x, seed - i};
  v2f m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
  v4f acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float wa = seed + threadIdx.x, wb = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int i = 0; i < 12; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);     // 60 packed FMAs, 12 independent chains
#pragma unroll
    for (int q = 0; q < NM; ++q) acc[q % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa + q, wb, acc[q % 3], 0, 0, 0);
    wa += 1e-3f;
  }
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += a[i].x + a[i].y;
  for (int q = 0; q < 3; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NM>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4 * 5 * 4;     // 4 rounds of 5 waves per SIMD
  hipLaunchKernelGGL(k<NM>, dim3(grid), dim3(64), 0, 0, out, iters, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NM>, dim3(grid), dim3(64), 0, 0, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 5 * 4 * 64 * sizeof(float));
  const int iters = 2000;
  float t0 = run<0>(out, iters), t3 = run<3>(out, iters), t6 = run<6>(out, iters), t12 = run<12>(out, iters);
  // per SIMD: 20 waves x iters iterations
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (20.0 * iters); };
  printf("{\"what\": \"cycles per (wave, iteration) at 2.4 GHz: 60 packed FMAs + NM fp32 MFMA 4x4x1\", \"NM0\": %.1f, \"NM3\": %.1f, \"NM6\": %.1f, \"NM12\": %.1f, \"ms\": [%.3f, %.3f, %.3f, %.3f]}\n",
         cyc(t0), cyc(t3), cyc(t6), cyc(t12), t0, t3, t6, t12);
  return 0;
}
