// Micro-benchmark (GPU box): returning 64-bit counting atomics, device scope vs. workgroup scope on a per-XCD copy.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics_bench scripts/micro/atomics_bench.hip && /tmp/atomics_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* c, int words, int per_thread, unsigned long long* sink, int coherent) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  unsigned long long acc = 0;
  unsigned long long* base = c;
  if (MODE == 1) base = c + (size_t)xcc_id() * words;
  for (int i = 0; i < per_thread; i += 4) {
    unsigned long long o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t w = coherent ? (hash(blockIdx.x) + (hash(tid * 131u + i + j) & 63u)) % words : hash(tid * 131u + i + j) % words;
      if (MODE == 2) o[j] = __hip_atomic_fetch_add((unsigned int*)(base + w) + (w & 1), 0x00010001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 3) { __hip_atomic_fetch_add(base + w, 0x0001000000010001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); o[j] = 0; }
      else if (MODE == 4) { __hip_atomic_fetch_add((unsigned int*)(base + w) + (w & 1), 0x00010001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); o[j] = 0; }
      else if (MODE == 0) o[j] = __hip_atomic_fetch_add(base + w, 0x0001000000010001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else o[j] = __hip_atomic_fetch_add(base + w, 0x0001000000010001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += o[j];
  }
  if (acc == 0x1234567ull) sink[0] = acc;
}
int main() {
  const int words = 14400, blocks = 1172, per_thread = 8;   // 2.4 M atomics
  unsigned long long *c, *sink;
  hipMalloc(&c, sizeof(unsigned long long) * words * 16);
  hipMalloc(&sink, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int coherent = 0; coherent < 2; ++coherent)
    for (int mode = 0; mode < 5; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipMemset(c, 0, sizeof(unsigned long long) * words * 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        switch (mode) { case 0: k<0><<<blocks, 256>>>(c, words, per_thread, sink, coherent); break; case 1: k<1><<<blocks, 256>>>(c, words, per_thread, sink, coherent); break; case 2: k<2><<<blocks, 256>>>(c, words, per_thread, sink, coherent); break; case 3: k<3><<<blocks, 256>>>(c, words, per_thread, sink, coherent); break; default: k<4><<<blocks, 256>>>(c, words, per_thread, sink, coherent); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      // check the per-XCD copies add up
      std::vector<unsigned long long> h(words * 16);
      hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long tot = 0; int used = 0;
      for (int x = 0; x < 16; ++x) { unsigned long long s = 0; for (int w = 0; w < words; ++w) s += h[(size_t)x * words + w] & 0xffff; tot += s; used += s != 0; }
      printf("coherent=%d mode=%s: %.1f us for %d atomics (%.1f G/s), total=%llu copies_used=%d\n", coherent, (mode == 0 ? "agent u64 ret" : mode == 1 ? "per-XCD u64 ret" : mode == 2 ? "u32 ret" : mode == 3 ? "u64 noret" : "u32 noret"), best * 1e3f,
             blocks * 256 * per_thread, blocks * 256 * per_thread / (best * 1e6f), tot, used);
    }
  return 0;
}
