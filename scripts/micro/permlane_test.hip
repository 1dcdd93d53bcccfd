// Micro-test (GPU box): lane semantics of v_permlane32_swap / v_permlane16_swap (gfx950) and of the DPP row_ror adds that
// the backward's cross-group reduction uses.   hipcc --offload-arch=gfx950 -O3 -o /tmp/pl scripts/micro/permlane_test.hip && /tmp/pl
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  float a = (float)lane, b = 100.f + (float)lane;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[lane] = a; out[64 + lane] = b;
  float c = (float)lane, d = 100.f + (float)lane;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  out[128 + lane] = c; out[192 + lane] = d;
  float e = (float)lane;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(e));
  out[256 + lane] = e;
  float f = (float)lane;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(f));
  out[320 + lane] = f;
}
int main() {
  float* d; hipMalloc(&d, 384 * 4);
  k<<<1, 64>>>(d);
  float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"permlane32_swap a", "permlane32_swap b", "permlane16_swap a", "permlane16_swap b", "add row_ror:8", "add row_ror:4"};
  for (int r = 0; r < 6; ++r) { printf("%-18s:", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
  return 0;
}
