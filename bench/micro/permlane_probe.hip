// What v_permlane16_swap / v_permlane32_swap do, lane by lane: a = lane, b = 100 + lane before the swap.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* names[4] = {"swap16 first ", "swap16 second", "swap32 first ", "swap32 second"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[v * 64 + l]); printf("\n"); }
  return 0;
}
