// Phase timing of the fused S<=128 attention backward kernel: builds attention.hip with -DPDT_ATT_STAMP so that
// wave 0 of every workgroup leaves 100-MHz time stamps (start, first fill done, after barrier, after dQ sweep,
// after the re-fill, after dK/dV sweep, end) in the delta buffer; prints the mean phase durations and the workgroup lifetime.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -DPDT_ATT_STAMP attn_stamp.hip -o attn_stamp
// Ablations: add -DPDT_ATT_NOCOMPUTE=1 (both sweeps compiled out: the kernel's memory traffic alone) and / or
// -DPDT_ATT_NOSTORE=1 (dQ / dK / dV not written).  Usage: attn_stamp [batch = 256] [dropout p = 0.1]
#include "../../csrc/kernels/attention.hip"
#include <cstdio>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, S = 128, A = 16, H = A * 64;
  const float p = argc > 2 ? atof(argv[2]) : 0.1f;
  const size_t T = (size_t)B * S;
  std::vector<_Float16> h(T * 3 * H);
  uint32_t x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 16) % 2001 - 1000) * 0.0007f); }
  _Float16 *qkv, *ctx, *dctx, *dqkv; float *lse, *delta, *dbias; int32_t *mask, *kvl; int64_t* rng;
  HC(hipMalloc(&qkv, T * 3 * H * 2)); HC(hipMalloc(&dqkv, T * 3 * H * 2)); HC(hipMalloc(&ctx, T * H * 2)); HC(hipMalloc(&dctx, T * H * 2));
  HC(hipMalloc(&lse, (size_t)B * A * S * 4)); HC(hipMalloc(&delta, (size_t)B * A * S * 4)); HC(hipMalloc(&dbias, (size_t)B * 3 * H * 4));
  HC(hipMalloc(&mask, T * 4)); HC(hipMalloc(&kvl, B * 4)); HC(hipMalloc(&rng, 16));
  HC(hipMemcpy(qkv, h.data(), T * 3 * H * 2, hipMemcpyHostToDevice));
  HC(hipMemcpy(dctx, h.data(), T * H * 2, hipMemcpyHostToDevice));
  std::vector<int32_t> ones(T, 1), lens(B, S);
  HC(hipMemcpy(mask, ones.data(), T * 4, hipMemcpyHostToDevice)); HC(hipMemcpy(kvl, lens.data(), B * 4, hipMemcpyHostToDevice));
  int64_t st[2] = {1234, 0}; HC(hipMemcpy(rng, st, 16, hipMemcpyHostToDevice));
  PdtRng r{p > 0 ? (const uint64_t*)rng : nullptr, 0, 0};
  const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f); const float sc = p > 0 ? 1.0f / (1.0f - p) : 1.0f;
  if (pdt_attn_fwd(PDT_F16, qkv, mask, kvl, nullptr, ctx, lse, B, S, A, sc, thr, r, nullptr)) return 2;
  hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) {
    HC(hipEventRecord(e0));
    if (pdt_attn_bwd(PDT_F16, qkv, mask, kvl, nullptr, ctx, dctx, lse, delta, dqkv, dbias, B, S, A, sc, thr, r, nullptr)) return 3;
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    printf("kernel %.1f us\n", ms * 1e3);
  }
  std::vector<float> d((size_t)B * A * S);
  HC(hipMemcpy(d.data(), delta, d.size() * 4, hipMemcpyDeviceToHost));
  double ph[6] = {0, 0, 0, 0, 0, 0}; double tmin = 1e30, tmax = 0;
  const int nb = B * A;
  for (int i = 0; i < nb; ++i) {
    const float* t = &d[(size_t)i * S];
    for (int k = 0; k < 6; ++k) { double dt = t[k + 1] - t[k]; if (dt < 0) dt += 16777216.0; ph[k] += dt; }
    if (t[0] < tmin) tmin = t[0];
    if (t[6] > tmax) tmax = t[6];
  }
  const char* names[6] = {"K,V fill+delta", "barrier wait", "dQ sweep", "own K,V rows", "dK/dV sweep", "reduce+tail"};
  double tot = 0;
  for (int k = 0; k < 6; ++k) { printf("%-14s %7.2f us\n", names[k], ph[k] / nb * 0.01); tot += ph[k] / nb * 0.01; }
  printf("workgroup life %7.2f us ; first start -> last end %.1f us ; %d workgroups\n", tot, (tmax - tmin) * 0.01, nb);
  return 0;
}
