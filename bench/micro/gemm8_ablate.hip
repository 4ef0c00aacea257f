// Standalone ablation harness for the ping-pong GEMM: builds the kernel with parts of a phase compiled out
// (PDT_ABLATE: 0 = full kernel, 1 = no LDS-DMA in the K loop, 2 = no LDS fragment reads, 3 = no MFMA) and times
// it on one BERT-large shape with random data.  Outputs of ablated builds are wrong by construction.
// -DPDT_LAYOUT=0 (default) NT: A[M,K] B[N,K];  1 NN: A[M,K] B[K,N];  2 TN: A[K,M] B[K,N] with split-K fp32 slabs
// (as many K slices as fill 256 workgroups).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
// -DPDT_STAMPS=1: wave 0 of every workgroup records the constant-rate clock (10 ns ticks) at the boundaries of every
// tile it computes (0 tile start, 1 K loop done, 2 wave groups re-aligned, 3 epilogue issued, 4 memory counter empty);
// the harness prints the mean length of each section.  -DPDT_EPI=<PDT_EPI_* code>: run with that fused epilogue
// (side inputs are random 16-bit tensors; dropout off).
#ifndef PDT_STAMPS
#define PDT_STAMPS 0
#endif
#if PDT_STAMPS
#include <hip/hip_runtime.h>
__device__ unsigned long long* pdt_stamp_buf;
#define PDT_STAMP_MAXT 40
// (stamps 0 and 1 also record the shader-cycle counter in slots 5 and 6: cycles / time of the K loop = the clock it ran at)
#define PDT_STAMP(id) do { if (threadIdx.x == 0 && stamp_seq < PDT_STAMP_MAXT && pdt_stamp_buf != nullptr) { \
  unsigned long long* q_ = pdt_stamp_buf + ((size_t)blockIdx.x * PDT_STAMP_MAXT + stamp_seq) * 8; \
  q_[(id)] = wall_clock64(); if ((id) < 2) q_[5 + (id)] = __builtin_readcyclecounter(); } } while (0)
#endif
#include "../../csrc/kernels/gemm8.hpp"
// -DPDT_KERNEL4=1: time the four-wave kernel (csrc/kernels/gemm4.hpp) instead of the ping-pong kernel
#ifndef PDT_KERNEL4
#define PDT_KERNEL4 0
#endif
#if PDT_KERNEL4
#include "gemm4.hpp"
#endif
#ifndef PDT_EPI
#define PDT_EPI PDT_EPI_PLAIN
#endif

#ifndef PDT_LAYOUT
#define PDT_LAYOUT 0
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
  std::vector<_Float16> ha((size_t)M * K), hb((size_t)N * K);
  srand(1);
  for (auto& v : ha) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  for (auto& v : hb) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  _Float16 *A, *B, *C;
  constexpr bool AKS = PDT_LAYOUT == 2, BKS = PDT_LAYOUT >= 1;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int splits = 1;
  if (PDT_LAYOUT == 2) { splits = 256 / tiles; if (splits < 1) splits = 1; }
  CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&B, hb.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4 * splits));
  CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  PdtGemm g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = AKS ? M : K; g.ldb = BKS ? N : K; g.ldc = N;
  g.epi = PDT_LAYOUT == 2 ? PDT_EPI_SLAB_F32 : PDT_EPI;
  if (g.epi != PDT_EPI_PLAIN && g.epi != PDT_EPI_SLAB_F32) {  // side inputs of the fused epilogues
    std::vector<_Float16> hs((size_t)M * N);
    for (auto& v : hs) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    _Float16 *S, *AUX, *BIAS;
    CK(hipMalloc(&S, hs.size() * 2)); CK(hipMalloc(&AUX, hs.size() * 2)); CK(hipMalloc(&BIAS, (size_t)N * 2));
    CK(hipMemcpy(S, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(AUX, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(BIAS, hs.data(), (size_t)N * 2, hipMemcpyHostToDevice));
    g.bias = BIAS; g.res = S; g.ldr = N; g.aux = AUX; g.ldaux = N; g.drop_scale = 1.0f; g.drop_thr = 0;
    if (getenv("PDT_DROP")) {  // dropout p = 0.1 with the row index as the dropout row
      uint64_t hst[2] = {1234ull, 7ull}, *dst;
      CK(hipMalloc(&dst, 16)); CK(hipMemcpy(dst, hst, 16, hipMemcpyHostToDevice));
      g.rng.state = dst; g.rng.micro = 1; g.rng.stream = 5; g.drop_thr = 6554; g.drop_scale = 1.0f / 0.9f;
    }
  }
  // argv[5]: 0 = the general epilogue everywhere, otherwise the straight-line epilogue for full tiles (needs 31-bit offsets)
  g.fast_epi = (argc > 5 ? atoi(argv[5]) : 1) != 0 && (long long)M * N * 2 < 0x7FFFFFFFll;
  g.tiles_m = (M + 255) / 256; g.tiles_n = (N + 255) / 256; g.splits = splits;
  g.k_per_split = ((K / 64 + splits - 1) / splits) * 64;
#if PDT_KERNEL4
  auto kern = pdt::gemm4_kernel<_Float16, AKS, BKS>;
  using KK = pdt::Gemm4<_Float16, AKS, BKS>;
#else
  auto kern = pdt::gemm8_kernel<_Float16, AKS, BKS>;
  using KK = pdt::Gemm8<_Float16, AKS, BKS>;
#endif
  const int threads = KK::NT;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, KK::LDS_BYTES));
  const int total = g.tiles_m * g.tiles_n * splits;
  // argv[4]: workgroups of a persistent launch (0 = one workgroup per tile); needs full tiles and an even K-tile count
  const int want = argc > 4 ? atoi(argv[4]) : 0;
  const int grid = (want > 0 && 2 * want <= total && want % 8 == 0 && PDT_LAYOUT != 2 && M % 256 == 0 && N % 256 == 0 && K % 128 == 0) ? want : total;
  if (grid < total) {  // ticket counters of a persistent launch: 2 fixed tiles per workgroup already handed out
    std::vector<uint32_t> init(8, 2u * (grid / 8));
    CK(hipMalloc(&g.sync, 32)); CK(hipMemcpy(g.sync, init.data(), 32, hipMemcpyHostToDevice));
  }
#if PDT_STAMPS
  // the stamp buffer exists before the first launch: every launch of this build writes stamps
  const size_t stamp_n = (size_t)grid * PDT_STAMP_MAXT * 8;
  unsigned long long* sb; CK(hipMalloc(&sb, stamp_n * 8)); CK(hipMemset(sb, 0, stamp_n * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(pdt_stamp_buf), &sb, sizeof(sb)));
  CK(hipDeviceSynchronize());
#endif
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), KK::LDS_BYTES, 0, g);
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(ev0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), KK::LDS_BYTES, 0, g);
    CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
    if (ms / 10 < best) best = ms / 10;
  }
#if PDT_STAMPS
  {
    const size_t n = stamp_n;
    CK(hipMemset(sb, 0, n * 8));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), KK::LDS_BYTES, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hsb(n);
    CK(hipMemcpy(hsb.data(), sb, n * 8, hipMemcpyDeviceToHost));
    double sum[5] = {0, 0, 0, 0, 0}, cyc = 0; long cnt = 0; unsigned long long t_first = ~0ull, t_last = 0;
    for (int b = 0; b < grid; ++b)
      for (int t = 0; t < PDT_STAMP_MAXT; ++t) {
        const unsigned long long* q = &hsb[((size_t)b * PDT_STAMP_MAXT + t) * 8];
        if (q[4] == 0) break;
        if (q[0] < t_first) t_first = q[0];
        if (q[4] > t_last) t_last = q[4];
        if (t == 0) continue;  // (the first tile of a workgroup includes the prologue)
        const unsigned long long* prev = q - 8;
        sum[0] += (double)(q[0] - prev[4]);  // hand-over: mailbox + barrier
        sum[1] += (double)(q[1] - q[0]);     // K loop
        sum[2] += (double)(q[2] - q[1]);     // re-align
        sum[3] += (double)(q[3] - q[2]);     // epilogue until its last instruction is issued
        sum[4] += (double)(q[4] - q[3]);     // memory counter drains
        cyc += (double)(q[6] - q[5]);        // shader cycles of the K loop
        ++cnt;
      }
    if (cnt) printf("stamps (us, mean over %ld tiles): handover %.2f  kloop %.2f  realign %.2f  epilogue %.2f  drain %.2f  | K loop %.0f cycles at %.2f GHz | kernel span %.1f us\n", cnt,
                    sum[0] / cnt * 0.01, sum[1] / cnt * 0.01, sum[2] / cnt * 0.01, sum[3] / cnt * 0.01, sum[4] / cnt * 0.01, cyc / cnt, cyc / (sum[1] * 10.0), (double)(t_last - t_first) * 0.01);
  }
#endif
  if (PDT_ABLATE == 0 && PDT_LAYOUT != 2 && PDT_EPI == PDT_EPI_PLAIN) {  // spot check against the host: 64 random elements
    // a fresh launch into a zeroed C (after the 55 launches above: the ticket counters must have been re-armed every time)
    CK(hipMemset(C, 0, (size_t)M * N * 2));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), KK::LDS_BYTES, 0, g);
    CK(hipDeviceSynchronize());
    std::vector<_Float16> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    // one sample in every 256x256 tile + random ones
    const int tm_n = M / 256 > 0 ? M / 256 : 1, tn_n = N / 256 > 0 ? N / 256 : 1;
    const int nsamp = tm_n * tn_n + 64;
    for (int s = 0; s < nsamp; ++s) {
      int m, n;
      if (s < tm_n * tn_n) { m = (s / tn_n) * 256 + rand() % 256; n = (s % tn_n) * 256 + rand() % 256; if (m >= M) m = M - 1; if (n >= N) n = N - 1; }
      else { m = rand() % M; n = rand() % N; }
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)(float)ha[(size_t)m * K + k] * (double)(float)(BKS ? hb[(size_t)k * N + n] : hb[(size_t)n * K + k]);
      const double err = fabs((double)(float)hc[(size_t)m * N + n] - ref) / (fabs(ref) + 1.0);
      if (err > worst) worst = err;
    }
    printf("grid=%d  worst relative error of %d sampled outputs (every tile): %.2e %s\n", grid, nsamp, worst, worst < 2e-2 ? "ok" : "MISMATCH");
  }
  printf("kernel=%s layout=%d phases=%d ablate=%d epi=%d fast=%d M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s(nominal)\n", PDT_KERNEL4 ? "4wave" : "pingpong", PDT_LAYOUT, PDT_PHASES, PDT_ABLATE, (int)g.epi, g.fast_epi, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
  return 0;
}
