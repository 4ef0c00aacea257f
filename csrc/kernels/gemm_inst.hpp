// Per-layout instantiation helper: each gemm_<layout>.hip defines PDT_AKS / PDT_BKS / PDT_LAYOUT_FN and
// includes this file, so the three operand layouts compile in parallel translation units.
#pragma once
#include "gemm8.hpp"

namespace pdt {

template <typename T, int BM, int BN, int WM, int WN>
static int launch_tile(PdtGemm& g, hipStream_t stream) {
  using K = GemmKernel<T, BM, BN, WM, WN, PDT_AKS, PDT_BKS>;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, PDT_AKS, PDT_BKS>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  if (g.epi != PDT_EPI_SLAB_F32) {
    g.splits = 1;
    g.k_per_split = ((g.K + 63) / 64) * 64;
  }
  const int grid = g.tiles_m * g.tiles_n * g.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(K::NT), K::LDS_BYTES, stream, g);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_pingpong(PdtGemm& g, hipStream_t stream) {
  using K = Gemm8<T, PDT_AKS, PDT_BKS>;
  auto kern = gemm8_kernel<T, PDT_AKS, PDT_BKS>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  g.tiles_m = (g.M + 255) / 256;
  g.tiles_n = (g.N + 255) / 256;
  if (g.epi != PDT_EPI_SLAB_F32) {
    g.splits = 1;
    g.k_per_split = ((g.K + 63) / 64) * 64;
  }
  const int grid = g.tiles_m * g.tiles_n * g.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(K::NT), K::LDS_BYTES, stream, g);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_dtype(int tile, PdtGemm& g, hipStream_t stream) {
  if (tile == 2) return launch_pingpong<T>(g, stream);
  if (tile == 1) return launch_tile<T, 256, 256, 2, 4>(g, stream);
  return launch_tile<T, 128, 128, 2, 2>(g, stream);
}

}  // namespace pdt

extern "C" int PDT_LAYOUT_FN(int dtype, int tile, PdtGemm* g, void* stream) {
  if (dtype == PDT_F16) return pdt::launch_dtype<_Float16>(tile, *g, (hipStream_t)stream);
  if (dtype == PDT_BF16) return pdt::launch_dtype<__bf16>(tile, *g, (hipStream_t)stream);
  return -1;
}
