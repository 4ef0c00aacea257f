// Per-layout instantiation helper: each gemm_<layout>.hip defines PDT_AKS / PDT_BKS / PDT_LAYOUT_FN and
// includes this file, so the three operand layouts compile in parallel translation units.
#pragma once
#include <mutex>
#include <unordered_map>
#include "gemm8.hpp"

extern "C" int pdt_gemm_persist_mode(void);  // gemm_dispatch.hip
extern "C" int pdt_gemm_fast_epi_mode(void);

namespace pdt {

// Second kernel of a split-K forward / dgrad GEMM: C = epilogue(sum_s ws[s]) on 8 consecutive columns per thread,
// with exactly the epilogue code of the GEMM kernels.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(PdtGemm g) {
  using K = GemmKernel<T, 128, 128, 2, 2, PDT_AKS, PDT_BKS>;
  const int nch = g.N >> 3;
  const int64_t total = (int64_t)g.M * nch;
  const DropKey dk = drop_key(g.rng);
  const bool dgelu = EPI == EPI_MUL_DGELU;
  const T* ext = dgelu ? (const T*)g.aux : (const T*)g.res;
  const int ldx = dgelu ? g.ldaux : g.ldr;
  const size_t slab = (size_t)g.M * g.N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / nch), n = (int)(idx - (int64_t)m * nch) * 8;
    const float* src = g.ws + (size_t)m * g.N + n;
    float4v lo = *reinterpret_cast<const float4v*>(src), hi = *reinterpret_cast<const float4v*>(src + 4);
    for (int s = 1; s < g.splits; ++s) {
      lo += *reinterpret_cast<const float4v*>(src + s * slab);
      hi += *reinterpret_cast<const float4v*>(src + s * slab + 4);
    }
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_DROP_RES || EPI == EPI_BIAS_TANH)
      unpack8<T>(*reinterpret_cast<const short8*>((const T*)g.bias + n), bias);
    short8 x = short8{0, 0, 0, 0, 0, 0, 0, 0};
    if (dgelu || EPI == EPI_BIAS_DROP_RES || EPI == EPI_ADD_RES) x = *reinterpret_cast<const short8*>(ext + (size_t)m * ldx + n);
    K::template epilogue8<EPI>(g, m, n, v, bias, x, dk, EPI == EPI_BIAS_DROP_RES ? drop_row(g.rng, (uint32_t)m) : 0u);
  }
}

template <typename T, int EPI>
static int launch_splitk_epilogue(const PdtGemm& g, hipStream_t stream) {
  const int64_t total = (int64_t)g.M * (g.N >> 3);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL((splitk_epilogue_kernel<T, EPI>), dim3(grid), dim3(256), 0, stream, g);
  return (int)hipGetLastError();
}

// split-K forward / dgrad: the GEMM kernel writes fp32 slabs into g.ws, then the epilogue kernel finishes
template <typename T, typename LaunchFn>
static int launch_split(PdtGemm& g, hipStream_t stream, LaunchFn&& launch_gemm) {
  PdtGemm part = g;
  part.C = g.ws;
  part.ldc = g.N;
  part.epi = PDT_EPI_SLAB_F32;
  part.colsum = nullptr;
  if (int e = launch_gemm(part)) return e;
  switch (g.epi) {
    case PDT_EPI_PLAIN: return launch_splitk_epilogue<T, EPI_PLAIN>(g, stream);
#if !PDT_AKS && !PDT_BKS
    case PDT_EPI_BIAS: return launch_splitk_epilogue<T, EPI_BIAS>(g, stream);
    case PDT_EPI_BIAS_GELU: return launch_splitk_epilogue<T, EPI_BIAS_GELU>(g, stream);
    case PDT_EPI_BIAS_DROP_RES: return launch_splitk_epilogue<T, EPI_BIAS_DROP_RES>(g, stream);
    case PDT_EPI_BIAS_TANH: return launch_splitk_epilogue<T, EPI_BIAS_TANH>(g, stream);
#endif
#if !PDT_AKS && PDT_BKS
    case PDT_EPI_MUL_DGELU: return launch_splitk_epilogue<T, EPI_MUL_DGELU>(g, stream);
    case PDT_EPI_ADD_RES: return launch_splitk_epilogue<T, EPI_ADD_RES>(g, stream);
#endif
    default: return -5;
  }
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES = 2>
static int launch_tile(PdtGemm& g, hipStream_t stream) {
  using K = GemmKernel<T, BM, BN, WM, WN, PDT_AKS, PDT_BKS, STAGES>;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, PDT_AKS, PDT_BKS, STAGES>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  g.colsum_rows = 2 * g.tiles_m;
  auto launch = [&](PdtGemm& gg) {
    const int grid = gg.tiles_m * gg.tiles_n * gg.splits;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(K::NT), K::LDS_BYTES, stream, gg);
    return (int)hipGetLastError();
  };
  if (g.epi != PDT_EPI_SLAB_F32 && g.ws != nullptr) return launch_split<T>(g, stream, launch);  // splits set by the dispatcher
  if (g.epi != PDT_EPI_SLAB_F32) {
    g.splits = 1;
    g.k_per_split = ((g.K + 63) / 64) * 64;
  }
  return launch(g);
}

constexpr int kPersistentWorkgroups = 256;  // one per CU (a workgroup owns all 160 KiB of LDS)
constexpr int kPersistentMinTiles = 4 * kPersistentWorkgroups;

// Ticket counters of persistent launches: 8 words (one per XCD) per stream - launches on one stream are ordered,
// and the kernel leaves the counters re-armed (2 fixed tiles per workgroup already handed out) when it ends.
// Returns nullptr when the stream is being captured and has no counters yet (no allocation inside a capture).
static uint32_t* persistent_counters(hipStream_t stream) {
  static std::mutex mu;
  static std::unordered_map<hipStream_t, uint32_t*> table;
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find(stream);
  if (it != table.end()) return it->second;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  uint32_t* p = nullptr;
  if (hipMalloc(&p, 8 * sizeof(uint32_t)) != hipSuccess) return nullptr;
  uint32_t init[8];
  for (auto& w : init) w = 2u * (kPersistentWorkgroups / 8);
  if (hipMemcpy(p, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return nullptr; }
  table.emplace(stream, p);
  return p;
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
  g.colsum_rows = 2 * g.tiles_m;
  if (g.epi != PDT_EPI_SLAB_F32) {
    g.splits = 1;
    g.k_per_split = ((g.K + 63) / 64) * 64;
  }
  // straight-line epilogue for full tiles (gemm8_epilogue.hpp): byte offsets into C / res / aux are 32-bit there
  {
    const int e = g.epi;
    auto fits = [&](int ld) { return (int64_t)g.M * ld * 2 < 0x7FFFFFFFll; };
    const bool kind = e == PDT_EPI_PLAIN || e == PDT_EPI_BIAS || e == PDT_EPI_BIAS_GELU || e == PDT_EPI_BIAS_DROP_RES ||
                      e == PDT_EPI_MUL_DGELU || e == PDT_EPI_ADD_RES;
    const bool side = (e != PDT_EPI_BIAS_DROP_RES && e != PDT_EPI_ADD_RES) || (g.res != nullptr && fits(g.ldr) && g.ldr >= g.N);
    const bool aux = (e != PDT_EPI_BIAS_GELU && e != PDT_EPI_MUL_DGELU) || (g.aux != nullptr && fits(g.ldaux) && g.ldaux >= g.N);
    g.fast_epi = (pdt_gemm_fast_epi_mode() != 0 && kind && side && aux && fits(g.ldc) && g.ldc >= g.N) ? 1 : 0;
  }
  int grid = g.tiles_m * g.tiles_n * g.splits;
  // persistent launch (one workgroup per CU walks several output tiles, see Gemm8::run): needs every tile full, an
  // even number of K-tiles, no K split and at least kPersistentMinTiles tiles (the kernel itself needs two per
  // workgroup; with fewer than four the cross-tile prefetch has too few tile boundaries to pay for the ticket
  // traffic, and the 8192-12288-token micro-batches that fall below it are the ones whose weight gradients run
  // concurrently on a second stream); PDT_GEMM_PERSIST=0 turns it off, =2 lowers the bound to two tiles per workgroup
  const int persist_mode = pdt_gemm_persist_mode();
  const bool allow_persist = persist_mode != 0;
  const int min_tiles = persist_mode == 2 ? 2 * kPersistentWorkgroups : kPersistentMinTiles;
  g.sync = nullptr;
  if (allow_persist && g.epi != PDT_EPI_SLAB_F32 && grid >= min_tiles && g.M % 256 == 0 && g.N % 256 == 0 &&
      g.K % 128 == 0 && (g.sync = persistent_counters(stream)) != nullptr)
    grid = kPersistentWorkgroups;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(K::NT), K::LDS_BYTES, stream, g);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_dtype(int tile, PdtGemm& g, hipStream_t stream) {
  if (tile == 2) return launch_pingpong<T>(g, stream);
  if (tile == 1) return launch_tile<T, 256, 256, 2, 4>(g, stream);
  if (tile == 3) return launch_tile<T, 128, 128, 2, 2, 4>(g, stream);
  // short micro-batches (few 128x128 tiles for 256 CUs): 64-row tiles on the ring kernel, 32 x 64 per wave.  A
  // K-strided operand image needs a 256-B pitch for its swizzle (128 columns), so the weight gradients and the
  // 64-column variant of the data gradients do not exist.
  if (tile == 4 && !PDT_AKS) return launch_tile<T, 64, 128, 2, 2, 4>(g, stream);
#if !PDT_AKS && !PDT_BKS
  if (tile == 5) return launch_tile<T, 64, 64, 2, 1, 4>(g, stream);
#endif
  if (tile == 4 || tile == 5) return launch_tile<T, 128, 128, 2, 2, 4>(g, stream);
  return launch_tile<T, 128, 128, 2, 2>(g, stream);
}

}  // namespace pdt

extern "C" int PDT_LAYOUT_FN(int dtype, int tile, PdtGemm* g, void* stream) {
  if (dtype == PDT_F16) return pdt::launch_dtype<_Float16>(tile, *g, (hipStream_t)stream);
  if (dtype == PDT_BF16) return pdt::launch_dtype<__bf16>(tile, *g, (hipStream_t)stream);
  return -1;
}
