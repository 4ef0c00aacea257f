// Host-side GEMM dispatch: argument checks, tile / split-K selection, split-K reduction kernel.
#include "common.hpp"

extern "C" int pdt_gemm_nt(int dtype, int tile, PdtGemm* g, void* stream);
extern "C" int pdt_gemm_nn(int dtype, int tile, PdtGemm* g, void* stream);
extern "C" int pdt_gemm_tn(int dtype, int tile, PdtGemm* g, void* stream);
extern "C" int pdt_gemm_persist_mode(void);

namespace pdt {
constexpr int kMaxSplits = 32;
constexpr int kNumCU = 256;

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n4,
                                                            float* __restrict__ out, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4v s = accumulate ? reinterpret_cast<const float4v*>(out)[i] : float4v{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) s += reinterpret_cast<const float4v*>(slab)[(int64_t)k * n4 + i];
    reinterpret_cast<float4v*>(out)[i] = s;
  }
}

// tile variant for a shape (tile >= 0: the caller's choice)
static int choose_tile(int layout, int tile, const PdtGemm* g) {
  if (tile == 4 || tile == 5) {  // 64-row tiles exist for K-contiguous A (4) and for forward GEMMs only (5)
    if (layout == PDT_GEMM_TN) return 3;
    if (layout == PDT_GEMM_NN) return 4;
  }
  if (tile >= 0) return tile;
  // measured on MI355X (bench/gemm_bench.py, BERT-large shapes, 1024..32768 tokens): the ping-pong 256x256
  // kernel wins for forward / dgrad once its tiles fill at least 3/4 of the 256 CUs (at exactly 128 tiles the
  // 128x128 kernel, two workgroups per CU, is 20-30 % ahead); for weight gradients (split-K fills the CUs
  // whatever the tile count) it wins from K = 8192 tokens on, for a 1024x1024 output only from 32768
  const int64_t t256 = (int64_t)((g->M + 255) / 256) * ((g->N + 255) / 256);
  // 128x128 tiles: with at most ~1.25 workgroups per CU (short micro-batches: 1024 tokens give 64-256 tiles) the
  // 4-deep K-tile ring, otherwise the double-buffered kernel at two workgroups per CU.  In place in a BERT-large
  // step of 12 x 8 samples the ring is worth 5.6 % (915 -> 967 samples/s) - operands come from HBM there; in a
  // back-to-back micro-benchmark, where they sit in L2, it is not.  PDT_GEMM_DEEP=<tiles> moves the bound (0: never).
  static const int deep = [] { const char* e = getenv("PDT_GEMM_DEEP"); return e ? atoi(e) : 320; }();
  const int64_t t128 = (int64_t)((g->M + 127) / 128) * ((g->N + 127) / 128);
  int small = (deep > 0 && t128 <= deep) ? 3 : 0;
  // a quarter of the CUs or fewer in 128x128 tiles (1024 tokens x 1024 outputs): 64x128 tiles on the same ring -
  // 9.4 against 14.8 us for 1024^3 in a back-to-back run (hipBLASLt's 64x64 kernel: 8.6); wider outputs (192-256
  // tiles of 128x128) are 15-20 % slower in 64-row tiles.  PDT_GEMM_SMALL64=<tiles> moves the bound (0: never).
  static const int small64 = [] { const char* e = getenv("PDT_GEMM_SMALL64"); return e ? atoi(e) : 64; }();
  if (small == 3 && layout != PDT_GEMM_TN && t128 <= small64) small = 4;
  if (layout == PDT_GEMM_TN)
    return (g->M >= 256 && g->N >= 256 && g->K >= 8192 && (t256 >= 32 || g->K >= 32768)) ? 2 : small;
  return t256 >= 192 ? 2 : small;
}

// K split of a weight gradient: fills g->splits / g->k_per_split (g->splits <= 0: choose)
static void plan_tn(int tile, PdtGemm* g) {
  const int bt = (tile == 0 || tile == 3) ? 128 : 256;
  const int tiles = ((g->M + bt - 1) / bt) * ((g->N + bt - 1) / bt);
  const int ktiles = (g->K + 63) / 64;
  if (g->splits <= 0) {
    // one full round of workgroups: tiles * splits <= CUs (x2 for the 128x128 kernel, 2 workgroups per CU)
    const int target = tile == 0 ? 2 * kNumCU : kNumCU;
    int s = target / tiles;
    s = s < 1 ? 1 : (s > kMaxSplits ? kMaxSplits : s);
    if (s > ktiles) s = ktiles;
    // a short K (micro-batches of up to 2048 tokens) on the ring kernel is not split at all: the tile goes
    // straight into the gradient (api.h: slab_accumulate) and the launch of a reduction kernel is saved - these
    // steps are bound by the number of launches, not by idle CUs
    if (tile == 3 && ktiles <= 32) s = 1;
    g->splits = s;
  }
  const int kt_per = (ktiles + g->splits - 1) / g->splits;
  g->k_per_split = kt_per * 64;
  g->splits = (ktiles + kt_per - 1) / kt_per;  // no empty split
}
}  // namespace pdt

extern "C" int pdt_gemm_max_splits(void) { return pdt::kMaxSplits; }

static int g_persist_mode = -1;  // -1: not decided yet (read PDT_GEMM_PERSIST at first use)
extern "C" int pdt_gemm_persist_mode(void) {
  if (g_persist_mode < 0) {
    const char* e = getenv("PDT_GEMM_PERSIST");
    g_persist_mode = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
  }
  return g_persist_mode;
}
extern "C" void pdt_gemm_set_persist_mode(int mode) { g_persist_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }

static int g_fast_epi = -1;  // -1: not decided yet (read PDT_GEMM_FAST_EPI at first use)
extern "C" int pdt_gemm_fast_epi_mode(void) {
  if (g_fast_epi < 0) {
    const char* e = getenv("PDT_GEMM_FAST_EPI");
    g_fast_epi = (e && e[0] == '0') ? 0 : 1;
  }
  return g_fast_epi;
}
extern "C" void pdt_gemm_set_fast_epi_mode(int on) { g_fast_epi = on ? 1 : 0; }

// number of fp32 slabs a weight-gradient GEMM of this shape will write (what the caller must allocate)
// K split of forward / dgrad GEMMs with few output tiles (pdt_gemm_splits): on by default - a lone short micro-batch
// is latency-bound and a split K shortens every GEMM of the dependent chain (BERT-large, batch 8: 809 vs 745
// samples/s; with two micro-batches in flight on two streams, Engine.accumulate, 12 x 8: 1453 vs 1397).
static int g_split_fwd = 1;
extern "C" void pdt_gemm_set_split_mode(int on) { g_split_fwd = on ? 1 : 0; }
extern "C" int pdt_gemm_split_mode(void) { return g_split_fwd; }

extern "C" int pdt_gemm_tn_splits(int tile, const PdtGemm* g) {
  PdtGemm c = *g;
  pdt::plan_tn(pdt::choose_tile(PDT_GEMM_TN, tile, &c), &c);
  return c.splits;
}

// Forward / dgrad GEMMs whose output has few tiles but whose K is long (e.g. 1024 tokens x 1024 outputs x 4096)
// leave most CUs idle with one workgroup per output tile: 128x128 tiles fill 2 x 256 workgroup slots, so cut K
// until the slots are ~full, keeping at least 16 K-tiles (K = 1024) per slice - shorter slices lose more to the slab
// round trip than they gain.  Measured against one workgroup per tile on
// MI355X: 1.5-2x at 1024 tokens, ~1.1x at 4096 tokens, nothing to gain once the tiles fill the chip.
extern "C" int pdt_gemm_splits(int layout, int tile, const PdtGemm* g) {
  using namespace pdt;
  if (layout == PDT_GEMM_TN || g->colsum != nullptr || g->epi == PDT_EPI_SLAB_F32) return 1;
  const int64_t t256 = (int64_t)((g->M + 255) / 256) * ((g->N + 255) / 256);
  if ((tile > 0 && tile != 3) || (tile < 0 && t256 >= 192)) return 1;  // only the 128x128 kernels take this path
  const int64_t t128 = (int64_t)((g->M + 127) / 128) * ((g->N + 127) / 128);
  const int ktiles = (g->K + 63) / 64;
  if (!g_split_fwd) return 1;
  static const int min_kt = [] { const char* e = getenv("PDT_SPLIT_MIN_KT"); return e && atoi(e) > 0 ? atoi(e) : 16; }();
  int s = (int)((2 * kNumCU) / t128);
  if (s > ktiles / min_kt) s = ktiles / min_kt;
  if (s > 8) s = 8;
  return s < 2 ? 1 : s;
}

extern "C" int pdt_gemm(int dtype, int layout, int tile, PdtGemm* g, void* stream) {
  using namespace pdt;
  if (g->M <= 0 || g->N <= 0 || g->K <= 0) return -2;
  if ((g->N & 7) || (g->lda & 7) || (g->ldb & 7) || (g->ldc & 7)) return -3;  // 16-B vector access
  if (layout != PDT_GEMM_TN && (g->K & 7)) return -3;
  if (layout == PDT_GEMM_TN && (g->M & 7)) return -3;
  // operand windows are addressed with 32-bit byte offsets; 0x80000000 marks an out-of-range lane
  const int64_t a_bytes = 2ll * (layout == PDT_GEMM_TN ? (int64_t)g->K * g->lda : 256ll * g->lda);
  const int64_t b_bytes = 2ll * (layout == PDT_GEMM_NT ? 256ll * g->ldb : (int64_t)g->K * g->ldb);
  if (a_bytes >= 0x7FFF0000ll || b_bytes >= 0x7FFF0000ll) return -4;
  tile = choose_tile(layout, tile, g);
  if (layout == PDT_GEMM_TN) {
    if (g->epi != PDT_EPI_SLAB_F32) return -5;
    plan_tn(tile, g);
    if (g->slab_accumulate && g->splits != 1) return -5;
    return pdt_gemm_tn(dtype, tile, g, stream);
  }
  // each operand layout carries the epilogues of its role (forward: bias family; data gradient: GELU-grad
  // multiply / residual add)
  const int e = g->epi;
  const bool fwd_epi = e == PDT_EPI_BIAS || e == PDT_EPI_BIAS_GELU || e == PDT_EPI_BIAS_DROP_RES || e == PDT_EPI_BIAS_TANH;
  const bool dgrad_epi = e == PDT_EPI_MUL_DGELU || e == PDT_EPI_ADD_RES;
  if (e != PDT_EPI_PLAIN && !(layout == PDT_GEMM_NT && fwd_epi) && !(layout == PDT_GEMM_NN && dgrad_epi)) return -5;
  if (g->ws != nullptr && g->ws_splits > 1) {
    if (g->colsum != nullptr || g->ws_splits > kMaxSplits) return -5;
    tile = 0;
    const int ktiles = (g->K + 63) / 64;
    const int kt_per = (ktiles + g->ws_splits - 1) / g->ws_splits;
    g->k_per_split = kt_per * 64;
    g->splits = (ktiles + kt_per - 1) / kt_per;  // no empty slice
  } else {
    g->ws = nullptr;
  }
  return layout == PDT_GEMM_NT ? pdt_gemm_nt(dtype, tile, g, stream) : pdt_gemm_nn(dtype, tile, g, stream);
}

extern "C" int pdt_splitk_reduce(const float* slab, int splits, int64_t n, float* out, int accumulate, void* stream) {
  if (n & 3) return -3;
  const int64_t n4 = n >> 2;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pdt::splitk_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, splits, n4, out, accumulate);
  return (int)hipGetLastError();
}
