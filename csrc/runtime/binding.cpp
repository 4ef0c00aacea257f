// Torch-facing runtime of the gfx950 kernel library: tensor checks, stream plumbing, and the fused
// encoder-layer forward/backward that chains the kernels (the per-layer "schedule" lives here in C++,
// so one Python call issues all launches of a layer on the current HIP stream).
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/HIPEvent.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../kernels/api.h"

namespace py = pybind11;
using at::Tensor;

namespace {

// PyTorch-ROCm names its device type "cuda"; these are the guard / stream classes that accept it.
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

// Weight-gradient lane: in the layer backward the wgrad GEMM of a projection does not feed anything later in the
// layer, so it can be issued on a second HIP stream and run concurrently with the dgrad chain.  Measured on
// MI355X (BERT-large, seq 128): +5 % step throughput at 8192 tokens per micro-batch and +1.6 % at 4096 (the
// GEMMs of such a micro-batch leave CUs idle in their tail rounds, the concurrent kernel fills them), but
// -1.3 % at 16384 and -3.4 % at 32768 tokens, where every GEMM fills the chip by itself and two of them only
// compete for L2.  Mode 1 (default) therefore uses the lane up to 12288 tokens; PDT_WGRAD_STREAM=0 never, =2
// always.  Fork = event on the main stream the side stream
// waits for; join = event on the side stream the main stream waits for before the layer returns, which also
// makes the caching allocator's stream-ordered reuse of this layer's temporaries safe.
using Stream = c10::hip::HIPStreamMasqueradingAsCUDA;
using StreamGuard = c10::hip::HIPStreamGuardMasqueradingAsCUDA;
int g_wgrad_mode = [] {
  const char* e = std::getenv("PDT_WGRAD_STREAM");
  return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
}();
constexpr int kWgradLaneMaxTokens = 12288;
// Deferred mode (set_wgrad_defer(true), used by Engine.accumulate for every backward of an accumulation window but
// the last): short micro-batches are bound by how fast the host can enqueue, and the four forks and the join of a
// layer are 10 of its ~25 runtime calls.  A deferred lane queues the layer's weight-gradient GEMMs, issues them
// behind ONE fork at the end of the layer and does not join: nothing on the main stream reads those gradients
// before the backward pass is over.  The tensors the queued GEMMs read are kept alive (per main stream) until
// wgrad_join() has made the main stream wait for the side stream, which the caller does once, after backward().
bool g_wgrad_defer = false;
std::mutex g_defer_mu;
std::unordered_map<int64_t, std::vector<Tensor>> g_defer_held;  // by main-stream id
class WgradLane {
 public:
  explicit WgradLane(bool enabled, bool defer = false)
      : on_(enabled), defer_(enabled && defer), main_(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA()), side_(main_) {
    if (on_) side_ = side_stream(main_.device_index());
  }
  // run `fn` (kernel launches + its temporaries) on the side stream, after everything issued so far on main
  // (deferred: after everything issued on main up to finish()); `reads` are the tensors it reads
  template <typename F>
  void run(F&& fn, std::initializer_list<Tensor> reads = {}) {
    if (!on_) { fn(); return; }
    if (defer_) {
      pending_.emplace_back(std::forward<F>(fn));
      keep_.insert(keep_.end(), reads.begin(), reads.end());
      return;
    }
    event(next_++).record(main_);
    event(next_ - 1).block(side_);
    StreamGuard g(side_);
    fn();
    used_ = true;
  }
  // end of the layer: join (or, deferred, fork once and issue what was queued)
  void finish() {
    if (defer_ && !pending_.empty()) {
      event(next_++).record(main_);
      event(next_ - 1).block(side_);
      {
        StreamGuard g(side_);
        for (auto& fn : pending_) fn();
      }
      pending_.clear();
      std::lock_guard<std::mutex> lock(g_defer_mu);
      auto& held = g_defer_held[(int64_t)main_.id()];
      held.insert(held.end(), keep_.begin(), keep_.end());
      keep_.clear();
      return;
    }
    join();
  }
  void join() {
    if (!on_ || !used_) return;
    event(next_++).record(side_);
    event(next_ - 1).block(main_);
    used_ = false;
  }
  // (queued launches are only ever issued by finish(): they capture the caller's locals, which are gone by the time
  // a destructor runs during stack unwinding)
  ~WgradLane() { pending_.clear(); join(); }
  // make the current stream wait for the side stream and drop the tensors kept for it (see g_wgrad_defer)
  static void join_deferred() {
    Stream main = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA();
    std::vector<Tensor> drop;
    {
      std::lock_guard<std::mutex> lock(g_defer_mu);
      auto it = g_defer_held.find((int64_t)main.id());
      if (it == g_defer_held.end() || it->second.empty()) return;
      drop.swap(it->second);
    }
    Stream side = side_stream(main.device_index());
    at::cuda::CUDAEvent& e = event(15);
    e.record(side);
    e.block(main);
    // (`drop` is released here: the caching allocator hands the blocks back to the stream they belong to, whose
    // later work is ordered behind the wait above)
  }

 private:
  static Stream side_stream(c10::DeviceIndex dev) {
    static std::vector<std::unique_ptr<Stream>> streams(64);
    if (!streams[dev]) streams[dev] = std::make_unique<Stream>(c10::hip::getStreamFromPoolMasqueradingAsCUDA(false, dev));
    return *streams[dev];
  }
  // events are re-recorded every layer: a small per-thread ring, never destroyed while work may reference them
  static at::cuda::CUDAEvent& event(int i) {
    static thread_local std::vector<std::unique_ptr<at::cuda::CUDAEvent>> ring(16);
    auto& e = ring[i & 15];
    if (!e) e = std::make_unique<at::cuda::CUDAEvent>();
    return *e;
  }
  bool on_, defer_, used_ = false;
  int next_ = 0;
  Stream main_, side_;
  std::vector<std::function<void()>> pending_;
  std::vector<Tensor> keep_;
};

int dt_code(const Tensor& t) {
  if (t.scalar_type() == at::kHalf) return PDT_F16;
  if (t.scalar_type() == at::kBFloat16) return PDT_BF16;
  TORCH_CHECK(false, "expected a float16/bfloat16 tensor, got ", t.scalar_type());
}
// Serialized debug mode (PDT_DEBUG_SYNC=1 or set_debug_sync(true)): the device is drained after every native
// launch, so an asynchronous fault or a sticky error is reported at the op that caused it instead of at the
// next unrelated synchronisation point.  Kernels of different streams never overlap in this mode, which also
// makes it the A/B switch for a suspected stream-ordering bug (a failure that disappears here is an ordering bug).
bool g_debug_sync = [] {
  const char* e = std::getenv("PDT_DEBUG_SYNC");
  return e && e[0] == '1';
}();
int64_t g_debug_sync_count = 0;
void check(int err, const char* what) {
  TORCH_CHECK(err == 0, "pdt kernel launch failed in ", what, ": code ", err,
              err < 0 ? " (argument check)" : " (hipError_t)");
  if (g_debug_sync) {
    const hipError_t e = hipDeviceSynchronize();
    ++g_debug_sync_count;
    TORCH_CHECK(e == hipSuccess, "pdt kernel faulted in ", what, ": ", hipGetErrorString(e));
  }
}
void chk16(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be on the GPU");
  TORCH_CHECK(t.scalar_type() == at::kHalf || t.scalar_type() == at::kBFloat16, name, " must be 16-bit float");
  TORCH_CHECK(t.stride(-1) == 1, name, " must have unit inner stride");
  TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0, name, " must be 16-byte aligned");
}
void chkf32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), name,
              " must be a contiguous fp32 GPU tensor");
}
uint32_t thr16(double p) { return p <= 0.0 ? 0u : (uint32_t)std::lround(p * 65536.0); }
float keep_scale(double p) { return p <= 0.0 ? 1.0f : (float)(1.0 / (1.0 - p)); }

using OptTensor = c10::optional<Tensor>;
// `sids` (optional int32 [B] on the GPU): global sample ids - the dropout counters of row r of a [B*seq, H]
// activation then derive from (sids[r / seq], r % seq), see PdtRng in api.h.
// `rowmap` (optional int32 [T] on the GPU): unpadded activations - the dropout row of every packed row (PdtRng::row_table).
PdtRng make_rng(const Tensor& state, int64_t micro, int64_t site, const OptTensor& sids = c10::nullopt, int64_t seq = 1,
                int64_t batch = -1, const OptTensor& rowmap = c10::nullopt) {
  TORCH_CHECK(state.is_cuda() && state.scalar_type() == at::kLong && state.numel() >= 2, "rng state: int64[2] on GPU");
  PdtRng r{reinterpret_cast<const uint64_t*>(state.data_ptr<int64_t>()), (uint32_t)micro, (uint32_t)site, nullptr, 1u, 0u};
  if (sids.has_value()) {
    TORCH_CHECK(sids->is_cuda() && sids->scalar_type() == at::kInt && sids->is_contiguous() && sids->dim() == 1,
                "sample_ids must be a contiguous int32 [B] GPU tensor");
    TORCH_CHECK(seq >= 1 && seq <= 512, "sample-keyed dropout: at most 512 tokens per sample");
    TORCH_CHECK(batch < 0 || sids->numel() == batch, "sample_ids has ", sids->numel(), " entries for a batch of ", batch);
    TORCH_CHECK(sids->numel() * seq < (int64_t(1) << 23), "sample-keyed dropout: too many rows for the exact row / seq trick");
    r.sample_ids = sids->data_ptr<int32_t>();
    r.seq = (uint32_t)seq;
    r.seq_magic = seq == 1 ? 0u : (uint32_t)(((uint64_t(1) << 32) + (uint64_t)seq - 1) / (uint64_t)seq);
  }
  if (rowmap.has_value()) {
    TORCH_CHECK(rowmap->is_cuda() && rowmap->scalar_type() == at::kInt && rowmap->is_contiguous() && rowmap->dim() == 1,
                "rowmap must be a contiguous int32 [T] GPU tensor");
    r.row_table = reinterpret_cast<const uint32_t*>(rowmap->data_ptr<int32_t>());
  }
  return r;
}

// ---------------------------------------------------------------------------------------------------
// GEMM wrappers.  2-D operands; leading dimension = stride(0).
// ---------------------------------------------------------------------------------------------------
// Column-sum finishing jobs (bias / gamma / beta gradients) of one layer backward, launched together at its end: with
// short micro-batches a layer's backward is a chain of ~15 launches of 5-40 us each, and four of them were finishers.
// The partial-sum tensors are kept alive until the launch; flush() runs on the caller's current stream, which must be
// the one their producers ran on.
class FinishQueue {
 public:
  void add(const Tensor& partial, int nblk, int H, int nseg, float* o0, float* o1, float* o2, bool accumulate) {
    if (n_ == PDT_COLSUM_MAX_JOBS) flush();
    keep_.push_back(partial);
    jobs_[n_++] = PdtColsumJob{partial.data_ptr<float>(), {o0, o1, o2, nullptr}, nblk, H, nseg, accumulate ? 1 : 0};
  }
  void flush() {
    if (!n_) return;
    check(pdt_colsum_finish_multi(jobs_, n_, cur_stream()), "colsum_finish_multi");
    n_ = 0;
    keep_.clear();
  }
  // (no flush in the destructor: it would launch, and possibly throw, while an exception unwinds the layer)

 private:
  PdtColsumJob jobs_[PDT_COLSUM_MAX_JOBS];
  int n_ = 0;
  std::vector<Tensor> keep_;
};

struct Epi {
  int kind = PDT_EPI_PLAIN;
  const Tensor* bias = nullptr;
  const Tensor* res = nullptr;
  Tensor* aux = nullptr;
  double drop_p = 0.0;
  PdtRng rng{nullptr, 0, 0};
  Tensor* colsum = nullptr;  // fp32 [N]: (+)= column sums of the stored output (a bias gradient)
  bool colsum_accumulate = true;
  FinishQueue* fq = nullptr;  // the finishing pass of `colsum` is queued here instead of being launched at once
};

void run_gemm(int layout, const Tensor& A, const Tensor& B, Tensor& C, int M, int N, int K, const Epi& e, int tile = -1) {
  PdtGemm g{};
  g.A = A.data_ptr(); g.B = B.data_ptr(); g.C = C.data_ptr();
  g.M = M; g.N = N; g.K = K;
  g.lda = (int)A.stride(0); g.ldb = (int)B.stride(0); g.ldc = (int)C.stride(0);
  g.epi = e.kind;
  if (e.bias) { chk16(*e.bias, "bias"); g.bias = e.bias->data_ptr(); }
  if (e.res) { chk16(*e.res, "res"); g.res = e.res->data_ptr(); g.ldr = (int)e.res->stride(0); }
  if (e.aux) { chk16(*e.aux, "aux"); g.aux = e.aux->data_ptr(); g.ldaux = (int)e.aux->stride(0); }
  g.drop_thr = thr16(e.drop_p);
  g.drop_scale = keep_scale(e.drop_p);
  g.rng = e.rng;
  // column sums: the kernel stores per-wave-row partial sums, a finishing kernel adds them up
  Tensor cs_part;
  if (e.colsum) {
    chkf32(*e.colsum, "colsum");
    TORCH_CHECK(e.colsum->numel() == N, "colsum must have N elements");
    cs_part = at::empty({2 * (((int64_t)M + 63) / 64), (int64_t)N}, C.options().dtype(at::kFloat));
    g.colsum = cs_part.data_ptr<float>();
  }
  // few output tiles + long K: split K over fp32 partial slabs, summed by the epilogue kernel
  Tensor ws;
  const int splits = pdt_gemm_splits(layout, tile, &g);
  if (splits > 1) {
    ws = at::empty({splits, (int64_t)M, (int64_t)N}, C.options().dtype(at::kFloat));
    g.ws = ws.data_ptr<float>();
    g.ws_splits = splits;
  }
  check(pdt_gemm(dt_code(A), layout, tile, &g, cur_stream()), "gemm");
  if (e.colsum && e.fq)
    e.fq->add(cs_part, g.colsum_rows, N, 1, e.colsum->data_ptr<float>(), nullptr, nullptr, e.colsum_accumulate);
  else if (e.colsum)
    check(pdt_colsum_finish(cs_part.data_ptr<float>(), g.colsum_rows, N, 1, e.colsum->data_ptr<float>(), nullptr, nullptr,
                            nullptr, e.colsum_accumulate ? 1 : 0, cur_stream()), "colsum_finish");
}

// C[M,N] = A[M,K] * W[N,K]^T (+ epilogue)
Tensor gemm_nt(const Tensor& A, const Tensor& W, const Epi& e, int tile = -1) {
  chk16(A, "A"); chk16(W, "W");
  const int M = (int)A.size(0), K = (int)A.size(1), N = (int)W.size(0);
  TORCH_CHECK(W.size(1) == K, "gemm_nt: inner dims differ");
  Tensor C = at::empty({M, N}, A.options());
  run_gemm(PDT_GEMM_NT, A, W, C, M, N, K, e, tile);
  return C;
}
// C[M,N] = A[M,K] * W[K,N] (+ epilogue)
Tensor gemm_nn(const Tensor& A, const Tensor& W, const Epi& e, int tile = -1) {
  chk16(A, "A"); chk16(W, "W");
  const int M = (int)A.size(0), K = (int)A.size(1), N = (int)W.size(1);
  TORCH_CHECK(W.size(0) == K, "gemm_nn: inner dims differ");
  Tensor C = at::empty({M, N}, A.options());
  run_gemm(PDT_GEMM_NN, A, W, C, M, N, K, e, tile);
  return C;
}
// G[M,N] (fp32, += when accumulate) = A[K,M]^T * B[K,N]  -- weight gradients; split-K slabs + reduce
void gemm_tn_into(const Tensor& A, const Tensor& B, Tensor& G, bool accumulate, int tile = -1, int splits = 0) {
  chk16(A, "A"); chk16(B, "B"); chkf32(G, "G");
  const int K = (int)A.size(0), M = (int)A.size(1), N = (int)B.size(1);
  TORCH_CHECK(B.size(0) == K && G.size(0) == M && G.size(1) == N, "gemm_tn: shape mismatch");
  // The kernels address an operand through a 32-bit byte offset into a window that, for a weight gradient, spans the
  // whole token range: more than ~2 GiB of either operand (e.g. 262144 tokens x 4096 columns) goes in token chunks,
  // each added to the gradient.
  const int64_t pitch = 2 * std::max<int64_t>(A.stride(0), B.stride(0));
  const int64_t max_rows = ((int64_t)0x7FFF0000 / pitch - 1) / 64 * 64;
  if (K > max_rows) {
    TORCH_CHECK(max_rows >= 64, "gemm_tn: operand rows too long for the 32-bit window");
    for (int64_t k0 = 0; k0 < K; k0 += max_rows) {
      const int64_t len = std::min<int64_t>(max_rows, K - k0);
      gemm_tn_into(A.narrow(0, k0, len), B.narrow(0, k0, len), G, accumulate || k0 > 0, tile, splits);
    }
    return;
  }
  PdtGemm g{};
  g.A = A.data_ptr(); g.B = B.data_ptr();
  g.M = M; g.N = N; g.K = K;
  g.lda = (int)A.stride(0); g.ldb = (int)B.stride(0); g.ldc = N;
  g.epi = PDT_EPI_SLAB_F32;
  g.splits = splits;
  g.splits = pdt_gemm_tn_splits(tile, &g);
  if (g.splits == 1) {  // K is not split: the kernel writes (or adds to) the gradient itself
    g.C = G.data_ptr();
    g.slab_accumulate = accumulate ? 1 : 0;
    check(pdt_gemm(dt_code(A), PDT_GEMM_TN, tile, &g, cur_stream()), "gemm_tn");
    return;
  }
  Tensor slab = at::empty({(int64_t)g.splits * M * N}, G.options());
  g.C = slab.data_ptr();
  check(pdt_gemm(dt_code(A), PDT_GEMM_TN, tile, &g, cur_stream()), "gemm_tn");
  check(pdt_splitk_reduce(slab.data_ptr<float>(), g.splits, (int64_t)M * N, G.data_ptr<float>(), accumulate ? 1 : 0,
                          cur_stream()), "splitk_reduce");
}

// bias gradient of a 16-bit [rows, ncols] matrix into fp32 out[ncols]
void colsum_into(const Tensor& X, Tensor& out, bool accumulate) {
  chk16(X, "X"); chkf32(out, "out");
  const int rows = (int)X.size(0), ncols = (int)X.size(1);
  Tensor partial = at::empty({256, ncols}, out.options());
  int nblk = 0;
  check(pdt_colsum_partial(dt_code(X), X.data_ptr(), rows, ncols, (int)X.stride(0), partial.data_ptr<float>(), &nblk,
                           cur_stream()), "colsum_partial");
  check(pdt_colsum_finish(partial.data_ptr<float>(), nblk, ncols, 1, out.data_ptr<float>(), nullptr, nullptr, nullptr,
                          accumulate ? 1 : 0, cur_stream()), "colsum_finish");
}

struct LnBwdOut { Tensor dx, ddrop; };
// LayerNorm backward + (optional) dropout-masked copy + gamma/beta/bias gradients
LnBwdOut ln_bwd(const Tensor& dy, const Tensor& x, const Tensor& stats, const Tensor& gamma, Tensor* g_gamma,
                Tensor* g_beta, Tensor* g_bias, double drop_p, PdtRng rng, bool accumulate, FinishQueue* fq = nullptr) {
  const int rows = (int)x.size(0), H = (int)x.size(1);
  LnBwdOut o;
  o.dx = at::empty_like(x);
  const bool masked = drop_p > 0.0;
  o.ddrop = masked ? at::empty_like(x) : o.dx;
  Tensor partial = at::empty({(int64_t)pdt_layernorm_bwd_max_blocks() * 3 * H}, stats.options());
  int nblk = 0;
  check(pdt_layernorm_bwd(dt_code(x), dy.data_ptr(), x.data_ptr(), stats.data_ptr<float>(), gamma.data_ptr(),
                          o.dx.data_ptr(), masked ? o.ddrop.data_ptr() : nullptr, partial.data_ptr<float>(), &nblk,
                          rows, H, keep_scale(drop_p), thr16(drop_p), rng, cur_stream()), "layernorm_bwd");
  float* o0 = g_gamma ? g_gamma->data_ptr<float>() : nullptr;
  float* o1 = g_beta ? g_beta->data_ptr<float>() : nullptr;
  float* o2 = g_bias ? g_bias->data_ptr<float>() : nullptr;
  if (fq)
    fq->add(partial, nblk, H, 3, o0, o1, o2, accumulate);
  else
    check(pdt_colsum_finish(partial.data_ptr<float>(), nblk, H, 3, o0, o1, o2, nullptr, accumulate ? 1 : 0, cur_stream()),
          "colsum_finish");
  return o;
}

Tensor ln_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, Tensor& stats, double eps) {
  Tensor y = at::empty_like(x);
  check(pdt_layernorm_fwd(dt_code(x), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                          stats.data_ptr<float>(), (int)x.size(0), (int)x.size(1), (float)eps, cur_stream()),
        "layernorm_fwd");
  return y;
}

// ---------------------------------------------------------------------------------------------------
// Encoder layer.  Weight list order (16-bit compute copies):
//   0 wqkv[3H,H] 1 bqkv[3H] 2 wo[H,H] 3 bo[H] 4 ln1_g 5 ln1_b 6 w1[F,H] 7 b1[F] 8 w2[H,F] 9 b2[H] 10 ln2_g 11 ln2_b
// Gradient list: the same order, fp32 views of the gradient arena.
// Saved list: 0 qkv 1 ctx 2 lse 3 z1 4 stats1 5 x1 6 u 7 gact 8 z2 9 stats2
// ---------------------------------------------------------------------------------------------------
struct LayerDims { int B, S, A; double p_attn, p_hidden, eps; int64_t micro, layer; };

std::vector<Tensor> layer_forward(const Tensor& x, const Tensor& mask, const Tensor& kv_len,
                                  const std::vector<Tensor>& w, const Tensor& rng_state, int64_t B, int64_t S, int64_t A,
                                  double p_attn, double p_hidden, double eps, int64_t micro, int64_t layer,
                                  const OptTensor& sids, const OptTensor& cu, const OptTensor& rowmap) {
  DeviceGuard guard(x.device());
  chk16(x, "x");
  TORCH_CHECK(w.size() == 12, "layer_forward: 12 weight tensors expected");
  const int T = (int)x.size(0), H = (int)x.size(1);
  // unpadded batch: x is the packed [T, H] of B samples whose first rows are cu[0..B) (cu[B] = T), S the longest one
  const int32_t* cup = nullptr;
  if (cu.has_value()) {
    TORCH_CHECK(cu->is_cuda() && cu->scalar_type() == at::kInt && cu->is_contiguous() && cu->numel() == B + 1,
                "cu_seqlens must be a contiguous int32 [B+1] GPU tensor");
    TORCH_CHECK(rowmap.has_value() && rowmap->numel() == T, "an unpadded batch needs the per-row dropout table");
    TORCH_CHECK(T <= B * S && H == A * 64, "layer_forward: packed x has more rows than B * S");
    cup = cu->data_ptr<int32_t>();
  } else {
    TORCH_CHECK(T == B * S && H == A * 64, "layer_forward: head_dim must be 64 and x must be [B*S, H]");
  }
  auto f32 = x.options().dtype(at::kFloat);
  const int64_t site = layer * 4;

  Epi e_bias; e_bias.kind = PDT_EPI_BIAS; e_bias.bias = &w[1];
  Tensor qkv = gemm_nt(x, w[0], e_bias);

  Tensor ctx = at::empty({T, H}, x.options());
  Tensor lse = at::empty({B, A, S}, f32);
  check(pdt_attn_fwd(dt_code(x), qkv.data_ptr(), mask.data_ptr<int32_t>(), kv_len.data_ptr<int32_t>(), cup, ctx.data_ptr(),
                     lse.data_ptr<float>(), (int)B, (int)S, (int)A, keep_scale(p_attn), thr16(p_attn),
                     make_rng(rng_state, micro, site + 0, sids, S, B), cur_stream()), "attn_fwd");

  Epi e_ao; e_ao.kind = PDT_EPI_BIAS_DROP_RES; e_ao.bias = &w[3]; e_ao.res = &x; e_ao.drop_p = p_hidden;
  e_ao.rng = make_rng(rng_state, micro, site + 1, sids, S, B, rowmap);
  Tensor z1 = gemm_nt(ctx, w[2], e_ao);
  Tensor stats1 = at::empty({T, 2}, f32);
  Tensor x1 = ln_fwd(z1, w[4], w[5], stats1, eps);

  Tensor u = at::empty({T, (int64_t)w[6].size(0)}, x.options());  // gelu'(FFN1 pre-activation), for backward
  Epi e_f1; e_f1.kind = PDT_EPI_BIAS_GELU; e_f1.bias = &w[7]; e_f1.aux = &u;
  Tensor gact = gemm_nt(x1, w[6], e_f1);

  Epi e_f2; e_f2.kind = PDT_EPI_BIAS_DROP_RES; e_f2.bias = &w[9]; e_f2.res = &x1; e_f2.drop_p = p_hidden;
  e_f2.rng = make_rng(rng_state, micro, site + 2, sids, S, B, rowmap);
  Tensor z2 = gemm_nt(gact, w[8], e_f2);
  Tensor stats2 = at::empty({T, 2}, f32);
  Tensor x2 = ln_fwd(z2, w[10], w[11], stats2, eps);
  return {x2, qkv, ctx, lse, z1, stats1, x1, u, gact, z2, stats2};
}

Tensor layer_backward(const Tensor& dy_in, const Tensor& x, const std::vector<Tensor>& sv, const Tensor& mask,
                      const Tensor& kv_len, const std::vector<Tensor>& w, std::vector<Tensor> gr,
                      const Tensor& rng_state, int64_t B, int64_t S, int64_t A, double p_attn, double p_hidden,
                      int64_t micro, int64_t layer, bool accumulate, const OptTensor& sids, const OptTensor& cu,
                      const OptTensor& rowmap) {
  DeviceGuard guard(x.device());
  const int32_t* cup = cu.has_value() ? cu->data_ptr<int32_t>() : nullptr;
  TORCH_CHECK(sv.size() == 10 && w.size() == 12 && gr.size() == 12, "layer_backward: bad list sizes");
  const Tensor dy = dy_in.contiguous();
  chk16(dy, "dy");
  const Tensor &qkv = sv[0], &ctx = sv[1], &lse = sv[2], &z1 = sv[3], &stats1 = sv[4], &x1 = sv[5], &u = sv[6],
               &gact = sv[7], &z2 = sv[8], &stats2 = sv[9];
  const int T = (int)x.size(0), H = (int)x.size(1);
  const int64_t site = layer * 4;

  FinishQueue fq;  // the four column-sum finishers of the layer run as one launch at its end
  // LN2 -> dz2 (residual path) and ddrop2 (through the FFN2 output dropout); d gamma/beta/b2
  LnBwdOut l2 = ln_bwd(dy, z2, stats2, w[10], &gr[10], &gr[11], &gr[9], p_hidden, make_rng(rng_state, micro, site + 2, sids, S, B, rowmap),
                       accumulate, &fq);
  // FFN2 dgrad fused with GELU' and with the FFN1 bias gradient (column sums of dU) ; wgrad
  Epi e_dg; e_dg.kind = PDT_EPI_MUL_DGELU; e_dg.aux = const_cast<Tensor*>(&u);
  e_dg.colsum = &gr[7]; e_dg.colsum_accumulate = accumulate; e_dg.fq = &fq;
  WgradLane lane(g_wgrad_mode == 2 || (g_wgrad_mode == 1 && T <= kWgradLaneMaxTokens), g_wgrad_defer);
  lane.run([&] { gemm_tn_into(l2.ddrop, gact, gr[8], accumulate); }, {l2.ddrop, gact});
  Tensor du = gemm_nn(l2.ddrop, w[8], e_dg);
  // FFN1 dgrad + residual gradient ; wgrad
  Epi e_add; e_add.kind = PDT_EPI_ADD_RES; e_add.res = &l2.dx;
  lane.run([&] { gemm_tn_into(du, x1, gr[6], accumulate); }, {du, x1});
  Tensor dx1 = gemm_nn(du, w[6], e_add);
  // LN1
  LnBwdOut l1 = ln_bwd(dx1, z1, stats1, w[4], &gr[4], &gr[5], &gr[3], p_hidden, make_rng(rng_state, micro, site + 1, sids, S, B, rowmap),
                       accumulate, &fq);
  // attention output projection
  lane.run([&] { gemm_tn_into(l1.ddrop, ctx, gr[2], accumulate); }, {l1.ddrop, ctx});
  Tensor dctx = gemm_nn(l1.ddrop, w[2], Epi{});
  // attention core
  Tensor dqkv = at::empty_like(qkv);
  Tensor delta = at::empty_like(lse);
  // the short-sequence backward kernel also emits per-sample column sums of dqkv (the QKV bias gradient)
  const bool fused_db = pdt_attn_bwd_fuses_dbias((int)S) != 0;
  const int db_rows = pdt_attn_bwd_dbias_rows((int)B, (int)S);
  Tensor db_part = fused_db ? at::empty({(int64_t)db_rows, 3 * (int64_t)H}, lse.options()) : Tensor();
  check(pdt_attn_bwd(dt_code(x), qkv.data_ptr(), mask.data_ptr<int32_t>(), kv_len.data_ptr<int32_t>(), cup, ctx.data_ptr(),
                     dctx.data_ptr(), lse.data_ptr<float>(), delta.data_ptr<float>(), dqkv.data_ptr(),
                     fused_db ? db_part.data_ptr<float>() : nullptr, (int)B, (int)S, (int)A, keep_scale(p_attn),
                     thr16(p_attn), make_rng(rng_state, micro, site + 0, sids, S, B), cur_stream()),
        "attn_bwd");
  if (fused_db)
    fq.add(db_part, db_rows, 3 * H, 1, gr[1].data_ptr<float>(), nullptr, nullptr, accumulate);
  else
    colsum_into(dqkv, gr[1], accumulate);
  // QKV projection dgrad + residual gradient ; wgrad
  Epi e_add2; e_add2.kind = PDT_EPI_ADD_RES; e_add2.res = &l1.dx;
  lane.run([&] { gemm_tn_into(dqkv, x, gr[0], accumulate); }, {dqkv, x});
  Tensor dx = gemm_nn(dqkv, w[0], e_add2);
  fq.flush();
  // the layer's gradients are complete on the main stream from here (bucket all-reduce, next layer) - or, deferred,
  // once the caller has joined after the backward pass
  lane.finish();
  (void)T; (void)H;
  return dx;
}

// ---------------------------------------------------------------------------------------------------
// A run of encoder layers as ONE call (one autograd node per stack instead of one per layer): with short
// micro-batches the host, not the GPU, bounds a step, and the Python / autograd round trip per layer is a fifth
// of the host time of a layer.  `layer_ids`: dropout site base of every layer.  Forward returns, per layer, the 11
// tensors of layer_forward (output first); the input of layer l > 0 is the output of layer l-1.
// ---------------------------------------------------------------------------------------------------
std::vector<Tensor> stack_forward(const Tensor& x, const Tensor& mask, const Tensor& kv_len,
                                  const std::vector<std::vector<Tensor>>& w, const Tensor& rng_state, int64_t B, int64_t S,
                                  int64_t A, double p_attn, double p_hidden, double eps, int64_t micro,
                                  const std::vector<int64_t>& layer_ids, const OptTensor& sids, const OptTensor& cu,
                                  const OptTensor& rowmap) {
  TORCH_CHECK(!w.empty() && w.size() == layer_ids.size(), "stack_forward: one weight list and one id per layer");
  std::vector<Tensor> all;
  all.reserve(w.size() * 11);
  Tensor h = x;
  for (size_t l = 0; l < w.size(); ++l) {
    std::vector<Tensor> o = layer_forward(h, mask, kv_len, w[l], rng_state, B, S, A, p_attn, p_hidden, eps, micro,
                                          layer_ids[l], sids, cu, rowmap);
    h = o[0];
    for (auto& t : o) all.push_back(std::move(t));
  }
  return all;
}

// Backward of the run, last layer first.  `saved[l]` = the 11 tensors stack_forward returned for layer l;
// `accumulate[l]`: add to layer l's gradients instead of overwriting them.  After every layer its saved tensors
// are dropped and `layer_done(l)` is called with the GIL held (the Python side drops ITS references there and
// reports the layer's gradient segment as complete: bucket all-reduce, gradient-lane merge).
Tensor stack_backward(const Tensor& dy_in, const Tensor& x0, std::vector<std::vector<Tensor>> saved, const Tensor& mask,
                      const Tensor& kv_len, const std::vector<std::vector<Tensor>>& w,
                      const std::vector<std::vector<Tensor>>& gr, const Tensor& rng_state, int64_t B, int64_t S, int64_t A,
                      double p_attn, double p_hidden, int64_t micro, const std::vector<int64_t>& layer_ids,
                      const std::vector<int64_t>& accumulate, const OptTensor& sids, const OptTensor& cu,
                      const OptTensor& rowmap, py::handle layer_done) {
  const size_t L = w.size();
  TORCH_CHECK(L > 0 && saved.size() == L && gr.size() == L && layer_ids.size() == L && accumulate.size() == L,
              "stack_backward: per-layer lists of different lengths");
  Tensor dy = dy_in;
  for (size_t i = L; i-- > 0;) {
    TORCH_CHECK(saved[i].size() == 11, "stack_backward: 11 saved tensors per layer expected");
    const Tensor& xin = i == 0 ? x0 : saved[i - 1][0];
    std::vector<Tensor> sv(saved[i].begin() + 1, saved[i].end());
    dy = layer_backward(dy, xin, sv, mask, kv_len, w[i], gr[i], rng_state, B, S, A, p_attn, p_hidden, micro, layer_ids[i],
                        accumulate[i] != 0, sids, cu, rowmap);
    sv.clear();
    saved[i].clear();
    if (!layer_done.is_none()) {
      py::gil_scoped_acquire gil;
      layer_done((int64_t)i);
    }
  }
  return dy;
}

// ---------------------------------------------------------------------------------------------------
// Embeddings
// ---------------------------------------------------------------------------------------------------
std::vector<Tensor> embed_forward(const c10::optional<Tensor>& ids, const Tensor& pos_ids, const Tensor& type_ids,
                                  const c10::optional<Tensor>& inputs_embeds, const Tensor& word, const Tensor& pos,
                                  const Tensor& type, const Tensor& gamma, const Tensor& beta, const Tensor& rng_state,
                                  double eps, double p, int64_t micro, int64_t site, const OptTensor& sids, int64_t seq,
                                  const OptTensor& rowmap) {
  DeviceGuard guard(word.device());
  const int H = (int)word.size(1);
  const int tokens = (int)pos_ids.numel();
  TORCH_CHECK(rowmap.has_value() || (seq >= 1 && tokens % seq == 0),
              "embed_forward: tokens must be a multiple of the sequence length");
  const OptTensor no_ids;
  const OptTensor& rsids = rowmap.has_value() ? no_ids : sids;  // (packed rows: the table alone defines the dropout rows)
  const int64_t rbatch = rowmap.has_value() ? -1 : tokens / seq;
  Tensor pre = at::empty({tokens, H}, word.options());
  Tensor y = at::empty({tokens, H}, word.options());
  Tensor stats = at::empty({tokens, 2}, word.options().dtype(at::kFloat));
  const int64_t* idp = ids.has_value() ? ids->data_ptr<int64_t>() : nullptr;
  const void* iep = nullptr;
  if (inputs_embeds.has_value()) { chk16(*inputs_embeds, "inputs_embeds"); iep = inputs_embeds->data_ptr(); }
  check(pdt_embed_fwd(dt_code(word), idp, pos_ids.data_ptr<int64_t>(), type_ids.data_ptr<int64_t>(), iep,
                      word.data_ptr(), pos.data_ptr(), type.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                      pre.data_ptr(), y.data_ptr(), stats.data_ptr<float>(), tokens, H, (int)word.size(0), (int)pos.size(0),
                      (int)type.size(0), (float)eps, keep_scale(p),
                      thr16(p), make_rng(rng_state, micro, site, rsids, seq, rbatch, rowmap), cur_stream()), "embed_fwd");
  return {y, pre, stats};
}

// grads order: 0 word 1 pos 2 type 3 gamma 4 beta (fp32 arena views).  Returns the gradient wrt the summed
// embedding (needed only when the caller fed inputs_embeds and wants its gradient).
Tensor embed_backward(const Tensor& dy_in, const Tensor& pre, const Tensor& stats, const c10::optional<Tensor>& ids,
                      const Tensor& pos_ids, const Tensor& type_ids, const Tensor& gamma, std::vector<Tensor> gr,
                      const Tensor& rng_state, double p, int64_t micro, int64_t site, int64_t pad_id, bool accumulate,
                      const OptTensor& sids, int64_t seq, const OptTensor& rowmap) {
  DeviceGuard guard(pre.device());
  const OptTensor no_ids;
  const OptTensor& rsids = rowmap.has_value() ? no_ids : sids;
  TORCH_CHECK(gr.size() == 5, "embed_backward: 5 gradient tensors expected");
  const Tensor dy = dy_in.contiguous();
  const int tokens = (int)pre.size(0), H = (int)pre.size(1);
  Tensor dmask = dy;
  if (p > 0.0) {
    dmask = at::empty_like(dy);
    check(pdt_dropout_apply(dt_code(dy), dy.data_ptr(), dmask.data_ptr(), dy.numel(), H, keep_scale(p), thr16(p),
                            make_rng(rng_state, micro, site, rsids, seq, rowmap.has_value() ? -1 : tokens / seq, rowmap),
                            cur_stream()), "dropout_apply");
  }
  LnBwdOut l = ln_bwd(dmask, pre, stats, gamma, &gr[3], &gr[4], nullptr, 0.0, make_rng(rng_state, micro, site),
                      accumulate);
  if (!accumulate) {
    gr[0].zero_();
    gr[1].zero_();
  }
  // word / position tables: sort the indices once (stable: equal indices keep their token order), then the
  // segmented-sum kernels add every run of equal indices to its row - deterministic, no atomics
  auto scatter = [&](const Tensor& idx, Tensor& table, int64_t skip) {
    auto sorted_perm = at::sort(idx, /*stable=*/true, /*dim=*/0, /*descending=*/false);
    const Tensor& sorted = std::get<0>(sorted_perm);
    const Tensor& perm = std::get<1>(sorted_perm);
    Tensor scratch = at::empty({(int64_t)pdt_embed_seg_chunks(tokens) * 2 * H}, stats.options());
    check(pdt_embed_scatter_sorted(dt_code(pre), l.dx.data_ptr(), sorted.data_ptr<int64_t>(), perm.data_ptr<int64_t>(),
                                   table.data_ptr<float>(), scratch.data_ptr<float>(), tokens, H, (int)table.size(0), skip,
                                   cur_stream()), "embed_scatter_sorted");
  };
  if (ids.has_value()) scatter(*ids, gr[0], pad_id);
  scatter(pos_ids, gr[1], -1);
  const int R = (int)gr[2].size(0);
  Tensor partial = at::empty({256, R, H}, stats.options());
  int nblk = 0;
  check(pdt_embed_type_partial(dt_code(pre), l.dx.data_ptr(), type_ids.data_ptr<int64_t>(), partial.data_ptr<float>(),
                               &nblk, tokens, H, R, cur_stream()), "embed_type_partial");
  float* gt = gr[2].data_ptr<float>();
  check(pdt_colsum_finish(partial.data_ptr<float>(), nblk, H, R, gt, R > 1 ? gt + H : nullptr,
                          R > 2 ? gt + 2 * H : nullptr, R > 3 ? gt + 3 * H : nullptr, accumulate ? 1 : 0, cur_stream()),
        "colsum_finish");
  return l.dx;
}

// ---------------------------------------------------------------------------------------------------
// Pooler (dense + tanh on token 0) and classifier + cross-entropy
// ---------------------------------------------------------------------------------------------------
Tensor dense_tanh_forward(const Tensor& first_tok, const Tensor& W, const Tensor& bias) {
  DeviceGuard guard(W.device());
  Epi e; e.kind = PDT_EPI_BIAS_TANH; e.bias = &bias;
  return gemm_nt(first_tok, W, e, 0);
}
// dfirst = dpre * W ; gW += dpre^T * first ; gb += colsum(dpre)     (dpre = grad wrt the pre-tanh value)
Tensor dense_backward(const Tensor& dpre, const Tensor& first_tok, const Tensor& W, Tensor gW, Tensor gb, bool accumulate) {
  DeviceGuard guard(W.device());
  Tensor dfirst = gemm_nn(dpre, W, Epi{}, 0);
  gemm_tn_into(dpre, first_tok, gW, accumulate, 0);
  colsum_into(dpre, gb, accumulate);
  return dfirst;
}

std::vector<Tensor> head_forward(const Tensor& pooled, const Tensor& W, const Tensor& bias,
                                 const c10::optional<Tensor>& labels, const Tensor& rng_state, double p, int64_t micro,
                                 int64_t site, const OptTensor& sids) {
  DeviceGuard guard(pooled.device());
  chk16(pooled, "pooled"); chkf32(W, "W"); chkf32(bias, "bias");
  const int B = (int)pooled.size(0), H = (int)pooled.size(1), C = (int)W.size(0);
  TORCH_CHECK(pooled.is_contiguous(), "pooled must be contiguous");
  if (labels.has_value())
    TORCH_CHECK(labels->is_contiguous() && labels->scalar_type() == at::kLong && labels->numel() == B,
                "labels must be a contiguous int64 [B] tensor");
  auto f32 = W.options();
  Tensor logits = at::empty({B, C}, f32);
  Tensor loss = at::zeros({}, f32);
  check(pdt_head_fwd(dt_code(pooled), pooled.data_ptr(), W.data_ptr<float>(), bias.data_ptr<float>(),
                     labels.has_value() ? labels->data_ptr<int64_t>() : nullptr, logits.data_ptr<float>(),
                     loss.data_ptr<float>(), B, H, C, keep_scale(p), thr16(p), make_rng(rng_state, micro, site, sids, 1, B),
                     cur_stream()), "head_fwd");
  return {logits, loss};
}

Tensor head_backward(const c10::optional<Tensor>& gloss, const c10::optional<Tensor>& dlogits, const Tensor& pooled,
                     const Tensor& W, const Tensor& logits, const c10::optional<Tensor>& labels, Tensor gW, Tensor gb,
                     const Tensor& rng_state, double p, int64_t micro, int64_t site, bool accumulate, bool tanh_bwd,
                     const OptTensor& sids) {
  DeviceGuard guard(pooled.device());
  const int B = (int)pooled.size(0), H = (int)pooled.size(1), C = (int)W.size(0);
  Tensor gl, dl;
  const bool ce = gloss.has_value() && labels.has_value();
  if (ce) gl = gloss->to(at::kFloat).contiguous();
  if (dlogits.has_value()) dl = dlogits->to(at::kFloat).contiguous();
  Tensor dpooled = at::empty_like(pooled);
  const int64_t ws_n = pdt_head_bwd_ws_floats(B, H, C);
  Tensor ws = ws_n > 0 ? at::empty({ws_n}, gW.options()) : Tensor();
  check(pdt_head_bwd(dt_code(pooled), pooled.data_ptr(), W.data_ptr<float>(), logits.data_ptr<float>(),
                     ce ? labels->data_ptr<int64_t>() : nullptr, ce ? gl.data_ptr<float>() : nullptr,
                     dl.defined() ? dl.data_ptr<float>() : nullptr, dpooled.data_ptr(), gW.data_ptr<float>(),
                     gb.data_ptr<float>(), B, H, C, keep_scale(p), thr16(p), make_rng(rng_state, micro, site, sids, 1, B),
                     accumulate ? 1 : 0, tanh_bwd ? 1 : 0, ws_n > 0 ? ws.data_ptr<float>() : nullptr, cur_stream()), "head_bwd");
  return dpooled;
}

// x: [B, H] (one row per sample)
Tensor dropout16(const Tensor& x, const Tensor& rng_state, double p, int64_t micro, int64_t site, const OptTensor& sids) {
  DeviceGuard guard(x.device());
  Tensor xc = x.contiguous();
  Tensor y = at::empty_like(xc);
  TORCH_CHECK(xc.dim() == 2, "dropout16 expects a [B, H] tensor");
  check(pdt_dropout_apply(dt_code(xc), xc.data_ptr(), y.data_ptr(), xc.numel(), (int)xc.size(1), keep_scale(p), thr16(p),
                          make_rng(rng_state, micro, site, sids, 1, xc.size(0)), cur_stream()), "dropout_apply");
  return y;
}

// ---------------------------------------------------------------------------------------------------
// Optimizer
// ---------------------------------------------------------------------------------------------------
void adamw_step(Tensor p, Tensor m, Tensor v, const Tensor& g, const c10::optional<Tensor>& p16, const Tensor& lr,
                const Tensor& step, Tensor scale_state, double beta1, double beta2, double eps, double weight_decay,
                bool correct_bias, bool check_finite) {
  DeviceGuard guard(p.device());
  chkf32(p, "p"); chkf32(m, "m"); chkf32(v, "v"); chkf32(g, "g"); chkf32(lr, "lr"); chkf32(scale_state, "scale_state");
  TORCH_CHECK(step.scalar_type() == at::kInt && step.is_cuda(), "step must be an int32 GPU scalar");
  TORCH_CHECK(scale_state.numel() >= 4, "scale_state: float[4]");
  if (check_finite) check(pdt_grad_check(g.data_ptr<float>(), g.numel(), scale_state.data_ptr<float>(), cur_stream()), "grad_check");
  PdtAdamW a{};
  a.p = p.data_ptr<float>(); a.m = m.data_ptr<float>(); a.v = v.data_ptr<float>(); a.g = g.data_ptr<float>();
  int dtype = PDT_F16;
  if (p16.has_value()) { chk16(*p16, "p16"); a.p16 = p16->data_ptr(); dtype = dt_code(*p16); }
  a.n = p.numel();
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
  a.correct_bias = correct_bias ? 1 : 0;
  a.lr = lr.data_ptr<float>(); a.step = step.data_ptr<int32_t>(); a.scale_state = scale_state.data_ptr<float>();
  check(pdt_adamw_step(dtype, &a, cur_stream()), "adamw_step");
}
void grad_check(const Tensor& g, Tensor scale_state) {
  DeviceGuard guard(g.device());
  chkf32(g, "g");
  check(pdt_grad_check(g.data_ptr<float>(), g.numel(), scale_state.data_ptr<float>(), cur_stream()), "grad_check");
}
void scaler_update(Tensor scale_state, Tensor step, double growth, double backoff, int64_t interval, bool dynamic) {
  DeviceGuard guard(scale_state.device());
  check(pdt_scaler_update(scale_state.data_ptr<float>(), step.data_ptr<int32_t>(), (float)growth, (float)backoff,
                          (int)interval, dynamic ? 1 : 0, cur_stream()), "scaler_update");
}
void cast_to_16(const Tensor& src, Tensor dst) {
  DeviceGuard guard(src.device());
  chkf32(src, "src"); chk16(dst, "dst");
  TORCH_CHECK(src.numel() == dst.numel(), "cast_to_16: size mismatch");
  check(pdt_cast_f32_to_16(dt_code(dst), src.data_ptr<float>(), dst.data_ptr(), src.numel(), cur_stream()), "cast");
}

// ---------------------------------------------------------------------------------------------------
// Single-op entry points (unit tests, micro-benchmarks)
// ---------------------------------------------------------------------------------------------------
Tensor op_gemm(const std::string& layout, const Tensor& A, const Tensor& B, int64_t epi,
               const c10::optional<Tensor>& bias, const c10::optional<Tensor>& res, c10::optional<Tensor> aux,
               const c10::optional<Tensor>& rng_state, double drop_p, int64_t micro, int64_t site, int64_t tile,
               c10::optional<Tensor> colsum, const OptTensor& sids, int64_t seq) {
  DeviceGuard guard(A.device());
  Epi e; e.kind = (int)epi; e.drop_p = drop_p;
  if (colsum.has_value()) e.colsum = &*colsum;  // adds to the tensor's content
  if (bias.has_value()) e.bias = &*bias;
  if (res.has_value()) e.res = &*res;
  if (aux.has_value()) e.aux = &*aux;
  if (rng_state.has_value()) e.rng = make_rng(*rng_state, micro, site, sids, seq, sids.has_value() ? A.size(0) / seq : -1);
  if (layout == "nt") return gemm_nt(A, B, e, (int)tile);
  if (layout == "nn") return gemm_nn(A, B, e, (int)tile);
  TORCH_CHECK(false, "op_gemm: layout must be nt or nn");
}
Tensor op_gemm_tn(const Tensor& A, const Tensor& B, c10::optional<Tensor> into, bool accumulate, int64_t tile, int64_t splits) {
  DeviceGuard guard(A.device());
  Tensor G = into.has_value() ? *into : at::zeros({A.size(1), B.size(1)}, A.options().dtype(at::kFloat));
  gemm_tn_into(A, B, G, accumulate, (int)tile, (int)splits);
  return G;
}
std::vector<Tensor> op_layernorm_fwd(const Tensor& x, const Tensor& gamma, const Tensor& beta, double eps) {
  DeviceGuard guard(x.device());
  chk16(x, "x");
  Tensor stats = at::empty({x.size(0), 2}, x.options().dtype(at::kFloat));
  Tensor y = ln_fwd(x, gamma, beta, stats, eps);
  return {y, stats};
}
std::vector<Tensor> op_layernorm_bwd(const Tensor& dy, const Tensor& x, const Tensor& stats, const Tensor& gamma,
                                     const c10::optional<Tensor>& rng_state, double drop_p, int64_t micro, int64_t site,
                                     const OptTensor& sids, int64_t seq) {
  DeviceGuard guard(x.device());
  const int H = (int)x.size(1);
  auto f32 = stats.options();
  Tensor gg = at::zeros({H}, f32), gb = at::zeros({H}, f32), gbias = at::zeros({H}, f32);
  PdtRng rng{nullptr, 0, 0};
  if (rng_state.has_value()) rng = make_rng(*rng_state, micro, site, sids, seq, sids.has_value() ? x.size(0) / seq : -1);
  LnBwdOut o = ln_bwd(dy.contiguous(), x, stats, gamma, &gg, &gb, &gbias, drop_p, rng, false);
  return {o.dx, o.ddrop, gg, gb, gbias};
}
std::vector<Tensor> op_attn_fwd(const Tensor& qkv, const Tensor& mask, const Tensor& kv_len, int64_t B, int64_t S, int64_t A,
                                const c10::optional<Tensor>& rng_state, double p, int64_t micro, int64_t site,
                                const OptTensor& sids, const OptTensor& cu) {
  DeviceGuard guard(qkv.device());
  chk16(qkv, "qkv");
  TORCH_CHECK(qkv.is_contiguous() && (cu.has_value() || qkv.size(0) == B * S) && qkv.size(1) == 3 * A * 64,
              "qkv must be [B*S, 3*A*64]");
  Tensor ctx = at::empty({qkv.size(0), A * 64}, qkv.options());
  Tensor lse = at::empty({B, A, S}, qkv.options().dtype(at::kFloat));
  PdtRng rng{nullptr, 0, 0};
  if (rng_state.has_value()) rng = make_rng(*rng_state, micro, site, sids, S, B);
  check(pdt_attn_fwd(dt_code(qkv), qkv.data_ptr(), mask.data_ptr<int32_t>(), kv_len.data_ptr<int32_t>(),
                     cu.has_value() ? cu->data_ptr<int32_t>() : nullptr, ctx.data_ptr(),
                     lse.data_ptr<float>(), (int)B, (int)S, (int)A, keep_scale(p), thr16(p), rng, cur_stream()), "attn_fwd");
  return {ctx, lse};
}
Tensor op_attn_bwd(const Tensor& qkv, const Tensor& mask, const Tensor& kv_len, const Tensor& ctx, const Tensor& dctx,
                   const Tensor& lse, int64_t B, int64_t S, int64_t A, const c10::optional<Tensor>& rng_state, double p,
                   int64_t micro, int64_t site, c10::optional<Tensor> dbias, const OptTensor& sids, const OptTensor& cu) {
  DeviceGuard guard(qkv.device());
  Tensor dqkv = at::empty_like(qkv);
  Tensor delta = at::empty_like(lse);
  Tensor dc = dctx.contiguous();
  PdtRng rng{nullptr, 0, 0};
  if (rng_state.has_value()) rng = make_rng(*rng_state, micro, site, sids, S, B);
  // `dbias` (fp32 [3H]): the column sums of dqkv are added to it (short-sequence kernel only)
  const int64_t H3 = qkv.size(1);
  Tensor part;
  const int db_rows = pdt_attn_bwd_dbias_rows((int)B, (int)S);
  if (dbias.has_value()) { chkf32(*dbias, "dbias"); part = at::empty({(int64_t)db_rows, H3}, lse.options()); }
  check(pdt_attn_bwd(dt_code(qkv), qkv.data_ptr(), mask.data_ptr<int32_t>(), kv_len.data_ptr<int32_t>(),
                     cu.has_value() ? cu->data_ptr<int32_t>() : nullptr, ctx.data_ptr(),
                     dc.data_ptr(), lse.data_ptr<float>(), delta.data_ptr<float>(), dqkv.data_ptr(),
                     dbias.has_value() ? part.data_ptr<float>() : nullptr, (int)B, (int)S, (int)A, keep_scale(p),
                     thr16(p), rng, cur_stream()), "attn_bwd");
  if (dbias.has_value())
    check(pdt_colsum_finish(part.data_ptr<float>(), db_rows, (int)H3, 1, dbias->data_ptr<float>(), nullptr, nullptr, nullptr, 1,
                            cur_stream()), "colsum_finish");
  return dqkv;
}

}  // namespace

void bind_rccl_comm(py::module_& m);  // rccl_comm.cpp
void bind_reducer(py::module_& m);    // reducer.cpp
void bind_wordpiece(py::module_& m);  // wordpiece.cpp

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "gfx950 kernels + native runtime of pytorch_distributed_training_amd";
  // (the trailing `sample_ids` arguments: optional int32 [B] global sample ids for sample-keyed dropout, see PdtRng)
  // The GIL is released while a call issues its launches (25-30 HIP calls for a layer's backward): a second Python
  // thread can enqueue the next micro-batch's forward on its own stream meanwhile (Engine.accumulate) - short
  // micro-batches are bound by the rate at which ONE host thread can launch kernels.
  using nogil = py::call_guard<py::gil_scoped_release>;
  m.def("layer_forward", &layer_forward, nogil());
  m.def("layer_backward", &layer_backward, nogil());
  m.def("stack_forward", &stack_forward, nogil());
  m.def("stack_backward", &stack_backward, nogil());
  m.def("embed_forward", &embed_forward, nogil());
  m.def("embed_backward", &embed_backward, nogil());
  m.def("dense_tanh_forward", &dense_tanh_forward, nogil());
  m.def("dense_backward", &dense_backward, nogil());
  m.def("head_forward", &head_forward, nogil());
  m.def("head_backward", &head_backward, nogil());
  m.def("dropout16", &dropout16);
  m.def("adamw_step", &adamw_step);
  m.def("grad_check", &grad_check);
  m.def("scaler_update", &scaler_update);
  m.def("cast_to_16", &cast_to_16);
  m.def("op_gemm", &op_gemm, py::arg("layout"), py::arg("A"), py::arg("B"), py::arg("epi"), py::arg("bias"),
        py::arg("res"), py::arg("aux"), py::arg("rng_state"), py::arg("drop_p"), py::arg("micro"), py::arg("site"),
        py::arg("tile"), py::arg("colsum") = py::none(), py::arg("sample_ids") = py::none(), py::arg("seq") = 1);
  m.def("op_gemm_tn", &op_gemm_tn);
  m.def("gemm_splits", [](const std::string& layout, const Tensor& A, const Tensor& B) {
    PdtGemm g{};
    g.M = (int)A.size(0); g.K = (int)A.size(1); g.N = (int)(layout == "nt" ? B.size(0) : B.size(1));
    return pdt_gemm_splits(layout == "nt" ? PDT_GEMM_NT : PDT_GEMM_NN, -1, &g);
  }, "K slices the dispatcher uses for a forward (nt) / dgrad (nn) GEMM of these operands (1 = one kernel)");
  m.def("op_layernorm_fwd", &op_layernorm_fwd);
  m.def("op_layernorm_bwd", &op_layernorm_bwd, py::arg("dy"), py::arg("x"), py::arg("stats"), py::arg("gamma"),
        py::arg("rng_state"), py::arg("drop_p"), py::arg("micro"), py::arg("site"), py::arg("sample_ids") = py::none(),
        py::arg("seq") = 1);
  m.def("op_attn_fwd", &op_attn_fwd, py::arg("qkv"), py::arg("mask"), py::arg("kv_len"), py::arg("B"), py::arg("S"), py::arg("A"),
        py::arg("rng_state"), py::arg("p"), py::arg("micro"), py::arg("site"), py::arg("sample_ids") = py::none(),
        py::arg("cu_seqlens") = py::none());
  m.def("op_attn_bwd", &op_attn_bwd, py::arg("qkv"), py::arg("mask"), py::arg("kv_len"), py::arg("ctx"), py::arg("dctx"),
        py::arg("lse"), py::arg("B"), py::arg("S"), py::arg("A"), py::arg("rng_state"), py::arg("p"), py::arg("micro"),
        py::arg("site"), py::arg("dbias") = py::none(), py::arg("sample_ids") = py::none(), py::arg("cu_seqlens") = py::none());
  m.def("set_wgrad_defer", [](bool on) { g_wgrad_defer = on; },
        "queue a layer's weight-gradient GEMMs behind one fork and skip the per-layer join (call wgrad_join after backward)");
  m.def("wgrad_join", [] { WgradLane::join_deferred(); },
        "current stream waits for the deferred weight-gradient GEMMs issued from it; releases the tensors kept for them");
  m.def("set_wgrad_side_stream", [](int64_t mode) { g_wgrad_mode = (int)mode; },
        "weight-gradient GEMMs of layer_backward on a second HIP stream: 0 never, 1 up to 12288 tokens (default), 2 always");
  m.def("set_gemm_persistent", [](int64_t mode) { pdt_gemm_set_persist_mode((int)mode); },
        "persistent launches of the ping-pong GEMM: 0 never, 1 from 4 tiles per workgroup (default), 2 from 2");
  m.def("gemm_persistent_mode", [] { return pdt_gemm_persist_mode(); });
  m.def("set_gemm_split_k", [](bool on) { pdt_gemm_set_split_mode(on ? 1 : 0); },
        "K split of forward / dgrad GEMMs with few output tiles: on (default) for latency-bound lone micro-batches, off "
        "while two micro-batches overlap on two streams");
  m.def("gemm_split_k", [] { return pdt_gemm_split_mode() != 0; });
  m.def("set_debug_sync", [](bool on) { g_debug_sync = on; },
        "drain the device after every native launch and report a fault at the op that caused it");
  m.def("debug_sync_count", [] { return g_debug_sync_count; });
  m.def("attn_bwd_fuses_dbias", [](int64_t S) { return pdt_attn_bwd_fuses_dbias((int)S) != 0; });
  bind_rccl_comm(m);  // (before the reducer: its constructor signature names RcclComm)
  bind_reducer(m);
  bind_wordpiece(m);
}
