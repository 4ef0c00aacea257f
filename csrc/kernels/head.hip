// Classification head: dropout -> linear(H -> C) -> softmax cross-entropy (mean over the batch), forward
// and backward, fp32 math on a 16-bit pooled input.  The problem is tiny (B x H x C), so one kernel per
// direction; sums are taken in a fixed order, so the loss is bitwise reproducible.
#include "common.hpp"

namespace pdt {

constexpr int HEAD_MAX_C = 8;

constexpr int HEAD_FWD_WAVES = 16;  // one workgroup (fixed-order, deterministic loss sum); one sample per wave and pass
template <typename T>
__global__ __launch_bounds__(HEAD_FWD_WAVES * 64) void head_fwd_kernel(const T* __restrict__ pooled, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const int64_t* __restrict__ labels,
                                                       float* __restrict__ logits, float* __restrict__ loss, int B, int H,
                                                       int C, float drop_scale, uint32_t drop_thr, RngArgs rng) {
  __shared__ float wloss[HEAD_FWD_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  DropKey dk = drop_key(rng);
  float my_loss = 0.f;
  for (int b = wave; b < B; b += HEAD_FWD_WAVES) {
    float dot[HEAD_MAX_C];
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c) dot[c] = 0.f;
    for (int ch = lane; ch < (H >> 3); ch += 64) {
      float x[8];
      unpack8<T>(*reinterpret_cast<const short8*>(pooled + (size_t)b * H + ch * 8), x);
      if (drop_thr != 0) {
        const uint32_t keep = drop_keep8(dk, ((uint64_t)drop_sample(rng, b) * H + ch * 8) >> 3, drop_thr);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = ((keep >> j) & 1u) ? x[j] * drop_scale : 0.f;
      }
#pragma unroll
      for (int c = 0; c < HEAD_MAX_C; ++c)
        if (c < C)
#pragma unroll
          for (int j = 0; j < 8; ++j) dot[c] += x[j] * W[(size_t)c * H + ch * 8 + j];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c)
      if (c < C) { dot[c] = wave_sum(dot[c]) + bias[c]; mx = fmaxf(mx, dot[c]); }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c)
      if (c < C) se += __expf(dot[c] - mx);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < HEAD_MAX_C; ++c)
        if (c < C) logits[(size_t)b * C + c] = dot[c];
      if (labels) {
        const int y = (int)labels[b];
        float ly = 0.f;
#pragma unroll
        for (int c = 0; c < HEAD_MAX_C; ++c)
          if (c == y) ly = dot[c];
        my_loss += (mx + __logf(se)) - ly;
      }
    }
  }
  if (lane == 0) wloss[wave] = my_loss;
  __syncthreads();
  if (threadIdx.x == 0 && loss) {
    float t = 0.f;
    for (int w = 0; w < HEAD_FWD_WAVES; ++w) t += wloss[w];
    loss[0] = t / B;
  }
}

// grid = ceil(H/256) blocks; every block recomputes the B x C logit gradients into LDS (cheap) and then
// owns 256 columns h: dW[c][h], dpooled[b][h].  Block 0 also writes dbias.
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ pooled, const float* __restrict__ W,
                                                       const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       const float* __restrict__ gloss,
                                                       const float* __restrict__ dlogits_in, T* __restrict__ dpooled,
                                                       float* __restrict__ dW, float* __restrict__ dbias, int B, int H, int C,
                                                       float drop_scale, uint32_t drop_thr, RngArgs rng, int accumulate,
                                                       int tanh_bwd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dl = reinterpret_cast<float*>(smem);  // [B][C]
  // gradient wrt the logits = built-in mean-CE term (when labels and gloss are given) + an explicit
  // upstream gradient (when the caller computed its own loss from the logits)
  const bool ce = labels != nullptr && gloss != nullptr;
  const float gl = ce ? gloss[0] / B : 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float mx = -INFINITY, se = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)b * C + c]);
    for (int c = 0; c < C; ++c) se += __expf(logits[(size_t)b * C + c] - mx);
    const int y = ce ? (int)labels[b] : -1;
    for (int c = 0; c < C; ++c) {
      float d = ce ? (__expf(logits[(size_t)b * C + c] - mx) / se - (c == y ? 1.f : 0.f)) * gl : 0.f;
      if (dlogits_in) d += dlogits_in[(size_t)b * C + c];
      dl[b * C + c] = d;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < C) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dl[b * C + threadIdx.x];
    dbias[threadIdx.x] = accumulate ? dbias[threadIdx.x] + s : s;
  }
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  DropKey dk = drop_key(rng);
  float w[HEAD_MAX_C], gw[HEAD_MAX_C];
#pragma unroll
  for (int c = 0; c < HEAD_MAX_C; ++c) { w[c] = c < C ? W[(size_t)c * H + h] : 0.f; gw[c] = 0.f; }
  for (int b = 0; b < B; ++b) {
    const float x = Half16<T>::to_f(reinterpret_cast<const short*>(pooled)[(size_t)b * H + h]);
    float keepscale = 1.f;
    if (drop_thr != 0) {
      const uint32_t keep = drop_keep8(dk, ((uint64_t)drop_sample(rng, b) * H + h) >> 3, drop_thr);
      keepscale = ((keep >> (h & 7)) & 1u) ? drop_scale : 0.f;
    }
    float dp = 0.f;
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c)
      if (c < C) { const float d = dl[b * C + c]; gw[c] += d * x * keepscale; dp += d * w[c]; }
    dp *= keepscale;
    if (tanh_bwd) dp *= (1.f - x * x);
    reinterpret_cast<short*>(dpooled)[(size_t)b * H + h] = Half16<T>::from_f(dp);
  }
#pragma unroll
  for (int c = 0; c < HEAD_MAX_C; ++c)
    if (c < C) dW[(size_t)c * H + h] = accumulate ? dW[(size_t)c * H + h] + gw[c] : gw[c];
}

}  // namespace pdt

using namespace pdt;

extern "C" int pdt_head_fwd(int dtype, const void* pooled, const float* W, const float* bias, const int64_t* labels,
                            float* logits, float* loss, int B, int H, int C, float drop_scale, uint32_t drop_thr,
                            PdtRng rng, void* stream) {
  if ((H & 7) || C < 1 || C > HEAD_MAX_C || B <= 0) return -3;
#define PDT_GO(TT)                                                                                              \
  hipLaunchKernelGGL(head_fwd_kernel<TT>, dim3(1), dim3(HEAD_FWD_WAVES * 64), 0, (hipStream_t)stream, (const TT*)pooled, W, bias, \
                     labels, logits, loss, B, H, C, drop_scale, drop_thr, rng)
  if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
  return (int)hipGetLastError();
}

extern "C" int pdt_head_bwd(int dtype, const void* pooled, const float* W, const float* logits, const int64_t* labels,
                            const float* gloss, const float* dlogits_in, void* dpooled, float* dW, float* dbias, int B,
                            int H, int C,
                            float drop_scale, uint32_t drop_thr, PdtRng rng, int accumulate, int tanh_bwd, void* stream) {
  if (C < 1 || C > HEAD_MAX_C || B <= 0 || (size_t)B * C * 4 > 60000) return -3;
  const size_t lds = (size_t)B * C * sizeof(float);
#define PDT_GO(TT)                                                                                                  \
  hipLaunchKernelGGL(head_bwd_kernel<TT>, dim3((H + 255) / 256), dim3(256), lds, (hipStream_t)stream,               \
                     (const TT*)pooled, W, logits, labels, gloss, dlogits_in, (TT*)dpooled, dW, dbias, B, H, C, drop_scale,     \
                     drop_thr, rng, accumulate, tanh_bwd)
  if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
  return (int)hipGetLastError();
}
