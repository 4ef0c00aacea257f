// Classification head: dropout -> linear(H -> C) -> softmax cross-entropy (mean over the batch), forward
// and backward, fp32 math on a 16-bit pooled input.  The problem is tiny (B x H x C), so one kernel per
// direction; sums are taken in a fixed order, so the loss is bitwise reproducible.
#include "common.hpp"

namespace pdt {

constexpr int HEAD_MAX_C = 8;

constexpr int HEAD_SLICE = 64;  // samples per workgroup of the sliced kernels (batches of more than HEAD_SLICE samples)

// Larger batches: the logits of HEAD_SLICE samples per workgroup (one sample per wave and pass), then ONE workgroup sums
// the per-sample losses from the logits in a fixed order (head_loss_kernel) - the single-workgroup kernel below took
// 0.38 ms for 1024 samples.
template <typename T>
__global__ __launch_bounds__(256) void head_logits_kernel(const T* __restrict__ pooled, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits, int B,
                                                          int H, int C, float drop_scale, uint32_t drop_thr, RngArgs rng) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  DropKey dk = drop_key(rng);
  const int b1 = min(B, ((int)blockIdx.x + 1) * HEAD_SLICE);
  for (int b = blockIdx.x * HEAD_SLICE + wave; b < b1; b += 4) {
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
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c)
      if (c < C) {
        const float v = wave_sum(dot[c]) + bias[c];
        if (lane == 0) logits[(size_t)b * C + c] = v;
      }
  }
}

// loss[0] = mean over the batch of CE(logits[b], labels[b]): thread t takes the samples t, t + 1024, ...; the partial
// sums are combined per wave and then across the 16 waves in a fixed order
__global__ __launch_bounds__(1024) void head_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         float* __restrict__ loss, int B, int C) {
  __shared__ float wloss[16];
  float my = 0.f;
  for (int b = threadIdx.x; b < B; b += 1024) {
    float mx = -INFINITY, se = 0.f, ly = 0.f;
    const int y = (int)labels[b];
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)b * C + c]);
    for (int c = 0; c < C; ++c) {
      const float v = logits[(size_t)b * C + c];
      se += __expf(v - mx);
      if (c == y) ly = v;
    }
    my += (mx + __logf(se)) - ly;
  }
  my = wave_sum(my);
  if ((threadIdx.x & 63) == 0) wloss[threadIdx.x >> 6] = my;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += wloss[w];
    loss[0] = t / B;
  }
}

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

// grid = (ceil(H/256), slices): a block recomputes the logit gradients of its batch slice into LDS (cheap) and then
// owns 256 columns h of that slice: dpooled[b][h], and the slice's share of dW[c][h] (blockIdx.x == 0: of dbias).  One
// slice (gridDim.y == 1): the shares ARE the gradients, written (or added) in place.  More slices: the shares go to
// part[slice][C*H] and partb[slice][C] and a finishing launch adds them up in slice order.
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ pooled, const float* __restrict__ W,
                                                       const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                       const float* __restrict__ gloss,
                                                       const float* __restrict__ dlogits_in, T* __restrict__ dpooled,
                                                       float* __restrict__ dW, float* __restrict__ dbias, int B, int H, int C,
                                                       float drop_scale, uint32_t drop_thr, RngArgs rng, int accumulate,
                                                       int tanh_bwd, int slice, float* __restrict__ part,
                                                       float* __restrict__ partb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dl = reinterpret_cast<float*>(smem);  // [slice][C]
  const int b0 = blockIdx.y * slice, b1 = min(B, b0 + slice);
  const bool sliced = gridDim.y > 1;
  // gradient wrt the logits = built-in mean-CE term (when labels and gloss are given) + an explicit
  // upstream gradient (when the caller computed its own loss from the logits)
  const bool ce = labels != nullptr && gloss != nullptr;
  const float gl = ce ? gloss[0] / B : 0.f;
  for (int b = b0 + threadIdx.x; b < b1; b += 256) {
    float mx = -INFINITY, se = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)b * C + c]);
    for (int c = 0; c < C; ++c) se += __expf(logits[(size_t)b * C + c] - mx);
    const int y = ce ? (int)labels[b] : -1;
    for (int c = 0; c < C; ++c) {
      float d = ce ? (__expf(logits[(size_t)b * C + c] - mx) / se - (c == y ? 1.f : 0.f)) * gl : 0.f;
      if (dlogits_in) d += dlogits_in[(size_t)b * C + c];
      dl[(b - b0) * C + c] = d;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && (int)threadIdx.x < C) {
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += dl[(b - b0) * C + threadIdx.x];
    if (sliced) partb[blockIdx.y * C + threadIdx.x] = s;
    else dbias[threadIdx.x] = accumulate ? dbias[threadIdx.x] + s : s;
  }
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  DropKey dk = drop_key(rng);
  float w[HEAD_MAX_C], gw[HEAD_MAX_C];
#pragma unroll
  for (int c = 0; c < HEAD_MAX_C; ++c) { w[c] = c < C ? W[(size_t)c * H + h] : 0.f; gw[c] = 0.f; }
  for (int b = b0; b < b1; ++b) {
    const float x = Half16<T>::to_f(reinterpret_cast<const short*>(pooled)[(size_t)b * H + h]);
    float keepscale = 1.f;
    if (drop_thr != 0) {
      const uint32_t keep = drop_keep8(dk, ((uint64_t)drop_sample(rng, b) * H + h) >> 3, drop_thr);
      keepscale = ((keep >> (h & 7)) & 1u) ? drop_scale : 0.f;
    }
    float dp = 0.f;
#pragma unroll
    for (int c = 0; c < HEAD_MAX_C; ++c)
      if (c < C) { const float d = dl[(b - b0) * C + c]; gw[c] += d * x * keepscale; dp += d * w[c]; }
    dp *= keepscale;
    if (tanh_bwd) dp *= (1.f - x * x);
    reinterpret_cast<short*>(dpooled)[(size_t)b * H + h] = Half16<T>::from_f(dp);
  }
#pragma unroll
  for (int c = 0; c < HEAD_MAX_C; ++c)
    if (c < C) {
      if (sliced) part[((size_t)blockIdx.y * C + c) * H + h] = gw[c];
      else dW[(size_t)c * H + h] = accumulate ? dW[(size_t)c * H + h] + gw[c] : gw[c];
    }
}

}  // namespace pdt

using namespace pdt;

extern "C" int pdt_head_fwd(int dtype, const void* pooled, const float* W, const float* bias, const int64_t* labels,
                            float* logits, float* loss, int B, int H, int C, float drop_scale, uint32_t drop_thr,
                            PdtRng rng, void* stream) {
  if ((H & 7) || C < 1 || C > HEAD_MAX_C || B <= 0) return -3;
  if (B > HEAD_SLICE) {  // logits in parallel over the samples, then the loss from the logits
    const int nb = (B + HEAD_SLICE - 1) / HEAD_SLICE;
#define PDT_GO(TT)                                                                                              \
  hipLaunchKernelGGL(head_logits_kernel<TT>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const TT*)pooled, W, bias, logits, \
                     B, H, C, drop_scale, drop_thr, rng)
    if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
    if (labels && loss) hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, loss, B, C);
    return (int)hipGetLastError();
  }
#define PDT_GO(TT)                                                                                              \
  hipLaunchKernelGGL(head_fwd_kernel<TT>, dim3(1), dim3(HEAD_FWD_WAVES * 64), 0, (hipStream_t)stream, (const TT*)pooled, W, bias, \
                     labels, logits, loss, B, H, C, drop_scale, drop_thr, rng)
  if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
  return (int)hipGetLastError();
}

// floats of workspace pdt_head_bwd needs for this batch (0: small batch, one slice)
extern "C" int64_t pdt_head_bwd_ws_floats(int B, int H, int C) {
  if (B <= HEAD_SLICE) return 0;
  const int64_t nb = (B + HEAD_SLICE - 1) / HEAD_SLICE;
  return nb * ((int64_t)C * H + C);
}

extern "C" int pdt_head_bwd(int dtype, const void* pooled, const float* W, const float* logits, const int64_t* labels,
                            const float* gloss, const float* dlogits_in, void* dpooled, float* dW, float* dbias, int B,
                            int H, int C,
                            float drop_scale, uint32_t drop_thr, PdtRng rng, int accumulate, int tanh_bwd, float* ws,
                            void* stream) {
  if (C < 1 || C > HEAD_MAX_C || B <= 0) return -3;
  const bool sliced = B > HEAD_SLICE;
  if (sliced && !ws) return -3;
  const int slice = sliced ? HEAD_SLICE : B, nb = (B + slice - 1) / slice;
  const size_t lds = (size_t)slice * C * sizeof(float);
  float* part = ws;
  float* partb = ws ? ws + (size_t)nb * C * H : nullptr;
#define PDT_GO(TT)                                                                                                  \
  hipLaunchKernelGGL(head_bwd_kernel<TT>, dim3((H + 255) / 256, nb), dim3(256), lds, (hipStream_t)stream,           \
                     (const TT*)pooled, W, logits, labels, gloss, dlogits_in, (TT*)dpooled, dW, dbias, B, H, C, drop_scale,     \
                     drop_thr, rng, accumulate, tanh_bwd, slice, part, partb)
  if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
  int e = (int)hipGetLastError();
  if (e || !sliced) return e;
  PdtColsumJob jobs[2] = {{part, {dW, nullptr, nullptr, nullptr}, nb, C * H, 1, accumulate},
                          {partb, {dbias, nullptr, nullptr, nullptr}, nb, C, 1, accumulate}};
  return pdt_colsum_finish_multi(jobs, 2, stream);
}
