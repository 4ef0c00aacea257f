// Fused optimizer-side kernels: non-finite check, HF-style AdamW with loss-scale un-scaling and the
// refresh of the 16-bit compute weights, dynamic loss-scale state machine.  All state (learning rate,
// step count, loss scale, overflow flag) lives in device memory, so a training step needs no host sync.
#include "common.hpp"

namespace pdt {

__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, int64_t n4, int64_t n,
                                                         float* __restrict__ scale_state) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4v v = reinterpret_cast<const float4v*>(g)[i];
    // x - x is 0 for finite x and NaN for +-inf / NaN
    const float t = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
    bad |= (t != 0.0f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) bad |= ((g[i] - g[i]) != 0.0f);
  if (bad) scale_state[2] = 1.0f;  // every writer stores the same value
}

template <typename T, bool HAS16>
__global__ __launch_bounds__(256) void adamw_kernel(PdtAdamW a) {
  if (a.scale_state[2] != 0.0f) return;  // overflow: the whole step is skipped
  const float inv_scale = 1.0f / a.scale_state[0];
  const float lr = a.lr[0];
  const float t = (float)(a.step[0] + 1);
  float step_size = lr;
  if (a.correct_bias) step_size *= sqrtf(1.0f - powf(a.beta2, t)) / (1.0f - powf(a.beta1, t));
  const float decay = lr * a.weight_decay;
  const int64_t n4 = a.n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4v p = reinterpret_cast<float4v*>(a.p)[i];
    float4v m = reinterpret_cast<float4v*>(a.m)[i];
    float4v v = reinterpret_cast<float4v*>(a.v)[i];
    const float4v g = reinterpret_cast<const float4v*>(a.g)[i] * inv_scale;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = a.beta1 * m[j] + (1.0f - a.beta1) * g[j];
      v[j] = a.beta2 * v[j] + (1.0f - a.beta2) * g[j] * g[j];
      p[j] -= step_size * m[j] / (sqrtf(v[j]) + a.eps);
      p[j] -= decay * p[j];
    }
    reinterpret_cast<float4v*>(a.p)[i] = p;
    reinterpret_cast<float4v*>(a.m)[i] = m;
    reinterpret_cast<float4v*>(a.v)[i] = v;
    if (HAS16) {
      short4v h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = Half16<T>::from_f(p[j]);
      reinterpret_cast<short4v*>(a.p16)[i] = h;
    }
  }
}

__global__ void scaler_update_kernel(float* scale_state, int32_t* step, float growth, float backoff, int interval,
                                     int dynamic) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const bool overflow = scale_state[2] != 0.0f;
  if (overflow) {
    if (dynamic) { scale_state[0] *= backoff; scale_state[1] = 0.0f; }
  } else {
    step[0] += 1;
    if (dynamic) {
      const float tr = scale_state[1] + 1.0f;
      if (tr >= (float)interval) { scale_state[0] *= growth; scale_state[1] = 0.0f; }
      else scale_state[1] = tr;
    }
  }
  scale_state[3] = overflow ? 1.0f : 0.0f;  // outcome of the step just finished (for logging)
  scale_state[2] = 0.0f;
}

template <typename T>
__global__ __launch_bounds__(256) void cast16_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    reinterpret_cast<short*>(dst)[i] = Half16<T>::from_f(src[i]);
}

__global__ __launch_bounds__(256) void fill_kernel(float* dst, float v, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

static inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace pdt

using namespace pdt;

extern "C" int pdt_grad_check(const float* g, int64_t n, float* scale_state, void* stream) {
  if (((uintptr_t)g) & 15) return -3;
  hipLaunchKernelGGL(grad_check_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, g, n >> 2, n,
                     scale_state);
  return (int)hipGetLastError();
}

extern "C" int pdt_adamw_step(int dtype, const PdtAdamW* a, void* stream) {
  if (a->n & 3) return -3;  // arenas are padded to 64 elements
  const dim3 grid(grid_for(a->n >> 2)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (!a->p16) hipLaunchKernelGGL((adamw_kernel<_Float16, false>), grid, block, 0, s, *a);
  else if (dtype == PDT_F16) hipLaunchKernelGGL((adamw_kernel<_Float16, true>), grid, block, 0, s, *a);
  else hipLaunchKernelGGL((adamw_kernel<__bf16, true>), grid, block, 0, s, *a);
  return (int)hipGetLastError();
}

extern "C" int pdt_scaler_update(float* scale_state, int32_t* step, float growth, float backoff, int interval,
                                 int dynamic, void* stream) {
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale_state, step, growth,
                     backoff, interval, dynamic);
  return (int)hipGetLastError();
}

extern "C" int pdt_cast_f32_to_16(int dtype, const float* src, void* dst, int64_t n, void* stream) {
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(cast16_kernel<_Float16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src,
                       (_Float16*)dst, n);
  else
    hipLaunchKernelGGL(cast16_kernel<__bf16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
  return (int)hipGetLastError();
}

extern "C" int pdt_fill_f32(float* dst, float value, int64_t n, void* stream) {
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, value, n);
  return (int)hipGetLastError();
}
