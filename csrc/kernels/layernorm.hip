// LayerNorm forward / backward and column-sum helpers (bias / gamma / beta gradients) for gfx950.
// One 64-lane wave owns one row; every global access is 16 B per lane; statistics in fp32.
#include "common.hpp"

namespace pdt {

constexpr int LN_MAX_CHUNKS = 4;   // 8-element chunks per lane  => H <= 2048
constexpr int LN_BWD_MAX_BLOCKS = 768;

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = H >> 3;
  float v[LN_MAX_CHUNKS][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      unpack8<T>(*reinterpret_cast<const short8*>(x + (size_t)row * H + c * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mean = wave_sum(sum) / H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / H + eps);
  if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float g[8], b[8], o[8];
      unpack8<T>(*reinterpret_cast<const short8*>(gamma + c * 8), g);
      unpack8<T>(*reinterpret_cast<const short8*>(beta + c * 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      *reinterpret_cast<short8*>(y + (size_t)row * H + c * 8) = pack8<T>(o);
    }
  }
}

// Backward.  Each wave walks rows (grid-stride) and keeps the column partial sums of its lanes'
// columns in registers; the 4 waves of a block are combined through LDS and written as one
// partial row, reduced afterwards by colsum_finish_kernel (deterministic, no atomics).
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats, const T* __restrict__ gamma,
                                                     T* __restrict__ dx, T* __restrict__ ddrop,
                                                     float* __restrict__ partial, int rows, int H,
                                                     float drop_scale, uint32_t drop_thr, RngArgs rng) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = H >> 3;
  float pg[NCH][8], pb[NCH][8], pd[NCH][8], gam[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pg[i][j] = 0.f; pb[i][j] = 0.f; pd[i][j] = 0.f; gam[i][j] = 0.f; }
    if (c < nchunk) unpack8<T>(*reinterpret_cast<const short8*>(gamma + c * 8), gam[i]);
  }
  DropKey dk = drop_key(rng);
  const float invH = 1.0f / H;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    // (looked up here, next to the row loads, not where the mask is drawn: the sample-id load is a dependent access)
    const uint32_t drow = (ddrop && drop_thr != 0) ? drop_row(rng, (uint32_t)row) : (uint32_t)row;
    float g[NCH][8], xh[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float d[8], xv[8];
        unpack8<T>(*reinterpret_cast<const short8*>(dy + (size_t)row * H + c * 8), d);
        unpack8<T>(*reinterpret_cast<const short8*>(x + (size_t)row * H + c * 8), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xv[j] - mean) * rstd;
          g[i][j] = d[j] * gam[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          pg[i][j] += d[j] * xh[i][j];
          pb[i][j] += d[j];
        }
      }
    }
    const float c1 = wave_sum(s1) * invH, c2 = wave_sum(s2) * invH;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - c1 - xh[i][j] * c2);
        *reinterpret_cast<short8*>(dx + (size_t)row * H + c * 8) = pack8<T>(o);
        if (ddrop) {
          if (drop_thr != 0) {
            const uint32_t keep = drop_keep8(dk, ((uint64_t)drow * H + c * 8) >> 3, drop_thr);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ((keep >> j) & 1u) ? o[j] * drop_scale : 0.f;
          }
          const short8 packed = pack8<T>(o);
          *reinterpret_cast<short8*>(ddrop + (size_t)row * H + c * 8) = packed;
          // the bias gradient sums exactly what the following dgrad/wgrad GEMMs consume
          float r[8];
          unpack8<T>(packed, r);
#pragma unroll
          for (int j = 0; j < 8; ++j) pd[i][j] += r[j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) pd[i][j] += o[j];
        }
      }
    }
  }
  // combine the block's 4 waves: LDS [4][3][H + H/32] fp32.  A lane owns 8 consecutive columns per chunk, so its
  // stores are 8 floats apart; one pad word per 32 columns (column x at word x + x/32) spreads the 32 lanes of a store
  // over 32 banks instead of 4, and keeps the block at 49.5 KiB for H = 1024: three blocks per CU (a pad word per
  // chunk made it 54 KiB, two blocks per CU, and the kernel 48 % slower).
  float* red = reinterpret_cast<float*>(smem);
  const int HP = H + (H >> 5);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 3 + 0) * HP + c * 8 + (c >> 2) + j] = pg[i][j];
        red[(wave * 3 + 1) * HP + c * 8 + (c >> 2) + j] = pb[i][j];
        red[(wave * 3 + 2) * HP + c * 8 + (c >> 2) + j] = pd[i][j];
      }
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * H; idx += 256) {
    const int seg = idx / H, col = idx - seg * H;
    const int w = seg * HP + col + (col >> 5);
    const float s = red[w] + red[3 * HP + w] + red[6 * HP + w] + red[9 * HP + w];
    partial[(size_t)blockIdx.x * 3 * H + idx] = s;
  }
}

struct ColsumOuts { float* out[4]; };

// Each block owns 32 columns: 32 row groups x 32 columns; a thread sums every 32nd partial row of its column
// (coalesced 128-B reads across the 32 lanes, 4 independent accumulators to keep loads in flight), then the
// row groups are combined in a fixed order through LDS.
__device__ __forceinline__ void colsum_finish_body(const float* __restrict__ partial, int nblk, int H, int nseg,
                                                   float* const (&out)[4], int accumulate, int block) {
  __shared__ float red[32][33];
  const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int total = nseg * H;
  const int idx = block * 32 + cg;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < total) {
    int b = rg;
    for (; b + 96 < nblk; b += 128) {
      s0 += partial[(size_t)b * total + idx];
      s1 += partial[(size_t)(b + 32) * total + idx];
      s2 += partial[(size_t)(b + 64) * total + idx];
      s3 += partial[(size_t)(b + 96) * total + idx];
    }
    for (; b < nblk; b += 32) s0 += partial[(size_t)b * total + idx];
  }
  red[rg][cg] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && idx < total) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][cg];
    const int seg = idx / H, c = idx % H;
    float* o = seg == 0 ? out[0] : (seg == 1 ? out[1] : (seg == 2 ? out[2] : out[3]));
    if (o) o[c] = accumulate ? o[c] + t : t;
  }
}

__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, int nblk, int H,
                                                             int nseg, ColsumOuts outs, int accumulate) {
  colsum_finish_body(partial, nblk, H, nseg, outs.out, accumulate, blockIdx.x);
}

// several jobs in one launch: block -> job through the running block counts (first[j] = first block of job j)
struct ColsumJobs {
  PdtColsumJob job[PDT_COLSUM_MAX_JOBS];
  int first[PDT_COLSUM_MAX_JOBS + 1];
  int njobs;
};
__global__ __launch_bounds__(1024) void colsum_finish_multi_kernel(ColsumJobs js) {
  int j = 0;
  while (j + 1 < js.njobs && (int)blockIdx.x >= js.first[j + 1]) ++j;  // block-uniform
  const PdtColsumJob& q = js.job[j];
  colsum_finish_body(q.partial, q.nblk, q.H, q.nseg, q.out, q.accumulate, (int)blockIdx.x - js.first[j]);
}

// column partial sums of a 16-bit matrix: grid (ceil(ncols/2048), nblk); 64 rows per block row
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int rows, int ncols, int ld,
                                                             float* __restrict__ partial, int rows_per_blk) {
  const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (col >= ncols) return;
  const int r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) {
    float v[8];
    unpack8<T>(*reinterpret_cast<const short8*>(x + (size_t)r * ld + col), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) partial[(size_t)blockIdx.y * ncols + col + j] = acc[j];
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n8,
                                                            int H, float scale, uint32_t thr, RngArgs rng) {
  DropKey dk = drop_key(rng);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    float v[8];
    unpack8<T>(reinterpret_cast<const short8*>(x)[i], v);
    if (thr != 0) {
      // the counters of element (row, col) of the [rows, H] tensor, as every other kernel derives them
      const uint32_t row = (uint32_t)((i * 8) / H);
      const uint64_t grp = ((uint64_t)drop_row(rng, row) * H + (uint64_t)(i * 8 - (int64_t)row * H)) >> 3;
      const uint32_t keep = drop_keep8(dk, grp, thr);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = ((keep >> j) & 1u) ? v[j] * scale : 0.f;
    }
    reinterpret_cast<short8*>(y)[i] = pack8<T>(v);
  }
}

}  // namespace pdt

using namespace pdt;

extern "C" int pdt_layernorm_bwd_max_blocks(void) { return LN_BWD_MAX_BLOCKS; }

extern "C" int pdt_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* stats,
                                 int rows, int H, float eps, void* stream) {
  if ((H & 7) || H > LN_MAX_CHUNKS * 512 || rows <= 0) return -3;
  const dim3 grid((rows + 3) / 4), block(256);
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(ln_fwd_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, (const _Float16*)x,
                       (const _Float16*)gamma, (const _Float16*)beta, (_Float16*)y, stats, rows, H, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)x,
                       (const __bf16*)gamma, (const __bf16*)beta, (__bf16*)y, stats, rows, H, eps);
  return (int)hipGetLastError();
}

extern "C" int pdt_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const void* gamma,
                                 void* dx, void* ddrop, float* partial, int* nblk, int rows, int H, float drop_scale,
                                 uint32_t drop_thr, PdtRng rng, void* stream) {
  if ((H & 7) || H > LN_MAX_CHUNKS * 512 || rows <= 0) return -3;
  // rows per workgroup: 16 (4 per wave) keeps the partial-sum matrix the finisher reads small; short inputs
  // (<= 4096 rows) get one row per wave instead, so that a 1024-row micro-batch runs on 256 CUs, not on 64
  // (BERT-large, 12 x 8-sample micro-batches: 966 -> 1009 samples/s)
  const int rpw = rows <= 4096 ? 4 : 16;
  int nb = (rows + rpw - 1) / rpw;
  nb = nb < 1 ? 1 : (nb > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : nb);
  *nblk = nb;
  const size_t lds = (size_t)4 * 3 * (H + H / 32) * sizeof(float);
  // registers are sized for the row length: 1 / 2 / 4 chunks of 8 elements per lane (H <= 512 / 1024 / 2048)
#define PDT_LNB(TT, NCH)                                                                                          \
  do {                                                                                                            \
    if (lds > 65536) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)ln_bwd_kernel<TT, NCH>,                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      if (e != hipSuccess) return (int)e;                                                                         \
    }                                                                                                             \
    hipLaunchKernelGGL((ln_bwd_kernel<TT, NCH>), dim3(nb), dim3(256), lds, (hipStream_t)stream, (const TT*)dy,     \
                       (const TT*)x, stats, (const TT*)gamma, (TT*)dx, (TT*)ddrop, partial, rows, H, drop_scale,  \
                       drop_thr, rng);                                                                            \
  } while (0)
  const int nch = H <= 512 ? 1 : (H <= 1024 ? 2 : 4);
  if (dtype == PDT_F16) {
    if (nch == 1) PDT_LNB(_Float16, 1); else if (nch == 2) PDT_LNB(_Float16, 2); else PDT_LNB(_Float16, 4);
  } else {
    if (nch == 1) PDT_LNB(__bf16, 1); else if (nch == 2) PDT_LNB(__bf16, 2); else PDT_LNB(__bf16, 4);
  }
#undef PDT_LNB
  return (int)hipGetLastError();
}

extern "C" int pdt_colsum_finish(const float* partial, int nblk, int H, int nseg, float* o0, float* o1, float* o2,
                                 float* o3, int accumulate, void* stream) {
  if (nseg < 1 || nseg > 4) return -3;
  ColsumOuts outs{{o0, o1, o2, o3}};
  const int total = nseg * H;
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((total + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial, nblk,
                     H, nseg, outs, accumulate);
  return (int)hipGetLastError();
}

extern "C" int pdt_colsum_finish_multi(const PdtColsumJob* jobs, int njobs, void* stream) {
  if (njobs < 1 || njobs > PDT_COLSUM_MAX_JOBS) return -3;
  ColsumJobs js{};
  js.njobs = njobs;
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    if (jobs[j].nseg < 1 || jobs[j].nseg > 4 || jobs[j].H < 1 || jobs[j].nblk < 0 || !jobs[j].partial) return -3;
    js.job[j] = jobs[j];
    js.first[j] = blocks;
    blocks += (jobs[j].nseg * jobs[j].H + 31) / 32;
  }
  js.first[njobs] = blocks;
  hipLaunchKernelGGL(colsum_finish_multi_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, js);
  return (int)hipGetLastError();
}

extern "C" int pdt_colsum_partial(int dtype, const void* x, int rows, int ncols, int ld, float* partial, int* nblk,
                                  void* stream) {
  if ((ncols & 7) || (ld & 7) || rows <= 0) return -3;
  int rpb = 64;
  int nb = (rows + rpb - 1) / rpb;
  if (nb > 256) { rpb = (rows + 255) / 256; nb = (rows + rpb - 1) / rpb; }
  *nblk = nb;
  const dim3 grid((ncols / 8 + 255) / 256, nb);
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(colsum_partial_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                       rows, ncols, ld, partial, rpb);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, rows,
                       ncols, ld, partial, rpb);
  return (int)hipGetLastError();
}

extern "C" int pdt_dropout_apply(int dtype, const void* x, void* y, int64_t n, int H, float drop_scale,
                                 uint32_t drop_thr, PdtRng rng, void* stream) {
  if ((n & 7) || H <= 0 || (H & 7) || n % H) return -3;
  const int64_t n8 = n >> 3;
  int grid = (int)((n8 + 255) / 256);
  grid = grid > 2048 ? 2048 : (grid < 1 ? 1 : grid);
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(dropout_apply_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x, (_Float16*)y, n8, H, drop_scale, drop_thr, rng);
  else
    hipLaunchKernelGGL(dropout_apply_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       (__bf16*)y, n8, H, drop_scale, drop_thr, rng);
  return (int)hipGetLastError();
}
