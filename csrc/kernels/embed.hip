// Embedding kernels: fused gather(word,pos,type) + add + LayerNorm + dropout forward, and the
// gradient scatter into the fp32 tables.  One wave per token, 16 B per lane.
#include "common.hpp"

namespace pdt {

constexpr int EMB_MAX_CHUNKS = 4;  // H <= 2048

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                        const int64_t* __restrict__ type_ids, const T* __restrict__ inputs_embeds,
                                                        const T* __restrict__ word, const T* __restrict__ pos,
                                                        const T* __restrict__ type, const T* __restrict__ gamma,
                                                        const T* __restrict__ beta, T* __restrict__ pre, T* __restrict__ y,
                                                        float* __restrict__ stats, int tokens, int H, int V, int P,
                                                        int R, float eps, float drop_scale, uint32_t drop_thr,
                                                        RngArgs rng) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= tokens) return;
  const int nchunk = H >> 3;
  // indices come from user data: clamp them into their tables (a bad id must not become a wild address)
  auto clampi = [](int64_t v, int n) { return (size_t)(v < 0 ? 0 : (v >= n ? n - 1 : v)); };
  const T* wrow = inputs_embeds ? inputs_embeds + (size_t)tok * H : word + clampi(ids[tok], V) * H;
  const T* prow = pos + clampi(pos_ids[tok], P) * H;
  const T* trow = type + clampi(type_ids[tok], R) * H;
  float v[EMB_MAX_CHUNKS][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < EMB_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float a[8], b[8], d[8];
      unpack8<T>(*reinterpret_cast<const short8*>(wrow + c * 8), a);
      unpack8<T>(*reinterpret_cast<const short8*>(prow + c * 8), b);
      unpack8<T>(*reinterpret_cast<const short8*>(trow + c * 8), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = a[j] + b[j] + d[j];
      // the backward pass normalises the stored 16-bit copy, so the statistics use the same values
      const short8 packed = pack8<T>(v[i]);
      *reinterpret_cast<short8*>(pre + (size_t)tok * H + c * 8) = packed;
      unpack8<T>(packed, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mean = wave_sum(sum) / H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < EMB_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / H + eps);
  if (lane == 0) { stats[2 * tok] = mean; stats[2 * tok + 1] = rstd; }
  DropKey dk = drop_key(rng);
#pragma unroll
  for (int i = 0; i < EMB_MAX_CHUNKS; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      float g[8], b[8], o[8];
      unpack8<T>(*reinterpret_cast<const short8*>(gamma + c * 8), g);
      unpack8<T>(*reinterpret_cast<const short8*>(beta + c * 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
      if (drop_thr != 0) {
        const uint32_t keep = drop_keep8(dk, ((uint64_t)drop_row(rng, tok) * H + c * 8) >> 3, drop_thr);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = ((keep >> j) & 1u) ? o[j] * drop_scale : 0.f;
      }
      *reinterpret_cast<short8*>(y + (size_t)tok * H + c * 8) = pack8<T>(o);
    }
  }
}

// Word / position table gradients: a deterministic segmented sum, no atomics.
// The host sorts the row indices (stable), so equal indices form runs in the original token order; `perm` maps a
// sorted position back to its token.  Stage 1 gives every ALIGNED chunk of SEG_CH sorted positions to one wave (per
// block of 512 columns): it walks the chunk in order, sums the rows of each run and, at the end of a run, either adds
// the sum to the table row (the run lies inside the chunk: this wave is the row's only writer) or parks it in
// scratch[chunk][slot] (the run crosses a chunk border: slot 0 when it comes in from the previous chunk, else slot 1
// when it goes on into the next one).  Stage 2 finishes the runs that span chunks: the wave of the chunk where such a
// run STARTS adds up its parked sums in chunk order and writes the row.  A hot row ([CLS], [SEP], every position
// row: >= 1024 summands at the benchmark batch) is thereby summed by many waves in parallel, yet always in the same
// order, so the gradient is bit-reproducible.  Indices equal to `skip_id` (nn.Embedding's padding_idx) or outside
// the table contribute nothing.
constexpr int SEG_CH = 32;

template <typename T>
__global__ __launch_bounds__(256) void embed_seg_stage1_kernel(const T* __restrict__ d, const int64_t* __restrict__ sorted,
                                                               const int64_t* __restrict__ perm, float* __restrict__ table,
                                                               float* __restrict__ scratch, int n, int H, int rows,
                                                               int64_t skip_id) {
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = chunk * SEG_CH;
  if (p0 >= n) return;
  const int col = blockIdx.y * 512 + lane * 8;
  if (col >= H) return;
  const int p1 = min(n, p0 + SEG_CH);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t cur = sorted[p0];
  int first = p0;
  auto flush = [&](int last) {  // the run of `cur` occupies sorted positions [first, last] of this chunk
    if (cur != skip_id && cur >= 0 && cur < rows) {
      const bool head_open = first == p0 && p0 > 0 && sorted[p0 - 1] == cur;
      const bool tail_open = last == p1 - 1 && p1 < n && sorted[p1] == cur;
      float* dst = (head_open || tail_open) ? scratch + ((size_t)chunk * 2 + (head_open ? 0 : 1)) * H + col
                                            : table + (size_t)cur * H + col;
      float4v a = {acc[0], acc[1], acc[2], acc[3]}, b = {acc[4], acc[5], acc[6], acc[7]};
      if (!(head_open || tail_open)) {
        a += *reinterpret_cast<const float4v*>(dst);
        b += *reinterpret_cast<const float4v*>(dst + 4);
      }
      *reinterpret_cast<float4v*>(dst) = a;
      *reinterpret_cast<float4v*>(dst + 4) = b;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  };
  for (int p = p0; p < p1; p += 8) {
    short8 v[8];
    int64_t id[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // 8 independent row loads in flight
      const int q = min(p + u, p1 - 1);
      id[u] = sorted[q];
      v[u] = *reinterpret_cast<const short8*>(d + (size_t)perm[q] * H + col);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p + u >= p1) break;
      if (id[u] != cur) {
        flush(p + u - 1);
        cur = id[u];
        first = p + u;
      }
      float f[8];
      unpack8<T>(v[u], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
  flush(p1 - 1);
}

__global__ __launch_bounds__(256) void embed_seg_stage2_kernel(const int64_t* __restrict__ sorted, float* __restrict__ table,
                                                               const float* __restrict__ scratch, int n, int H, int rows,
                                                               int64_t skip_id) {
  const int lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p0 = chunk * SEG_CH;
  if (p0 >= n) return;
  const int col = blockIdx.y * 512 + lane * 8;
  if (col >= H) return;
  const int p1 = min(n, p0 + SEG_CH);
  const int64_t id = sorted[p1 - 1];
  if (!(p1 < n && sorted[p1] == id)) return;                       // the last run of this chunk ends here
  if (sorted[p0] == id && p0 > 0 && sorted[p0 - 1] == id) return;   // ... or started in an earlier chunk
  if (id == skip_id || id < 0 || id >= rows) return;
  // this chunk holds the start of a run that goes on: its own tail sum, then the head sums of the chunks it crosses
  const float* src = scratch + ((size_t)chunk * 2 + 1) * H + col;
  float4v a = *reinterpret_cast<const float4v*>(src), b = *reinterpret_cast<const float4v*>(src + 4);
  for (int c = chunk + 1; c * SEG_CH < n; ++c) {
    src = scratch + (size_t)c * 2 * H + col;
    a += *reinterpret_cast<const float4v*>(src);
    b += *reinterpret_cast<const float4v*>(src + 4);
    const int e = min(n, (c + 1) * SEG_CH);
    if (!(sorted[e - 1] == id && e < n && sorted[e] == id)) break;  // the run ends inside chunk c
  }
  float* dst = table + (size_t)id * H + col;
  a += *reinterpret_cast<const float4v*>(dst);
  b += *reinterpret_cast<const float4v*>(dst + 4);
  *reinterpret_cast<float4v*>(dst) = a;
  *reinterpret_cast<float4v*>(dst + 4) = b;
}

// Token-type table (<= 4 rows): masked column sums, deterministic.  partial[nblk][R][H]
template <typename T>
__global__ __launch_bounds__(256) void embed_type_partial_kernel(const T* __restrict__ d, const int64_t* __restrict__ type_ids,
                                                                 float* __restrict__ partial, int tokens, int H, int R,
                                                                 int rows_per_blk) {
  const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (col >= H) return;
  const int r0 = blockIdx.y * rows_per_blk, r1 = min(tokens, r0 + rows_per_blk);
  float acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
  for (int t = r0; t < r1; ++t) {
    float v[8];
    unpack8<T>(*reinterpret_cast<const short8*>(d + (size_t)t * H + col), v);
    const int64_t tr = type_ids[t];
    const int ty = tr < 0 ? 0 : (tr >= R ? R - 1 : (int)tr);  // (clamped like the forward lookup)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r == ty)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] += v[j];
  }
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) partial[((size_t)blockIdx.y * R + r) * H + col + j] = acc[r][j];
}

}  // namespace pdt

using namespace pdt;

extern "C" int pdt_embed_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                             const void* inputs_embeds, const void* word, const void* pos, const void* type,
                             const void* gamma, const void* beta, void* pre, void* y, float* stats, int tokens, int H,
                             int V, int P, int R, float eps, float drop_scale, uint32_t drop_thr, PdtRng rng,
                             void* stream) {
  if ((H & 7) || H > EMB_MAX_CHUNKS * 512 || tokens <= 0 || V <= 0 || P <= 0 || R <= 0) return -3;
  if (!ids && !inputs_embeds) return -2;
  const dim3 grid((tokens + 3) / 4), block(256);
#define PDT_GO(TT)                                                                                                   \
  hipLaunchKernelGGL(embed_fwd_kernel<TT>, grid, block, 0, (hipStream_t)stream, ids, pos_ids, type_ids,               \
                     (const TT*)inputs_embeds, (const TT*)word, (const TT*)pos, (const TT*)type, (const TT*)gamma,    \
                     (const TT*)beta, (TT*)pre, (TT*)y, stats, tokens, H, V, P, R, eps, drop_scale, drop_thr, rng)
  if (dtype == PDT_F16) PDT_GO(_Float16); else PDT_GO(__bf16);
#undef PDT_GO
  return (int)hipGetLastError();
}

extern "C" int pdt_embed_seg_chunks(int n) { return (n + SEG_CH - 1) / SEG_CH; }

extern "C" int pdt_embed_scatter_sorted(int dtype, const void* d, const int64_t* sorted, const int64_t* perm, float* table,
                                        float* scratch, int n, int H, int rows, int64_t skip_id, void* stream) {
  if (n <= 0 || rows <= 0 || (H & 7) || !table || !scratch) return -3;
  const int chunks = pdt_embed_seg_chunks(n);
  const dim3 grid((chunks + 3) / 4, (H + 511) / 512), block(256);
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(embed_seg_stage1_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, (const _Float16*)d, sorted,
                       perm, table, scratch, n, H, rows, skip_id);
  else
    hipLaunchKernelGGL(embed_seg_stage1_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)d, sorted, perm,
                       table, scratch, n, H, rows, skip_id);
  hipLaunchKernelGGL(embed_seg_stage2_kernel, grid, block, 0, (hipStream_t)stream, sorted, table, (const float*)scratch, n, H,
                     rows, skip_id);
  return (int)hipGetLastError();
}

extern "C" int pdt_embed_type_partial(int dtype, const void* d, const int64_t* type_ids, float* partial, int* nblk,
                                      int tokens, int H, int R, void* stream) {
  if ((H & 7) || R < 1 || R > 4 || tokens <= 0) return -3;
  int rpb = 64;
  int nb = (tokens + rpb - 1) / rpb;
  if (nb > 256) { rpb = (tokens + 255) / 256; nb = (tokens + rpb - 1) / rpb; }
  *nblk = nb;
  const dim3 grid((H / 8 + 255) / 256, nb), block(256);
  if (dtype == PDT_F16)
    hipLaunchKernelGGL(embed_type_partial_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, (const _Float16*)d,
                       type_ids, partial, tokens, H, R, rpb);
  else
    hipLaunchKernelGGL(embed_type_partial_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)d,
                       type_ids, partial, tokens, H, R, rpb);
  return (int)hipGetLastError();
}
