// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of this framework.
// Everything here is written for 64-wide wavefronts and the gfx950 MFMA / LDS instruction set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "api.h"

namespace pdt {

typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4r __attribute__((__vector_size__(4 * sizeof(short))));

#define PDT_LDS_PTR(p) ((void __attribute__((address_space(3)))*)(p))
typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));

// ---- LDS-DMA (buffer_load_dwordx4 ... lds: 16 B per lane from global memory straight into LDS) -------------
// Issued through inline asm on purpose.  With the compiler builtin, the waitcnt-insertion pass knows an LDS-DMA
// is in flight and, having no alias information for it, puts `s_waitcnt vmcnt(0)` in front of every later LDS
// access it treats as a possible store - which includes every ds_read_b64_tr_b16 - so each K-tile of a kernel
// with transposed fragment reads drained its whole prefetch queue.  All kernels here order DMA data with their
// own counted `s_waitcnt vmcnt(N)` + barrier, so the compiler does not need to (and must not) know.
// Untracked VMEM operations only make the compiler's own vmcnt waits for ordinary loads more conservative
// (the counter is in order).
// Buffer resource: {base[31:0], base[47:32], num_records (bytes), flags}; offsets >= num_records read as zeros.
__device__ __forceinline__ int4v pdt_make_rsrc(const void* base, uint32_t bytes) {
  // (readfirstlane: the descriptor must live in SGPRs whatever the divergence analysis thinks of `base`)
  const uint64_t a = (uint64_t)base;
  return int4v{__builtin_amdgcn_readfirstlane((int)(uint32_t)a),
               __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xFFFFu)),
               __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
// lds_dst: wave-uniform LDS address of the 1-KiB piece (lane l lands at lds_dst + 16*l); voff: per-lane byte offset.
// M0 (the LDS base of the transfer) is written inside the asm; it is a reserved register the compiler only sets
// right in front of its own M0 users (LDS-DMA builtins, movrel indexing), none of which these kernels contain.
__device__ __forceinline__ void pdt_lds_dma16(int4v rsrc, const void* lds_dst, uint32_t voff) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)PDT_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc));
}

// The same with a scalar byte offset added to every lane's (the K-tile's position: the per-lane offsets of a tile
// window then never change and no vector instruction is spent per transfer).  The scalar offset may not be covered by
// the descriptor's range check: only pass offsets that keep in-range lanes inside the window.
__device__ __forceinline__ void pdt_lds_dma16_s(int4v rsrc, const void* lds_dst, uint32_t voff, uint32_t soff) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)PDT_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}

// One dword per lane through the same path (buffer_load_dword ... lds: lane l lands at lds_dst + 4*l): used to pull the
// cache lines the lanes point at into the XCD's L2 ahead of the real transfer; the 256 B it writes are junk.
__device__ __forceinline__ void pdt_lds_dma4(int4v rsrc, const void* lds_dst, uint32_t voff) {
  const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)PDT_LDS_PTR(lds_dst));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc));
}

// dtype codes shared with the host side
enum : int { DT_F16 = 0, DT_BF16 = 1 };

// ---------------------------------------------------------------------------------------
// 16-bit float traits: raw 16-bit lanes travel as `short`, math is always fp32.
// ---------------------------------------------------------------------------------------
template <typename T> struct Half16;
template <> struct Half16<_Float16> {
  static constexpr int code = DT_F16;
  static __device__ __forceinline__ float to_f(short s) { return (float)__builtin_bit_cast(_Float16, s); }
  static __device__ __forceinline__ short from_f(float f) { return __builtin_bit_cast(short, (_Float16)f); }
  static __device__ __forceinline__ float4v mfma(short8 a, short8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
  }
};
template <> struct Half16<__bf16> {
  static constexpr int code = DT_BF16;
  static __device__ __forceinline__ float to_f(short s) {
    return __builtin_bit_cast(float, ((uint32_t)(uint16_t)s) << 16);
  }
  static __device__ __forceinline__ short from_f(float f) { return __builtin_bit_cast(short, (__bf16)f); }
  static __device__ __forceinline__ float4v mfma(short8 a, short8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ void unpack8(short8 v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = Half16<T>::to_f(v[i]);
}
template <typename T> __device__ __forceinline__ short8 pack8(const float (&f)[8]) {
  short8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = Half16<T>::from_f(f[i]);
  return v;
}

// ---------------------------------------------------------------------------------------
// wave64 reductions
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------
// Counter-based RNG (Philox4x32-10).  A dropout mask is a pure function of
// (seed, offset, stream, element index): the backward pass regenerates it instead of storing it,
// and it does not depend on how a kernel maps elements to lanes.
// ---------------------------------------------------------------------------------------
using RngArgs = PdtRng;  // {device state {seed, step offset}, micro-batch index, dropout-site id}

__device__ __forceinline__ uint4 philox4x32_10(uint2 key, uint4 ctr) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}

struct DropKey {
  uint2 key;
  uint32_t stream, micro;
};
__device__ __forceinline__ DropKey drop_key(const RngArgs& r) {
  // a null state means "no dropout at this call site": never dereference it
  uint64_t seed = 0, off = 0;
  if (r.state) { seed = r.state[0]; off = r.state[1]; }
  uint64_t mixed = seed ^ (off * 0x9E3779B97F4A7C15ull);
  DropKey k;
  k.key = make_uint2((uint32_t)mixed, (uint32_t)(mixed >> 32));
  k.stream = r.stream;
  k.micro = r.micro;
  return k;
}
// Row of a [B*seq, H] activation -> the row its dropout counters are derived from (see PdtRng::sample_ids):
// global sample id * 512 + position inside the sample (512 = the longest sequence the position table allows).
__device__ __forceinline__ uint32_t drop_row(const RngArgs& r, uint32_t row) {
  if (r.row_table) return r.row_table[row];
  if (!r.sample_ids) return row;
  const uint32_t b = __umulhi(row, r.seq_magic);
  return (uint32_t)r.sample_ids[b] * 512u + (row - b * r.seq);
}
// p[idx] for a wave-uniform idx of a table no kernel of the step writes (sample ids): read through the constant
// address space, i.e. as a scalar load (s_load_dword, lgkmcnt).  As a plain global load the compiler cannot prove
// that the kernel's own stores leave the table alone and emits a vector load, whose s_waitcnt vmcnt(0) then also
// waits for every output store in flight - in the GEMM epilogue four such round trips in a row per tile.
__device__ __forceinline__ int32_t uniform_load_i32(const int32_t* p, uint32_t idx) {
  return ((const __attribute__((address_space(4))) int32_t*)p)[idx];
}
__device__ __forceinline__ uint32_t drop_sample(const RngArgs& r, uint32_t b) {
  return r.sample_ids ? (uint32_t)r.sample_ids[b] : b;
}
// keep-bits for the 8 elements of group `g` (element index / 8): bit i set => element kept.
// thr16 = round(p * 65536): an element is dropped when its 16-bit draw is < thr16.
__device__ __forceinline__ uint32_t drop_keep8(const DropKey& k, uint64_t g, uint32_t thr16) {
  uint4 r = philox4x32_10(k.key, make_uint4((uint32_t)g, (uint32_t)(g >> 32), k.stream, k.micro));
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
  uint32_t bits = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bits |= ((w[i] & 0xFFFFu) >= thr16 ? 1u : 0u) << (2 * i);
    bits |= ((w[i] >> 16) >= thr16 ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}

// Exact-erf GELU evaluated with the Abramowitz-Stegun 7.1.26 rational form of erf (|error| < 1.5e-7, far
// below 16-bit output rounding): one v_exp_f32 + one v_rcp_f32 + a 5-term Horner chain.  With x = u/sqrt(2) the
// exponential exp(-x^2) = exp(-u^2/2) is also the Gaussian of the derivative, so gelu and gelu' share it.
// The GEMM epilogues call this once per output element; the libm erff costs ~3x more VALU work there.
// Two elements at a time on <2 x float>: the multiply/add chain becomes packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32, two lanes' worth of math per issue); exp and rcp stay scalar.
__device__ __forceinline__ void gelu_parts2(float2v u, float2v& cdf, float2v& pdf) {
  const float2v x = float2v{fabsf(u.x), fabsf(u.y)} * 0.70710678118654752f;
  const float2v nx2 = -(x * x) * 1.4426950408889634f;
  const float2v e = float2v{__builtin_amdgcn_exp2f(nx2.x), __builtin_amdgcn_exp2f(nx2.y)};
  const float2v d = x * 0.3275911f + 1.0f;
  const float2v t = float2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  float2v p = t * 1.061405429f + -1.453152027f;
  p = t * p + 1.421413741f;
  p = t * p + -0.284496736f;
  p = t * p + 0.254829592f;
  const float2v h = (p * t) * (e * 0.5f);  // 0.5 * erfc(|x|)
  cdf = float2v{u.x >= 0.f ? 1.0f - h.x : h.x, u.y >= 0.f ? 1.0f - h.y : h.y};
  pdf = e * 0.3989422804014327f;
}
// v[i] = gelu(v[i]) and d[i] = gelu'(v[i]) on 8 values (the unit of the GEMM epilogues): the forward epilogue has
// cdf and pdf in registers anyway, so it emits the derivative for the backward pass at the cost of one packed fma
__device__ __forceinline__ void gelu8(float (&v)[8], float (&d)[8]) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    float2v c, p;
    const float2v u = float2v{v[i], v[i + 1]};
    gelu_parts2(u, c, p);
    const float2v r = u * c;
    const float2v g = u * p + c;
    v[i] = r.x; v[i + 1] = r.y;
    d[i] = g.x; d[i + 1] = g.y;
  }
}
// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (and its
// L2), so give each XCD a contiguous run of tiles.  Pure speed lever, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// Linear tile index -> (tile row, tile column), in groups of PDT_GROUP_M tile rows walked column by column.  The
// 32 workgroups an XCD runs at a time then cover a ~4 x 8 block of tiles instead of 2 x 16: fewer distinct
// operand panels stream through that XCD's 4 MiB L2 per K step (12 instead of 18 at 16 tile columns).
#ifndef PDT_GROUP_M
#define PDT_GROUP_M 4
#endif
// (GM = tile rows per group: 4 for the 128x128 kernels; the 256x256 ping-pong kernel walks groups of 8 - an XCD's 32
// tiles are then 8 rows x 4 columns, and the next 32 the same 8 rows (the same 8 activation panels) x the next 4
// columns: +0.5-0.9 % per step at 131072 tokens (seq 128 and 512) against groups of 4; groups of 16 measured -1.2 %,
// groups of 2 -0.3 %; the 8-sample micro-batches on the 128x128 kernels lose 0.7 % with 8)
template <int GM = PDT_GROUP_M>
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  if (GM <= 1) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
  const int per_group = GM * tiles_n;
  const int grp = t / per_group, r = t - grp * per_group;
  const int first = grp * GM;
  const int gsz = min(GM, tiles_m - first);
  tn = r / gsz;
  tm = first + (r - tn * gsz);
}

}  // namespace pdt
