// Fused scaled-dot-product attention for BERT-class encoders on gfx950: head_dim 64, S <= 512, key padding
// mask, dropout on the probabilities.  Forward streams the keys through LDS in 128-row chunks with an online
// softmax (one chunk = the whole sequence at S <= 128, where the rescale is a no-op).
// Backward recomputes the probabilities from the saved log-sum-exp in two sweeps with no atomics on dQ/dK/dV:
//   dQ sweep    : one wave = 16 query rows, sweeps the keys       (S^T, dP^T -> dS^T -> dQ^T)
//   dK/dV sweep : one wave = 16 keys,       sweeps the queries    (S, dP -> P, dS -> dV^T, dK^T)
// For S <= 128 both sweeps run in ONE kernel on shared LDS images (and the QKV bias gradient is reduced in
// it); longer sequences use one kernel per sweep.
// Products are arranged "key on the register axis" so that an accumulator tile is directly the B operand
// of the next MFMA (16x16x32, k index permuted consistently on both operands), and the other operand is
// fetched from a row-major LDS image with ds_read_b64_tr_b16.  One XOR-swizzled image layout
// (16-B chunk ^= row & 6) is conflict-free for both the ds_read_b128 row reads and the transposed reads.
#include <mutex>
#include <type_traits>
#include <unordered_set>

#include "common.hpp"

namespace pdt {

constexpr int HD = 64;       // head dim (the MFMA tiling of every kernel here is built on it; the model checks it)
constexpr float ATT_SCALE = 0.125f;  // 1 / sqrt(HD)
static_assert(ATT_SCALE * ATT_SCALE * HD == 1.0f, "softmax scale must be 1 / sqrt(head dim)");
constexpr float LOG2E = 1.4426950408889634f;
constexpr int ATT_WAVES = 8;  // waves per workgroup: 8 x 16 = 128 query rows (or keys) share one K/V (or Q/dO) image
constexpr int ATT_THREADS = ATT_WAVES * 64;
constexpr int ATT_ROWS = ATT_WAVES * 16;

struct AttnArgs {
  const void* qkv;   // [B*S, 3H]
  const int32_t* mask;
  const int32_t* kv_len;
  // Unpadded ("varlen") batches: cu[b] = first token row of sample b in the packed [T, ...] tensors, cu[B] = T; S is
  // then the LONGEST sequence (grid extent, stride of lse / delta) and every key of a sample is valid (mask and
  // kv_len are not read).  Null: padded [B, S] layout.
  const int32_t* cu;
  void* ctx;         // fwd out [B*S, H]
  float* lse;        // [B, A, S]
  const void* octx;  // bwd: forward output
  const void* dctx;  // bwd: grad of ctx
  float* delta;      // [B, A, S]
  void* dqkv;        // [B*S, 3H]
  int B, S, A, H;
  float scale;       // 1 / sqrt(head dim)
  float drop_scale;
  uint32_t drop_thr;  // round(p * 65536), 0 = off
  RngArgs rng;
};

// Dropout decision for probability (bh, q, key): a function of (seed, offset, site, micro, b*A+h, q, key) only,
// so every kernel agrees whatever its lane mapping is.  It is split in two so that the per-element cost is one
// 32-bit multiply: a strongly mixed 32-bit seed per query row (computed once per row - in the dK/dV sweeps it
// is staged in LDS), and per element  x = low24(row_seed ^ key * GOLD) * MIX24,  keep <=> x[31:16] >= thr.
// key * GOLD is a Weyl sequence over the keys; an unrolled sweep forms it with one add per element.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
constexpr uint32_t ATT_GOLD = 0x9E3779B9u, ATT_MIX = 0xEBCA6Bu;
__device__ __forceinline__ uint32_t attn_row_seed(const DropKey& dk, uint32_t bh, uint32_t q) {
  uint32_t x = lowbias32(q ^ dk.key.x);
  return lowbias32(x ^ (bh * 0x9E3779B9u) ^ dk.key.y ^ (dk.stream * 0x85EBCA6Bu) ^ (dk.micro * 0xC2B2AE35u));
}
// thr_hi = drop threshold << 16 (see attn_thr_hi); kmul = key * ATT_GOLD
__device__ __forceinline__ bool attn_keep(uint32_t row_seed, uint32_t kmul, uint32_t thr_hi) {
  return __umul24(row_seed ^ kmul, ATT_MIX) >= thr_hi;  // 24 x 24 -> low 32 bits: a full-rate multiply
}
__device__ __forceinline__ uint32_t attn_thr_hi(uint32_t thr) { return (thr > 65535u ? 65535u : thr) << 16; }

// LDS-DMA fill of one 128-row image of 64 16-bit values per row (2 x 1 KiB pieces per wave) from rows
// [0, 128) of `src` (row stride `stride` elements): the swizzle (16-B chunk ^= row & 6) is applied on the per-lane
// SOURCE address because a DMA write is lane-linear; rows at or beyond `nvalid` read out of bounds and arrive as
// zeros.  Completion: s_waitcnt vmcnt(0) + barrier by the caller.
template <typename T>
__device__ __forceinline__ void dma_image128(char* img, const T* src, int stride, int nvalid, int tid, int wave) {
  const uint32_t bytes = nvalid > 0 ? (uint32_t)(((size_t)(nvalid - 1) * stride + HD) * 2) : 0u;
  const int4v rsrc = pdt_make_rsrc(src, bytes);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = u * ATT_THREADS + tid, row = idx >> 3, c = (idx & 7) ^ (row & 6);
    const uint32_t off = row < nvalid ? ((uint32_t)row * stride + c * 8) * 2u : 0x80000000u;
    pdt_lds_dma16(rsrc, img + (u * ATT_THREADS + wave * 64) * 16, off);
  }
}
// rows x0..x0+15 as an MFMA A operand: lane -> row x0+(lane&15), k = 8*(lane>>4)+j of k-step ks (k = head dim)
__device__ __forceinline__ short8 row_frag(const char* img, int x0, int ks, int lane) {
  const int row = x0 + (lane & 15), c = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const short8*>(img + row * 128 + ((c ^ (row & 6)) << 4));
}
// transposed operand over a 32-row block starting at r0: lane -> column 16*db+(lane&15), k-values
// j<4: row r0+4g+j, j>=4: row r0+16+4g+(j-4)   (g = lane>>4)  -- the same k permutation as pack_pair()
__device__ __forceinline__ short8 tr_frag(const char* img, int r0, int db, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int row = r0 + 4 * g + q;
  const int c16 = 2 * db + (p >> 1);
  const char* a = img + row * 128 + ((c16 ^ (row & 6)) << 4) + (p & 1) * 8;
  short4r lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4r __attribute__((address_space(3)))*)PDT_LDS_PTR(a));
  short4r hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4r __attribute__((address_space(3)))*)PDT_LDS_PTR(a + 16 * 128));
  short8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// The per-score arithmetic of these kernels is written on whole float4v accumulator tiles (scale + mask, subtract the
// maximum, dropout factor, dS = P * (dP - delta)): element-wise operations on 4-vectors become v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32, two elements per instruction, and pairs of results are converted with one
// v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32.  Only exp2 and the dropout hash stay per element.  (This file is built without
// the SLP vectoriser, see build_ext.py: it would also pair up the adds of the DPP row reductions, which then cost a
// v_mov_b32_dpp each instead of folding into v_add_f32_dpp.)
typedef float float2v __attribute__((ext_vector_type(2)));
template <typename T> struct Pk2;
template <> struct Pk2<_Float16> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct Pk2<__bf16> { typedef __bf16 type __attribute__((ext_vector_type(2))); };
template <typename T>
__device__ __forceinline__ int cvt_pk(float lo, float hi) {  // two fp32 -> one dword of two 16-bit values (round to nearest even)
  return __builtin_bit_cast(int, __builtin_convertvector(float2v{lo, hi}, typename Pk2<T>::type));
}
__device__ __forceinline__ float4v splat4(float v) { return float4v{v, v, v, v}; }
// A 4-vector whose elements are consumed one by one (exp2 of each) would be taken apart by the optimiser BEFORE the
// operation that produced it ("extract of a vector op = scalar op of the extracts"), i.e. four v_sub_f32 instead of two
// v_pk_add_f32.  Passing the two halves through an empty asm statement as 64-bit operands keeps the producing operation
// packed; it emits nothing.
__device__ __forceinline__ float4v packed4(float4v v) {
  float2v lo = {v[0], v[1]}, hi = {v[2], v[3]};
  asm("" : "+v"(lo));
  asm("" : "+v"(hi));
  return float4v{lo[0], lo[1], hi[0], hi[1]};
}
template <typename T>
__device__ __forceinline__ short8 pack_pair(const float4v& a, const float4v& b) {
  const int4v r = {cvt_pk<T>(a[0], a[1]), cvt_pk<T>(a[2], a[3]), cvt_pk<T>(b[0], b[1]), cvt_pk<T>(b[2], b[3])};
  return __builtin_bit_cast(short8, r);
}
// lane's 8 k-values of row (lane&15) of a row-major global matrix, k-step ks: B-operand style fragment
template <typename T>
__device__ __forceinline__ short8 gfrag(const T* base, int row, int stride, int ks, int lane) {
  return *reinterpret_cast<const short8*>(base + (size_t)row * stride + ks * 32 + (lane >> 4) * 8);
}
__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Store a wave's 16 x 64 tile as 16-bit values: lane (lm, g) holds the fp32 values t[db][r] of row lm, columns
// 16 db + 4 g + r (the accumulator layout), `row` points at the 64 elements of its row.  Lanes g and g ^ 1 of a row swap
// half of their blocks first, so that each lane writes two 16-byte chunks and every store instruction covers 64
// contiguous bytes of a row; four 8-byte stores per lane (32-byte pieces) measured 7 % slower on the fused backward.
// Call with the whole wave (the exchange is a cross-lane operation); `valid` guards the lane's stores only.
template <typename T>
__device__ __forceinline__ void store_tile16x64(T* row, const float4v (&t)[4], float mul, int g, bool valid) {
  int2v pk[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const float4v v = t[db] * mul;
    pk[db] = int2v{cvt_pk<T>(v[0], v[1]), cvt_pk<T>(v[2], v[3])};
  }
  const bool odd = g & 1;  // even g keeps blocks 0 and 2 and receives its neighbour's; odd g keeps 1 and 3
  const int2v own0 = odd ? pk[1] : pk[0], own1 = odd ? pk[3] : pk[2];
  const int2v snd0 = odd ? pk[0] : pk[1], snd1 = odd ? pk[2] : pk[3];
  int2v rcv0, rcv1;
  rcv0[0] = __shfl_xor(snd0[0], 16, 64); rcv0[1] = __shfl_xor(snd0[1], 16, 64);
  rcv1[0] = __shfl_xor(snd1[0], 16, 64); rcv1[1] = __shfl_xor(snd1[1], 16, 64);
  const int4v c0 = odd ? int4v{rcv0[0], rcv0[1], own0[0], own0[1]} : int4v{own0[0], own0[1], rcv0[0], rcv0[1]};
  const int4v c1 = odd ? int4v{rcv1[0], rcv1[1], own1[0], own1[1]} : int4v{own1[0], own1[1], rcv1[0], rcv1[1]};
  if (valid) {
    T* dst = row + (odd ? 16 : 0) + 4 * (g & ~1);
    *reinterpret_cast<int4v*>(dst) = c0;
    *reinterpret_cast<int4v*>(dst + 32) = c1;
  }
}

// All three kernels below walk the key (or query) axis in chunks of 128 rows: two 16-KiB LDS images are re-filled
// per chunk, so the footprint is ~34-38 KiB for any S <= 512 and four workgroups (32 waves) share a CU - these
// kernels are VALU-bound (exp, dropout hash, scaling per score element against 4 MFMAs per 16x16 block at
// head dim 64), so what they need is waves to interleave, not a resident K/V panel.
constexpr int ATT_SMAX = 512;   // longest sequence
constexpr int ATT_CHUNK = 128;  // rows per LDS image
// Two image PAIRS (K | V, or Q | dO): the chunk loops are double-buffered - chunk c+1 is staged while chunk c is worked
// on.  (With one pair every chunk began with an exposed fill; at seq 512, four chunks per workgroup, the three kernels
// spent half their wave-cycles waiting.)
constexpr int ATT_IMG_BYTES = 4 * ATT_CHUNK * 128;
// Top of a chunk-loop iteration (`stage(c0, buf)` in scope; each wave issues 4 DMA pieces per chunk): every wave is done
// with the chunk before, so its pair may be refilled with the chunk after this one; then this chunk has landed for
// every wave (all but the 4 newest pieces of this wave, then the barrier).
#define PDT_ATT_NEXT_CHUNK(C0, BUF, END)                                              \
  do {                                                                                \
    if (C0) __syncthreads();                                                          \
    if ((C0) + ATT_CHUNK < (END)) {                                                   \
      stage((C0) + ATT_CHUNK, (BUF) ^ 1);                                             \
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                \
    } else {                                                                          \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
    }                                                                                 \
    __syncthreads();                                                                  \
  } while (0)

// ------------------------------------------------------------------------------------------------ forward
// One wave = 16 query rows.  Online softmax over the key chunks in the log2 domain: running maximum m, running
// sum l (per lane, reduced once at the end), output accumulators rescaled by exp2(m_old - m_new) per chunk.
// Dropout multiplies the un-normalised probabilities (the 1/l factor is applied to the output at the end).
// DBUF: two image pairs, chunk c+1 staged under chunk c (sequences of more than one chunk); without it one pair,
// filled at the top of every chunk (the single-chunk launch: nothing to stage ahead, and the simpler loop measured
// 10 % faster there)
// (experiments) minimum waves per SIMD the forward's register allocation must allow; 0: the compiler's choice (120
// registers, two workgroups per CU).  Measured at batch 1024: 6 (80 registers, three workgroups per CU, 44 spilled
// registers) takes 431 us per layer instead of 255 us - the spills go through scratch memory in a memory-bound kernel.
#ifndef PDT_ATT_FWD_WAVES_EU
#define PDT_ATT_FWD_WAVES_EU 0
#endif
#if PDT_ATT_FWD_WAVES_EU > 0
#define PDT_ATT_FWD_OCC __attribute__((amdgpu_waves_per_eu(PDT_ATT_FWD_WAVES_EU, 8)))
#else
#define PDT_ATT_FWD_OCC
#endif
template <typename T, bool DROP, bool DBUF>
__global__ __launch_bounds__(ATT_THREADS) PDT_ATT_FWD_OCC void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* mbias = reinterpret_cast<float*>(smem + (DBUF ? ATT_IMG_BYTES : ATT_IMG_BYTES / 2));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int SM = a.S, ld = 3 * a.H;                    // SM: longest sequence (lse / delta stride)
  const int row0 = a.cu ? a.cu[b] : b * SM;            // first token row of this sample
  const int S = a.cu ? a.cu[b + 1] - row0 : SM;        // its length
  const T* base = (const T*)a.qkv + (size_t)row0 * ld + h * HD;
  const int kvl = (a.cu ? S : min(a.kv_len[b], S));
  const int npb = (kvl + 31) >> 5;
  // chunk c0 of the keys and values into image pair `buf` (4 LDS-DMA pieces per wave)
  auto stage = [&](int c0, int buf) {
    char* kb = smem + buf * (2 * ATT_CHUNK * 128);
    dma_image128<T>(kb, base + a.H + (size_t)c0 * ld, ld, S - c0, tid, wave);
    dma_image128<T>(kb + ATT_CHUNK * 128, base + 2 * a.H + (size_t)c0 * ld, ld, S - c0, tid, wave);
  };
  if (DBUF && npb > 0) stage(0, 0);
  for (int k = tid; k < npb * 32; k += ATT_THREADS) mbias[k] = (k < S && (a.cu || a.mask[(size_t)b * SM + k] != 0)) ? 0.f : -INFINITY;

  const int q0 = blockIdx.y * ATT_ROWS + wave * 16;
  const bool active = q0 < S;  // wave-uniform
  const int lm = lane & 15, g = lane >> 4;
  const int q = q0 + lm, qc = min(q, S - 1);
  const short8 qf0 = gfrag<T>(base, qc, ld, 0, lane), qf1 = gfrag<T>(base, qc, ld, 1, lane);
  const float k2 = a.scale * LOG2E;
  const uint32_t rseed = attn_row_seed(drop_key(a.rng), drop_sample(a.rng, b) * a.A + h, q), thr_hi = attn_thr_hi(a.drop_thr);
  const uint32_t kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
  constexpr bool drop = DROP;

  float m = -INFINITY;
  float4v l4 = {0.f, 0.f, 0.f, 0.f};  // running sum, one partial per accumulator register
  const float4v k2v = splat4(k2);
  float4v o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0, buf = 0; c0 < npb * 32; c0 += ATT_CHUNK, buf ^= (DBUF ? 1 : 0)) {
    if (DBUF) {
      PDT_ATT_NEXT_CHUNK(c0, buf, npb * 32);
    } else {
      if (c0) __syncthreads();  // every wave is done with the previous chunk
      stage(c0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const char* kimg = smem + buf * (2 * ATT_CHUNK * 128);
    const char* vimg = kimg + ATT_CHUNK * 128;
    if (!active) continue;
    const int nb = min(ATT_CHUNK / 32, npb - (c0 >> 5));  // 32-key blocks in this chunk
    // The body exists twice: for a full chunk (four 32-key blocks: every sequence of the fixed-length benchmark
    // shape, most chunks of long ones) every bound is a compile-time constant, so the eight score blocks are
    // independent straight-line code the scheduler can interleave (MFMA latency of one block under the VALU work of
    // another); a partial chunk takes the predicated copy.
    auto chunk = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
      float4v s[8];
      float cmax = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        if (FULL || kb < 2 * nb) {
          float4v acc = {0.f, 0.f, 0.f, 0.f};
          acc = Half16<T>::mfma(row_frag(kimg, kb * 16, 0, lane), qf0, acc);
          acc = Half16<T>::mfma(row_frag(kimg, kb * 16, 1, lane), qf1, acc);
          const float4v mb = *reinterpret_cast<const float4v*>(mbias + c0 + kb * 16 + 4 * g);
          acc = acc * k2v + mb;
          cmax = fmaxf(fmaxf(cmax, acc[0]), fmaxf(fmaxf(acc[1], acc[2]), acc[3]));
          s[kb] = acc;
        }
      }
      cmax = quad_max(cmax);
      const float m_new = fmaxf(m, cmax);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;  // all keys so far masked: exp2(-inf - 0) = 0
      const float alpha = __builtin_amdgcn_exp2f(m - m_use);
      m = m_new;
      l4 *= alpha;
      const float4v nm4 = splat4(-m_use);  // (packed fp32 has add / mul / fma: subtractions are additions of a negated term)
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] *= alpha;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        if (FULL || pb < nb) {
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            float4v p = packed4(s[2 * pb + hb] + nm4);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
            l4 += p;
            if (drop) {
              float4v keep;  // drop_scale for the kept probabilities, 0 for the dropped ones
#pragma unroll
              for (int r = 0; r < 4; ++r)
                keep[r] = attn_keep(rseed, kmul0 + (uint32_t)(c0 + pb * 32 + hb * 16 + r) * ATT_GOLD, thr_hi) ? a.drop_scale : 0.f;
              p *= keep;
            }
            s[2 * pb + hb] = p;
          }
          const short8 pf = pack_pair<T>(s[2 * pb], s[2 * pb + 1]);
#pragma unroll
          for (int db = 0; db < 4; ++db) o[db] = Half16<T>::mfma(tr_frag(vimg, pb * 32, db, lane), pf, o[db]);
        }
      }
    };
    if (nb == ATT_CHUNK / 32) chunk(std::true_type{}); else chunk(std::false_type{});
  }
  if (!active) return;
  const float l = quad_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  // a row whose keys are all masked has no probability mass: lse = +inf makes every recomputed p exactly 0
  if (g == 0 && q < S) a.lse[(size_t)bh * SM + q] = l > 0.f ? (m + __log2f(l)) * (1.0f / LOG2E) : INFINITY;
  store_tile16x64<T>((T*)a.ctx + ((size_t)row0 + qc) * a.H + h * HD, o, inv, g, q < S);
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane of the row receives the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0x128>(v);  // row_ror:8
  v += dpp_f<0x124>(v);  // row_ror:4
  v += dpp_f<0x122>(v);  // row_ror:2
  v += dpp_f<0x121>(v);  // row_ror:1
  return v;
}
// ------------------------------------------------------------------------------------------------ dQ (S > 128)
// One wave = 16 query rows, sweeps the keys chunk by chunk; also produces delta = rowsum(dO * O) for the
// dK/dV kernel.  Log2-domain recomputation with the softmax scale folded into the exponent.
// dbias (nullable): fp32 [B * gridDim.y][3H] - this workgroup's column sums of dQ go to row b * gridDim.y + blockIdx.y,
// columns h*64.. (the dK/dV kernel fills the other two thirds of the same rows; pdt_colsum_finish adds the rows up)
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(AttnArgs a, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* mbias = reinterpret_cast<float*>(smem + ATT_IMG_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int SM = a.S, ld = 3 * a.H;                    // SM: longest sequence (lse / delta stride)
  const int row0 = a.cu ? a.cu[b] : b * SM;            // first token row of this sample
  const int S = a.cu ? a.cu[b + 1] - row0 : SM;        // its length
  const T* base = (const T*)a.qkv + (size_t)row0 * ld + h * HD;
  const T* dob = (const T*)a.dctx + (size_t)row0 * a.H + h * HD;
  const T* ob = (const T*)a.octx + (size_t)row0 * a.H + h * HD;
  const int kvl = (a.cu ? S : min(a.kv_len[b], S));
  const int npb = (kvl + 31) >> 5;
  auto stage = [&](int c0, int buf) {
    char* kb = smem + buf * (2 * ATT_CHUNK * 128);
    dma_image128<T>(kb, base + a.H + (size_t)c0 * ld, ld, S - c0, tid, wave);
    dma_image128<T>(kb + ATT_CHUNK * 128, base + 2 * a.H + (size_t)c0 * ld, ld, S - c0, tid, wave);
  };
  if (npb > 0) stage(0, 0);
  for (int k = tid; k < npb * 32; k += ATT_THREADS) mbias[k] = (k < S && (a.cu || a.mask[(size_t)b * SM + k] != 0)) ? 0.f : -INFINITY;

  const int q0 = blockIdx.y * ATT_ROWS + wave * 16;
  const bool active = q0 < S;  // wave-uniform
  const int lm = lane & 15, g = lane >> 4;
  const int q = q0 + lm, qc = min(q, S - 1);
  const short8 qf0 = gfrag<T>(base, qc, ld, 0, lane), qf1 = gfrag<T>(base, qc, ld, 1, lane);
  const short8 df0 = gfrag<T>(dob, qc, a.H, 0, lane), df1 = gfrag<T>(dob, qc, a.H, 1, lane);
  float ndelta;
  {
    const short8 of0 = gfrag<T>(ob, qc, a.H, 0, lane), of1 = gfrag<T>(ob, qc, a.H, 1, lane);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      d += Half16<T>::to_f(df0[j]) * Half16<T>::to_f(of0[j]) + Half16<T>::to_f(df1[j]) * Half16<T>::to_f(of1[j]);
    d = quad_sum(d);
    if (active && g == 0 && q < S) a.delta[(size_t)bh * SM + q] = d;
    ndelta = -d;
  }
  const float k2 = a.scale * LOG2E;
  // exp2(score2 + mask - lse2q) = p * scale; rows >= S get lse = +inf, i.e. p = 0 and a zero dQ row
  const float lse2q = (q < S ? a.lse[(size_t)bh * SM + qc] * LOG2E : INFINITY) - __log2f(a.scale);
  const uint32_t rseed = attn_row_seed(drop_key(a.rng), drop_sample(a.rng, b) * a.A + h, q), thr_hi = attn_thr_hi(a.drop_thr);
  const uint32_t kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
  const bool drop = a.drop_thr != 0;
  const float4v k2v = splat4(k2), nlse4 = splat4(-lse2q), nd4 = splat4(ndelta);
  float4v dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) dq[db] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0, buf = 0; c0 < npb * 32; c0 += ATT_CHUNK, buf ^= 1) {
    PDT_ATT_NEXT_CHUNK(c0, buf, npb * 32);
    const char* kimg = smem + buf * (2 * ATT_CHUNK * 128);
    const char* vimg = kimg + ATT_CHUNK * 128;
    if (!active) continue;
    const int nb = min(ATT_CHUNK / 32, npb - (c0 >> 5));
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      if (pb < nb) {
        float4v ds[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int kb16 = pb * 32 + hb * 16;
          float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = Half16<T>::mfma(row_frag(kimg, kb16, 0, lane), qf0, sc);
          sc = Half16<T>::mfma(row_frag(kimg, kb16, 1, lane), qf1, sc);
          dp = Half16<T>::mfma(row_frag(vimg, kb16, 0, lane), df0, dp);
          dp = Half16<T>::mfma(row_frag(vimg, kb16, 1, lane), df1, dp);
          const float4v mb = *reinterpret_cast<const float4v*>(mbias + c0 + kb16 + 4 * g);
          float4v ps = packed4(sc * k2v + (mb + nlse4));
#pragma unroll
          for (int r = 0; r < 4; ++r) ps[r] = __builtin_amdgcn_exp2f(ps[r]);
          float4v e = dp + nd4;
          if (drop) {
            float4v keep;  // drop_scale for the kept probabilities, 0 for the dropped ones
#pragma unroll
            for (int r = 0; r < 4; ++r)
              keep[r] = attn_keep(rseed, kmul0 + (uint32_t)(c0 + kb16 + r) * ATT_GOLD, thr_hi) ? a.drop_scale : 0.f;
            e = dp * keep + nd4;
          }
          ds[hb] = ps * e;
        }
        const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = Half16<T>::mfma(tr_frag(kimg, pb * 32, db, lane), dsf, dq[db]);
      }
    }
  }
  if (active) store_tile16x64<T>((T*)a.dqkv + ((size_t)row0 + qc) * ld + h * HD, dq, 1.0f, g, q < S);
  if (dbias) {  // rows >= S of an active wave are exactly zero (lse = +inf)
    float* csq = mbias + ATT_SMAX;  // [ATT_WAVES][64]
    if (active) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        float4v t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = row16_sum(dq[db][r]);
        if (lm == 0) *reinterpret_cast<float4v*>(csq + wave * 64 + db * 16 + 4 * g) = t;
      }
    } else {
      csq[wave * 64 + lane] = 0.f;
    }
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < ATT_WAVES; ++w) t += csq[w * 64 + tid];
      dbias[((size_t)b * gridDim.y + blockIdx.y) * 3 * a.H + h * HD + tid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV (S > 128)
// One wave = 16 keys, sweeps the queries chunk by chunk (Q and dO images); per-row lse / delta / dropout seeds
// of the whole sequence are staged in LDS once.
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(AttnArgs a, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lse_s = reinterpret_cast<float*>(smem + ATT_IMG_BYTES);
  float* del_s = lse_s + ATT_SMAX;
  uint32_t* seed_s = reinterpret_cast<uint32_t*>(del_s + ATT_SMAX);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int SM = a.S, ld = 3 * a.H;                    // SM: longest sequence (lse / delta stride)
  const int row0 = a.cu ? a.cu[b] : b * SM;            // first token row of this sample
  const int S = a.cu ? a.cu[b + 1] - row0 : SM;        // its length
  const T* base = (const T*)a.qkv + (size_t)row0 * ld + h * HD;
  const T* dob = (const T*)a.dctx + (size_t)row0 * a.H + h * HD;
  const int nqb = (S + 31) >> 5;  // 32-row query blocks
  const DropKey dk = drop_key(a.rng);
  const int kvl = (a.cu ? S : min(a.kv_len[b], S));
  const int k0 = blockIdx.y * ATT_ROWS + wave * 16;
  // keys at or beyond kv_len receive exactly zero gradient: such a workgroup / wave only writes zeros
  const bool blk_active = blockIdx.y * ATT_ROWS < kvl;  // workgroup-uniform
  const bool wave_active = k0 < kvl;                    // wave-uniform
  const int lm = lane & 15, g = lane >> 4;
  const int key = k0 + lm, kc = min(key, S - 1);
  float4v dk_[4], dv_[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) { dk_[db] = float4v{0.f, 0.f, 0.f, 0.f}; dv_[db] = float4v{0.f, 0.f, 0.f, 0.f}; }
  if (blk_active) {
    for (int i = tid; i < nqb * 32; i += ATT_THREADS) {
      // NEGATED log2-domain lse (it is only ever subtracted), -inf for the padding rows (p = 0); delta pre-multiplied by -scale
      lse_s[i] = i < S ? -a.lse[(size_t)bh * SM + i] * LOG2E : -INFINITY;
      del_s[i] = i < S ? -a.scale * a.delta[(size_t)bh * SM + i] : 0.f;
      seed_s[i] = attn_row_seed(dk, drop_sample(a.rng, b) * a.A + h, i);
    }
    const short8 kf0 = gfrag<T>(base + a.H, kc, ld, 0, lane), kf1 = gfrag<T>(base + a.H, kc, ld, 1, lane);
    const short8 vf0 = gfrag<T>(base + 2 * a.H, kc, ld, 0, lane), vf1 = gfrag<T>(base + 2 * a.H, kc, ld, 1, lane);
    const float mb = (key < S && (a.cu || a.mask[(size_t)b * SM + kc] != 0)) ? 0.f : -INFINITY;
    const uint32_t kmul = (uint32_t)key * ATT_GOLD, thr_hi = attn_thr_hi(a.drop_thr);
    const float k2 = a.scale * LOG2E;
    const float4v k2v = splat4(k2), mb4 = splat4(mb);
    const bool drop = a.drop_thr != 0;
    // chunk c0 of the queries and of dO into image pair `buf`
    auto stage = [&](int c0, int buf) {
      char* qb = smem + buf * (2 * ATT_CHUNK * 128);
      dma_image128<T>(qb, base + (size_t)c0 * ld, ld, S - c0, tid, wave);
      dma_image128<T>(qb + ATT_CHUNK * 128, dob + (size_t)c0 * a.H, a.H, S - c0, tid, wave);
    };
    if (nqb > 0) stage(0, 0);
    for (int c0 = 0, buf = 0; c0 < nqb * 32; c0 += ATT_CHUNK, buf ^= 1) {
      PDT_ATT_NEXT_CHUNK(c0, buf, nqb * 32);  // (its barrier also orders the lse / delta / seed staging before its first use)
      const char* qimg = smem + buf * (2 * ATT_CHUNK * 128);
      const char* dimg = qimg + ATT_CHUNK * 128;
      if (!wave_active) continue;
      const int nb = min(ATT_CHUNK / 32, nqb - (c0 >> 5));
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        if (qb < nb) {
          float4v pp[2], ds[2];
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            const int q16 = qb * 32 + hb * 16;
            float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            sc = Half16<T>::mfma(row_frag(qimg, q16, 0, lane), kf0, sc);
            sc = Half16<T>::mfma(row_frag(qimg, q16, 1, lane), kf1, sc);
            dp = Half16<T>::mfma(row_frag(dimg, q16, 0, lane), vf0, dp);
            dp = Half16<T>::mfma(row_frag(dimg, q16, 1, lane), vf1, dp);
            const float4v ls = *reinterpret_cast<const float4v*>(lse_s + c0 + q16 + 4 * g);
            const float4v nde = *reinterpret_cast<const float4v*>(del_s + c0 + q16 + 4 * g);
            const uint4v sd = *reinterpret_cast<const uint4v*>(seed_s + c0 + q16 + 4 * g);
            float4v p = packed4(sc * k2v + (mb4 + ls));
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
            float4v pd = p, e = dp * a.scale + nde;
            if (drop) {
              float4v keep;  // drop_scale for the kept probabilities, 0 for the dropped ones
#pragma unroll
              for (int r = 0; r < 4; ++r) keep[r] = attn_keep(sd[r], kmul, thr_hi) ? a.drop_scale : 0.f;
              pd = p * keep;
              e = dp * (keep * a.scale) + nde;
            }
            pp[hb] = pd;
            ds[hb] = p * e;
          }
          const short8 pf = pack_pair<T>(pp[0], pp[1]);
          const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            dv_[db] = Half16<T>::mfma(tr_frag(dimg, qb * 32, db, lane), pf, dv_[db]);
            dk_[db] = Half16<T>::mfma(tr_frag(qimg, qb * 32, db, lane), dsf, dk_[db]);
          }
        }
      }
    }
  }
  {
    T* outk = (T*)a.dqkv + ((size_t)row0 + kc) * ld + a.H + h * HD;
    store_tile16x64<T>(outk, dk_, 1.0f, g, key < S);
    store_tile16x64<T>(outk + a.H, dv_, 1.0f, g, key < S);
  }
  if (dbias) {  // rows of keys >= S or >= kv_len are exactly zero in dk_ / dv_ (also those of an inactive wave)
    float* cskv = reinterpret_cast<float*>(seed_s + ATT_SMAX);  // [ATT_WAVES][128] dK | dV
    __syncthreads();  // (every wave has left the sweep before the staging area behind its scalars is reused)
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      float4v tk, tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { tk[r] = row16_sum(dk_[db][r]); tv[r] = row16_sum(dv_[db][r]); }
      if (lm == 0) {
        *reinterpret_cast<float4v*>(cskv + wave * 128 + db * 16 + 4 * g) = tk;
        *reinterpret_cast<float4v*>(cskv + wave * 128 + 64 + db * 16 + 4 * g) = tv;
      }
    }
    __syncthreads();
    if (tid < 128) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < ATT_WAVES; ++w) t += cskv[w * 128 + tid];
      dbias[((size_t)b * gridDim.y + blockIdx.y) * 3 * a.H + (1 + (tid >> 6)) * a.H + h * HD + (tid & 63)] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused backward, S <= 128
// One workgroup owns a whole (batch, head): Q, K, V and dO images (4 x 16 KiB, LDS-DMA) are loaded once, wave w
// first produces dQ for query rows 16w.. (keys on the sweep axis), then dK/dV for keys 16w.. (queries on the
// sweep axis).  Compared with the two-kernel path every input is read from HBM once instead of twice, delta
// never leaves LDS, and the column sums of dQ/dK/dV (the QKV bias gradient) are reduced here without atomics:
// DPP row reductions over the 16 rows of a wave, one LDS slot per wave, then the workgroup writes its 192
// sums to dbias_partial[b][3H] (summed over b afterwards by the deterministic column-sum finisher).
// bench/micro/attn_stamp.hip builds this file with PDT_ATT_NOCOMPUTE=1 (both sweeps of the fused backward compiled out:
// what is left is its memory traffic) and / or PDT_ATT_NOSTORE=1 (dQ / dK / dV are not written)
#ifndef PDT_ATT_NOCOMPUTE
#define PDT_ATT_NOCOMPUTE 0
#endif
#ifndef PDT_ATT_NOSTORE
#define PDT_ATT_NOSTORE 0
#endif
#ifndef PDT_ATT_FUSED_WAVES_EU  // minimum waves per SIMD the register allocation of the fused kernel must allow
#define PDT_ATT_FUSED_WAVES_EU 4
#endif
// DROP: dropout on the probabilities - a compile-time switch, so that the per-element code is straight-line (with a
// run-time flag every element carried a uniform branch)
template <typename T, bool DROP>
__global__ __launch_bounds__(ATT_THREADS) __attribute__((amdgpu_waves_per_eu(PDT_ATT_FUSED_WAVES_EU, 8)))
void attn_bwd_fused128_kernel(AttnArgs a, float* __restrict__ dbias) {
  constexpr int SMAX = 128, NPB = SMAX / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // four 16-KiB images, K and V for the dQ sweep, Q and dO for the dK/dV sweep: 2 workgroups per CU.  (Re-using
  // two images for both sweeps fits 4 workgroups in LDS, but the dK/dV sweep needs ~90 VGPRs, so the register
  // file allows no more than 2 anyway - measured slower.)
  char* img0 = smem;
  char* img1 = smem + SMAX * 128;
  char* img2 = smem + 2 * SMAX * 128;
  char* img3 = smem + 3 * SMAX * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 4 * SMAX * 128);
  float* del_s = lse_s + SMAX;
  float* mbias = del_s + SMAX;
  uint32_t* seed_s = reinterpret_cast<uint32_t*>(mbias + SMAX);
  float* csq = reinterpret_cast<float*>(seed_s + SMAX);  // [ATT_WAVES][64] dQ column sums per wave
  float* cskv = csq + ATT_WAVES * 64;                    // [ATT_WAVES][128] dK|dV column sums per wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int SM = a.S, ld = 3 * a.H;                    // SM: longest sequence (lse / delta stride)
  const int row0 = a.cu ? a.cu[b] : b * SM;            // first token row of this sample
  const int S = a.cu ? a.cu[b + 1] - row0 : SM;        // its length
#ifdef PDT_ATT_STAMP  // phase time stamps of wave 0 (100 MHz counter) into the otherwise unused delta buffer
#define PDT_STAMP(i) do { if (tid == 0) a.delta[(size_t)bh * SM + (i)] = (float)(wall_clock64() & 0xFFFFFF); } while (0)
#else
#define PDT_STAMP(i) do { } while (0)
#endif
  PDT_STAMP(0);
  const T* base = (const T*)a.qkv + (size_t)row0 * ld + h * HD;
  const T* dob = (const T*)a.dctx + (size_t)row0 * a.H + h * HD;
  const T* ob = (const T*)a.octx + (size_t)row0 * a.H + h * HD;
  const int kvl = (a.cu ? S : min(a.kv_len[b], S));
  const int npb = (kvl + 31) >> 5, nqb = (S + 31) >> 5;
  const DropKey dk = drop_key(a.rng);
  const uint32_t thr_hi = attn_thr_hi(a.drop_thr);
  const int lm = lane & 15, g = lane >> 4;
  const float k2 = a.scale * LOG2E;  // scores -> log2 domain
  constexpr bool drop = DROP;
  const int r0 = wave * 16;  // this wave's query rows in the dQ sweep, its keys in the dK/dV sweep

  // Issue order = completion order (vmcnt is in order): the wave's own rows and the per-row scalars first, then the
  // K / V images the dQ sweep needs, then the Q / dO images, which are only needed by the dK/dV sweep and keep
  // landing while the dQ sweep runs (s_waitcnt vmcnt(4) leaves exactly their four pieces outstanding).
  // own query rows of Q and dO straight from global memory (B-operand fragments), and delta = rowsum(dO * O)
  short8 qf0, qf1, df0, df1;
  {
    const int q = r0 + lm, qc = min(q, S - 1);
    qf0 = gfrag<T>(base, qc, ld, 0, lane); qf1 = gfrag<T>(base, qc, ld, 1, lane);
    df0 = gfrag<T>(dob, qc, a.H, 0, lane); df1 = gfrag<T>(dob, qc, a.H, 1, lane);
    const short8 of0 = gfrag<T>(ob, qc, a.H, 0, lane), of1 = gfrag<T>(ob, qc, a.H, 1, lane);
    // raw per-row scalars: loaded here, converted after the DMA block so that no wait sits in front of it
    int mask_raw = 0;
    float lse_raw = 0.f;
    if (tid < S) {
      mask_raw = a.cu ? 1 : a.mask[(size_t)b * SM + tid];
      lse_raw = a.lse[(size_t)bh * SM + tid];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    dma_image128<T>(img0, base + a.H, ld, S, tid, wave);      // K
    dma_image128<T>(img1, base + 2 * a.H, ld, S, tid, wave);  // V
    asm volatile("" ::: "memory");
    dma_image128<T>(img2, base, ld, S, tid, wave);            // Q
    dma_image128<T>(img3, dob, a.H, S, tid, wave);            // dO
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (tid < SMAX) {
      mbias[tid] = (tid < S && mask_raw != 0) ? 0.f : -INFINITY;
      // NEGATED log2-domain lse (it is only ever subtracted); -inf for the padding rows makes their recomputed probabilities exactly 0
      lse_s[tid] = tid < S ? -lse_raw * LOG2E : -INFINITY;
      seed_s[tid] = attn_row_seed(dk, drop_sample(a.rng, b) * a.A + h, tid);
    }
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      d += Half16<T>::to_f(df0[j]) * Half16<T>::to_f(of0[j]) + Half16<T>::to_f(df1[j]) * Half16<T>::to_f(of1[j]);
    d = quad_sum(d);
    if (q >= S) {  // clamped duplicates of row S-1: no contribution
      d = 0.f;
      qf0 = qf1 = df0 = df1 = short8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (g == 0) del_s[q] = d;
  }
  __builtin_amdgcn_sched_barrier(0);
  // own rows, scalars, K and V have landed and this wave's LDS writes are done; Q / dO may still be in flight
  // (a raw barrier: __syncthreads() would drain vmcnt to 0)
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  PDT_STAMP(1);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  PDT_STAMP(2);

  // ---- dQ for query rows r0..r0+15 (keys on the sweep axis)
  float4v dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) dq[db] = float4v{0.f, 0.f, 0.f, 0.f};
  if (r0 < S && !PDT_ATT_NOCOMPUTE) {
    const int q = r0 + lm;
    // exp2(score2 + mask - lse2q) = p * scale: the softmax scale of dS is folded into the exponent
    const float lse2q = -lse_s[q] - __log2f(a.scale), ndelta = -del_s[q];
    const uint32_t rseed = seed_s[q], kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
    const float4v k2v = splat4(k2), nlse4 = splat4(-lse2q), nd4 = splat4(ndelta);
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      if (pb < npb) {
        float4v ds[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int kb16 = pb * 32 + hb * 16;
          float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = Half16<T>::mfma(row_frag(img0, kb16, 0, lane), qf0, sc);
          sc = Half16<T>::mfma(row_frag(img0, kb16, 1, lane), qf1, sc);
          dp = Half16<T>::mfma(row_frag(img1, kb16, 0, lane), df0, dp);
          dp = Half16<T>::mfma(row_frag(img1, kb16, 1, lane), df1, dp);
          const float4v mb = *reinterpret_cast<const float4v*>(mbias + kb16 + 4 * g);
          float4v ps = packed4(sc * k2v + (mb + nlse4));
#pragma unroll
          for (int r = 0; r < 4; ++r) ps[r] = __builtin_amdgcn_exp2f(ps[r]);
          float4v e = dp + nd4;
          if (drop) {
            float4v keep;  // drop_scale for the kept probabilities, 0 for the dropped ones
#pragma unroll
            for (int r = 0; r < 4; ++r)
              keep[r] = attn_keep(rseed, kmul0 + (uint32_t)(kb16 + r) * ATT_GOLD, thr_hi) ? a.drop_scale : 0.f;
            e = dp * keep + nd4;
          }
          ds[hb] = ps * e;
        }
        const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = Half16<T>::mfma(tr_frag(img0, pb * 32, db, lane), dsf, dq[db]);
      }
    }
  }
  // the Q / dO images are needed from here on: everything this wave issued has landed (the dQ sweep issued no
  // memory operation), then every wave's share has
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (r0 < S) {
    const int q = r0 + lm;
    store_tile16x64<T>((T*)a.dqkv + ((size_t)row0 + min(q, S - 1)) * ld + h * HD, dq, 1.0f, g, q < S && !PDT_ATT_NOSTORE);
    if (dbias) {  // rows >= S are exactly zero (zero Q / dO fragments, p = 0)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        float4v t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = row16_sum(dq[db][r]);
        if (lm == 0) *reinterpret_cast<float4v*>(csq + wave * 64 + db * 16 + 4 * g) = t;
      }
    }
  } else if (dbias) {
    csq[wave * 64 + lane] = 0.f;
  }
  PDT_STAMP(3);

  // ---- dK, dV for keys r0..r0+15 (queries on the sweep axis); keys at or beyond kv_len get exactly zero
  const int key = r0 + lm;
  const bool kv_active = r0 < kvl;
  short8 kf0, kf1, vf0, vf1;  // the wave's own keys
  if (kv_active) {
    kf0 = row_frag(img0, r0, 0, lane); kf1 = row_frag(img0, r0, 1, lane);
    vf0 = row_frag(img1, r0, 0, lane); vf1 = row_frag(img1, r0, 1, lane);
  }
  const float mbk = mbias[key];
  PDT_STAMP(4);
  float4v dk_[4], dv_[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) { dk_[db] = float4v{0.f, 0.f, 0.f, 0.f}; dv_[db] = float4v{0.f, 0.f, 0.f, 0.f}; }
  if (kv_active && !PDT_ATT_NOCOMPUTE) {
    const uint32_t kmul = (uint32_t)key * ATT_GOLD;
    const float4v k2v = splat4(k2), mb4 = splat4(mbk);
#pragma unroll
    for (int qb = 0; qb < NPB; ++qb) {
      if (qb < nqb) {
        float4v pp[2], ds[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int q16 = qb * 32 + hb * 16;
          float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = Half16<T>::mfma(row_frag(img2, q16, 0, lane), kf0, sc);
          sc = Half16<T>::mfma(row_frag(img2, q16, 1, lane), kf1, sc);
          dp = Half16<T>::mfma(row_frag(img3, q16, 0, lane), vf0, dp);
          dp = Half16<T>::mfma(row_frag(img3, q16, 1, lane), vf1, dp);
          const float4v ls = *reinterpret_cast<const float4v*>(lse_s + q16 + 4 * g);
          const float4v nde = *reinterpret_cast<const float4v*>(del_s + q16 + 4 * g) * (-a.scale);
          const uint4v sd = *reinterpret_cast<const uint4v*>(seed_s + q16 + 4 * g);
          float4v p = packed4(sc * k2v + (mb4 + ls));
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
          float4v pd = p, e = dp * a.scale + nde;
          if (drop) {
            float4v keep;  // drop_scale for the kept probabilities, 0 for the dropped ones
#pragma unroll
            for (int r = 0; r < 4; ++r) keep[r] = attn_keep(sd[r], kmul, thr_hi) ? a.drop_scale : 0.f;
            pd = p * keep;
            e = dp * (keep * a.scale) + nde;
          }
          pp[hb] = pd;
          ds[hb] = p * e;
        }
        const short8 pf = pack_pair<T>(pp[0], pp[1]);
        const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          dv_[db] = Half16<T>::mfma(tr_frag(img3, qb * 32, db, lane), pf, dv_[db]);
          dk_[db] = Half16<T>::mfma(tr_frag(img2, qb * 32, db, lane), dsf, dk_[db]);
        }
      }
    }
  }
  {
    T* outk = (T*)a.dqkv + ((size_t)row0 + min(key, S - 1)) * ld + a.H + h * HD;
    store_tile16x64<T>(outk, dk_, 1.0f, g, key < S && !PDT_ATT_NOSTORE);
    store_tile16x64<T>(outk + a.H, dv_, 1.0f, g, key < S && !PDT_ATT_NOSTORE);
  }
  PDT_STAMP(5);
  if (dbias) {
    // rows of keys >= S or >= kv_len are exactly zero in dk_/dv_
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      float4v tk, tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { tk[r] = row16_sum(dk_[db][r]); tv[r] = row16_sum(dv_[db][r]); }
      if (lm == 0) {
        *reinterpret_cast<float4v*>(cskv + wave * 128 + db * 16 + 4 * g) = tk;
        *reinterpret_cast<float4v*>(cskv + wave * 128 + 64 + db * 16 + 4 * g) = tv;
      }
    }
    __syncthreads();
    if (tid < 192) {
      float t = 0.f;
      if (tid < 64) {
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) t += csq[w * 64 + tid];
      } else {
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) t += cskv[w * 128 + tid - 64];
      }
      dbias[(size_t)b * 3 * a.H + (tid >> 6) * a.H + h * HD + (tid & 63)] = t;
    }
  }
  PDT_STAMP(6);
#undef PDT_STAMP
}

// more than 64 KiB of dynamic LDS needs the attribute; set once per kernel (not a stream operation: done at the first
// launch, i.e. in the warm-up of a step that is later captured into a graph)
static int allow_big_lds(const void* kern) {
  static std::mutex mu;
  static std::unordered_set<const void*> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(kern)) return 0;
  // (the largest footprint any launch of these kernels asks for: two image pairs + scalars + column-sum staging)
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) return (int)e;
  done.insert(kern);
  return 0;
}

template <typename KernT, typename... Extra>
static int launch_attn(KernT kern, const AttnArgs& a, size_t lds, hipStream_t stream, Extra... extra) {
  if (lds > 80 * 1024) return -3;
  if (int e = allow_big_lds((const void*)kern)) return e;
  hipLaunchKernelGGL(kern, dim3(a.B * a.A, (a.S + ATT_ROWS - 1) / ATT_ROWS), dim3(ATT_THREADS), lds, stream, a, extra...);
  return (int)hipGetLastError();
}

// which: 0 forward, 1 dQ, 2 dK/dV
template <typename T>
static int attn_dispatch(int which, const AttnArgs& a, float* dbias, hipStream_t s) {
  const size_t imgs = (size_t)ATT_IMG_BYTES;
  if (which == 0) {
    const bool drop = a.drop_thr != 0;
    if (a.S > ATT_CHUNK) {
      const size_t lds = imgs + ATT_SMAX * sizeof(float);
      return drop ? launch_attn(attn_fwd_kernel<T, true, true>, a, lds, s) : launch_attn(attn_fwd_kernel<T, false, true>, a, lds, s);
    }
    const size_t lds = imgs / 2 + ATT_SMAX * sizeof(float);
    return drop ? launch_attn(attn_fwd_kernel<T, true, false>, a, lds, s) : launch_attn(attn_fwd_kernel<T, false, false>, a, lds, s);
  }
  if (which == 1) return launch_attn(attn_bwd_dq_kernel<T>, a, imgs + (ATT_SMAX + ATT_WAVES * 64) * sizeof(float), s, dbias);
  return launch_attn(attn_bwd_dkv_kernel<T>, a, imgs + (3 * ATT_SMAX + ATT_WAVES * 128) * sizeof(float), s, dbias);
}

}  // namespace pdt

using namespace pdt;

static int check_attn(int B, int S, int A) { return (B <= 0 || S <= 0 || S > ATT_SMAX || A <= 0) ? -3 : 0; }

extern "C" int pdt_attn_fwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, const int32_t* cu,
                            void* ctx, float* lse, int B, int S, int A, float drop_scale, uint32_t drop_thr, PdtRng rng,
                            void* stream) {
  if (int e = check_attn(B, S, A)) return e;
  if (!cu && (!mask || !kv_len)) return -3;
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.kv_len = kv_len; a.cu = cu; a.ctx = ctx; a.lse = lse;
  a.B = B; a.S = S; a.A = A; a.H = A * HD; a.scale = ATT_SCALE;
  a.drop_scale = drop_scale; a.drop_thr = drop_thr; a.rng = rng;
  return dtype == PDT_F16 ? attn_dispatch<_Float16>(0, a, nullptr, (hipStream_t)stream)
                          : attn_dispatch<__bf16>(0, a, nullptr, (hipStream_t)stream);
}

template <typename T>
static int launch_fused128(const AttnArgs& a, float* dbias, hipStream_t stream) {
  const size_t lds = (size_t)4 * 128 * 128 + (size_t)(4 * 128 + ATT_WAVES * 192) * sizeof(float);
  const bool drop = a.drop_thr != 0;
  auto kern = drop ? attn_bwd_fused128_kernel<T, true> : attn_bwd_fused128_kernel<T, false>;
  static bool configured[2] = {false, false};  // once per process: not a stream operation, keep it out of graph captures
  if (!configured[drop]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    configured[drop] = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.A), dim3(ATT_THREADS), lds, stream, a, dbias);
  return (int)hipGetLastError();
}

extern "C" int pdt_attn_bwd_fuses_dbias(int S) { return S <= ATT_SMAX ? 1 : 0; }
extern "C" int pdt_attn_bwd_dbias_rows(int B, int S) { return B * ((S + ATT_ROWS - 1) / ATT_ROWS); }

extern "C" int pdt_attn_bwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, const int32_t* cu,
                            const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, float* dbias,
                            int B, int S, int A, float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream) {
  if (int e = check_attn(B, S, A)) return e;
  if (dbias && !pdt_attn_bwd_fuses_dbias(S)) return -3;
  if (!cu && (!mask || !kv_len)) return -3;
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.kv_len = kv_len; a.cu = cu; a.octx = ctx; a.dctx = dctx; a.lse = const_cast<float*>(lse);
  a.delta = delta; a.dqkv = dqkv;
  a.B = B; a.S = S; a.A = A; a.H = A * HD; a.scale = ATT_SCALE;
  a.drop_scale = drop_scale; a.drop_thr = drop_thr; a.rng = rng;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 128) return dtype == PDT_F16 ? launch_fused128<_Float16>(a, dbias, s) : launch_fused128<__bf16>(a, dbias, s);
  int e = dtype == PDT_F16 ? attn_dispatch<_Float16>(1, a, dbias, s) : attn_dispatch<__bf16>(1, a, dbias, s);
  if (e) return e;
  return dtype == PDT_F16 ? attn_dispatch<_Float16>(2, a, dbias, s) : attn_dispatch<__bf16>(2, a, dbias, s);
}
