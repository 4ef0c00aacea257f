// Fused scaled-dot-product attention for BERT-class encoders on gfx950: head_dim 64, S <= 512, key padding
// mask, dropout on the probabilities.  Forward keeps the whole key/value panel of one (batch, head) in LDS
// (S=512: 2 x 64 KiB) and a full score row in registers, so there is no online-softmax rescale.
// Backward recomputes the probabilities from the saved log-sum-exp in two sweeps with no atomics on dQ/dK/dV:
//   dQ sweep    : one wave = 16 query rows, sweeps the keys       (S^T, dP^T -> dS^T -> dQ^T)
//   dK/dV sweep : one wave = 16 keys,       sweeps the queries    (S, dP -> P, dS -> dV^T, dK^T)
// For S <= 128 both sweeps run in ONE kernel on shared LDS images (and the QKV bias gradient is reduced in
// it); longer sequences use one kernel per sweep.
// Products are arranged "key on the register axis" so that an accumulator tile is directly the B operand
// of the next MFMA (16x16x32, k index permuted consistently on both operands), and the other operand is
// fetched from a row-major LDS image with ds_read_b64_tr_b16.  One XOR-swizzled image layout
// (16-B chunk ^= row & 6) is conflict-free for both the ds_read_b128 row reads and the transposed reads.
#include "common.hpp"

namespace pdt {

constexpr int HD = 64;       // head dim
constexpr int ATT_WAVES = 8;  // waves per workgroup: 8 x 16 = 128 query rows (or keys) share one K/V (or Q/dO) image
constexpr int ATT_THREADS = ATT_WAVES * 64;
constexpr int ATT_ROWS = ATT_WAVES * 16;

struct AttnArgs {
  const void* qkv;   // [B*S, 3H]
  const int32_t* mask;
  const int32_t* kv_len;
  void* ctx;         // fwd out [B*S, H]
  float* lse;        // [B, A, S]
  const void* octx;  // bwd: forward output
  const void* dctx;  // bwd: grad of ctx
  float* delta;      // [B, A, S]
  void* dqkv;        // [B*S, 3H]
  int B, S, A, H;
  float scale;       // 1/sqrt(64)
  float drop_scale;
  uint32_t drop_thr;  // round(p * 65536), 0 = off
  RngArgs rng;
};

// Dropout decision for probability (bh, q, key): a function of (seed, offset, site, micro, b*A+h, q, key) only,
// so every kernel agrees whatever its lane mapping is.  It is split in two so that the per-element cost is one
// 32-bit multiply: a strongly mixed 32-bit seed per query row (computed once per row - in the dK/dV sweeps it
// is staged in LDS), and per element  x = (row_seed ^ key * GOLD) * MIX,  keep <=> high16(x) >= thr.
// key * GOLD is a Weyl sequence over the keys; an unrolled sweep forms it with one add per element.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
constexpr uint32_t ATT_GOLD = 0x9E3779B9u, ATT_MIX = 0x85EBCA6Bu;
__device__ __forceinline__ uint32_t attn_row_seed(const DropKey& dk, uint32_t bh, uint32_t q) {
  uint32_t x = lowbias32(q ^ dk.key.x);
  return lowbias32(x ^ (bh * 0x9E3779B9u) ^ dk.key.y ^ (dk.stream * 0x85EBCA6Bu) ^ (dk.micro * 0xC2B2AE35u));
}
// thr_hi = drop threshold << 16 (see attn_thr_hi); kmul = key * ATT_GOLD
__device__ __forceinline__ bool attn_keep(uint32_t row_seed, uint32_t kmul, uint32_t thr_hi) {
  return (row_seed ^ kmul) * ATT_MIX >= thr_hi;
}
__device__ __forceinline__ uint32_t attn_thr_hi(uint32_t thr) { return (thr > 65535u ? 65535u : thr) << 16; }

// cooperative load of `nrows_pad` rows of 64 16-bit values (row stride `stride` elements) into an image
template <typename T>
__device__ __forceinline__ void load_image(char* img, const T* src, int stride, int nvalid, int nrows_pad, int tid) {
  for (int idx = tid; idx < nrows_pad * 8; idx += ATT_THREADS) {
    const int row = idx >> 3, c = idx & 7;
    short8 v = short8{0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nvalid) v = *reinterpret_cast<const short8*>(src + (size_t)row * stride + c * 8);
    *reinterpret_cast<short8*>(img + row * 128 + ((c ^ (row & 6)) << 4)) = v;
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
template <typename T>
__device__ __forceinline__ short8 pack_pair(const float4v& a, const float4v& b) {
  short8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = Half16<T>::from_f(a[j]); r[4 + j] = Half16<T>::from_f(b[j]); }
  return r;
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

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int SMAX>
__global__ __launch_bounds__(ATT_THREADS) void attn_fwd_kernel(AttnArgs a) {
  constexpr int NPB = SMAX / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = smem + SMAX * 128;
  float* mbias = reinterpret_cast<float*>(smem + 2 * SMAX * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int S = a.S, ld = 3 * a.H;
  const T* base = (const T*)a.qkv + (size_t)b * S * ld + h * HD;
  const int kvl = min(a.kv_len[b], S);
  const int npb = (kvl + 31) >> 5;
  load_image<T>(kimg, base + a.H, ld, S, npb * 32, tid);
  load_image<T>(vimg, base + 2 * a.H, ld, S, npb * 32, tid);
  for (int k = tid; k < npb * 32; k += ATT_THREADS) mbias[k] = (k < S && a.mask[(size_t)b * S + k] != 0) ? 0.f : -INFINITY;
  __syncthreads();

  const int q0 = blockIdx.y * ATT_ROWS + wave * 16;
  if (q0 >= S) return;
  const int lm = lane & 15, g = lane >> 4;
  const int q = q0 + lm, qc = min(q, S - 1);
  const short8 qf0 = gfrag<T>(base, qc, ld, 0, lane), qf1 = gfrag<T>(base, qc, ld, 1, lane);

  float4v s[2 * NPB];
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 2 * NPB; ++kb) {
    if (kb < 2 * npb) {
      float4v acc = {0.f, 0.f, 0.f, 0.f};
      acc = Half16<T>::mfma(row_frag(kimg, kb * 16, 0, lane), qf0, acc);
      acc = Half16<T>::mfma(row_frag(kimg, kb * 16, 1, lane), qf1, acc);
      const float4v mb = *reinterpret_cast<const float4v*>(mbias + kb * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[r] = acc[r] * a.scale + mb[r]; mx = fmaxf(mx, acc[r]); }
      s[kb] = acc;
    }
  }
  mx = quad_max(mx);
  if (mx == -INFINITY) mx = 0.f;
  float l = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2 * NPB; ++kb) {
    if (kb < 2 * npb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[kb][r] = __expf(s[kb][r] - mx); l += s[kb][r]; }
    }
  }
  l = quad_sum(l);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (g == 0 && q < S) a.lse[(size_t)bh * S + q] = mx + __logf(l);

  const uint32_t rseed = attn_row_seed(drop_key(a.rng), bh, q), thr_hi = attn_thr_hi(a.drop_thr);
  const uint32_t kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
  float4v o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    if (pb < npb) {
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = s[2 * pb + hb][r] * inv;
          if (a.drop_thr != 0)
            p = attn_keep(rseed, kmul0 + (uint32_t)(pb * 32 + hb * 16 + r) * ATT_GOLD, thr_hi) ? p * a.drop_scale : 0.f;
          s[2 * pb + hb][r] = p;
        }
      }
      const short8 pf = pack_pair<T>(s[2 * pb], s[2 * pb + 1]);
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = Half16<T>::mfma(tr_frag(vimg, pb * 32, db, lane), pf, o[db]);
    }
  }
  if (q < S) {
    T* out = (T*)a.ctx + ((size_t)b * S + q) * a.H + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      short4v pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = Half16<T>::from_f(o[db][r]);
      *reinterpret_cast<short4v*>(out + db * 16 + 4 * g) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <typename T, int SMAX>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dq_kernel(AttnArgs a) {
  constexpr int NPB = SMAX / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = smem + SMAX * 128;
  float* mbias = reinterpret_cast<float*>(smem + 2 * SMAX * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int S = a.S, ld = 3 * a.H;
  const T* base = (const T*)a.qkv + (size_t)b * S * ld + h * HD;
  const int kvl = min(a.kv_len[b], S);
  const int npb = (kvl + 31) >> 5;
  load_image<T>(kimg, base + a.H, ld, S, npb * 32, tid);
  load_image<T>(vimg, base + 2 * a.H, ld, S, npb * 32, tid);
  for (int k = tid; k < npb * 32; k += ATT_THREADS) mbias[k] = (k < S && a.mask[(size_t)b * S + k] != 0) ? 0.f : -INFINITY;
  __syncthreads();

  const int q0 = blockIdx.y * ATT_ROWS + wave * 16;
  if (q0 >= S) return;
  const int lm = lane & 15, g = lane >> 4;
  const int q = q0 + lm, qc = min(q, S - 1);
  const short8 qf0 = gfrag<T>(base, qc, ld, 0, lane), qf1 = gfrag<T>(base, qc, ld, 1, lane);
  const T* dob = (const T*)a.dctx + (size_t)b * S * a.H + h * HD;
  const T* ob = (const T*)a.octx + (size_t)b * S * a.H + h * HD;
  const short8 df0 = gfrag<T>(dob, qc, a.H, 0, lane), df1 = gfrag<T>(dob, qc, a.H, 1, lane);
  float delta;
  {
    const short8 of0 = gfrag<T>(ob, qc, a.H, 0, lane), of1 = gfrag<T>(ob, qc, a.H, 1, lane);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      d += Half16<T>::to_f(df0[j]) * Half16<T>::to_f(of0[j]) + Half16<T>::to_f(df1[j]) * Half16<T>::to_f(of1[j]);
    delta = quad_sum(d);
    if (g == 0 && q < S) a.delta[(size_t)bh * S + q] = delta;
  }
  const float lse = a.lse[(size_t)bh * S + qc];
  const uint32_t rseed = attn_row_seed(drop_key(a.rng), bh, q), thr_hi = attn_thr_hi(a.drop_thr);
  const uint32_t kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
  float4v dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) dq[db] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pb = 0; pb < npb; ++pb) {
    float4v ds[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      const int kb16 = pb * 32 + hb * 16;
      float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      sc = Half16<T>::mfma(row_frag(kimg, kb16, 0, lane), qf0, sc);
      sc = Half16<T>::mfma(row_frag(kimg, kb16, 1, lane), qf1, sc);
      dp = Half16<T>::mfma(row_frag(vimg, kb16, 0, lane), df0, dp);
      dp = Half16<T>::mfma(row_frag(vimg, kb16, 1, lane), df1, dp);
      const float4v mb = *reinterpret_cast<const float4v*>(mbias + kb16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(sc[r] * a.scale + mb[r] - lse);
        float d = dp[r];
        if (a.drop_thr != 0) d = attn_keep(rseed, kmul0 + (uint32_t)(kb16 + r) * ATT_GOLD, thr_hi) ? d * a.drop_scale : 0.f;
        ds[hb][r] = p * (d - delta) * a.scale;
      }
    }
    const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = Half16<T>::mfma(tr_frag(kimg, pb * 32, db, lane), dsf, dq[db]);
  }
  if (q < S) {
    T* out = (T*)a.dqkv + ((size_t)b * S + q) * ld + h * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      short4v pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = Half16<T>::from_f(dq[db][r]);
      *reinterpret_cast<short4v*>(out + db * 16 + 4 * g) = pk;
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <typename T, int SMAX>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_dkv_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;
  char* dimg = smem + SMAX * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * SMAX * 128);
  float* del_s = lse_s + SMAX;
  uint32_t* seed_s = reinterpret_cast<uint32_t*>(del_s + SMAX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int S = a.S, ld = 3 * a.H;
  const T* base = (const T*)a.qkv + (size_t)b * S * ld + h * HD;
  const T* dob = (const T*)a.dctx + (size_t)b * S * a.H + h * HD;
  const int nqb = (S + 31) >> 5;  // 32-row query blocks
  const DropKey dk = drop_key(a.rng);
  const int kvl = min(a.kv_len[b], S);
  const int k0 = blockIdx.y * ATT_ROWS + wave * 16;
  // keys at or beyond kv_len receive exactly zero gradient: the block only has to write zeros
  const bool active = blockIdx.y * ATT_ROWS < kvl;
  if (active) {
    load_image<T>(qimg, base, ld, S, nqb * 32, tid);
    load_image<T>(dimg, dob, a.H, S, nqb * 32, tid);
    for (int i = tid; i < nqb * 32; i += ATT_THREADS) {
      lse_s[i] = i < S ? a.lse[(size_t)bh * S + i] : 0.f;
      del_s[i] = i < S ? a.delta[(size_t)bh * S + i] : 0.f;
      seed_s[i] = attn_row_seed(dk, bh, i);
    }
  }
  __syncthreads();
  if (k0 >= S) return;
  const int lm = lane & 15, g = lane >> 4;
  const int key = k0 + lm, kc = min(key, S - 1);
  float4v dk_[4], dv_[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) { dk_[db] = float4v{0.f, 0.f, 0.f, 0.f}; dv_[db] = float4v{0.f, 0.f, 0.f, 0.f}; }
  if (active && k0 < kvl) {
    const short8 kf0 = gfrag<T>(base + a.H, kc, ld, 0, lane), kf1 = gfrag<T>(base + a.H, kc, ld, 1, lane);
    const short8 vf0 = gfrag<T>(base + 2 * a.H, kc, ld, 0, lane), vf1 = gfrag<T>(base + 2 * a.H, kc, ld, 1, lane);
    const float mb = (key < S && a.mask[(size_t)b * S + kc] != 0) ? 0.f : -INFINITY;
    const uint32_t kmul = (uint32_t)key * ATT_GOLD, thr_hi = attn_thr_hi(a.drop_thr);
#pragma unroll 1
    for (int qb = 0; qb < nqb; ++qb) {
      float4v pp[2], ds[2];
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        const int q16 = qb * 32 + hb * 16;
        float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        sc = Half16<T>::mfma(row_frag(qimg, q16, 0, lane), kf0, sc);
        sc = Half16<T>::mfma(row_frag(qimg, q16, 1, lane), kf1, sc);
        dp = Half16<T>::mfma(row_frag(dimg, q16, 0, lane), vf0, dp);
        dp = Half16<T>::mfma(row_frag(dimg, q16, 1, lane), vf1, dp);
        const float4v ls = *reinterpret_cast<const float4v*>(lse_s + q16 + 4 * g);
        const float4v de = *reinterpret_cast<const float4v*>(del_s + q16 + 4 * g);
        const uint4v sd = *reinterpret_cast<const uint4v*>(seed_s + q16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q16 + 4 * g + r;
          float p = __expf(sc[r] * a.scale + mb - ls[r]);
          if (q >= S) p = 0.f;
          float d = dp[r], pd = p;
          if (a.drop_thr != 0) {
            const bool keep = attn_keep(sd[r], kmul, thr_hi);
            d = keep ? d * a.drop_scale : 0.f;
            pd = keep ? p * a.drop_scale : 0.f;
          }
          pp[hb][r] = pd;
          ds[hb][r] = p * (d - de[r]) * a.scale;
        }
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
  if (key < S) {
    T* outk = (T*)a.dqkv + ((size_t)b * S + key) * ld + a.H + h * HD;
    T* outv = outk + a.H;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      short4v pk, pv;
#pragma unroll
      for (int r = 0; r < 4; ++r) { pk[r] = Half16<T>::from_f(dk_[db][r]); pv[r] = Half16<T>::from_f(dv_[db][r]); }
      *reinterpret_cast<short4v*>(outk + db * 16 + 4 * g) = pk;
      *reinterpret_cast<short4v*>(outv + db * 16 + 4 * g) = pv;
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused backward, S <= 128
// One workgroup owns a whole (batch, head): Q, K, V and dO images (4 x 16 KiB) are loaded once, wave w
// first produces dQ for query rows 16w.. (keys on the sweep axis), then dK/dV for keys 16w.. (queries on the
// sweep axis).  Compared with the two-kernel path every input is read from HBM once instead of twice, delta
// never leaves LDS, and the column sums of dQ/dK/dV (the QKV bias gradient) are reduced here - 16-lane
// shuffles, LDS float atomics across the 8 waves, then 192 global atomics per workgroup into `dbias`.
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attn_bwd_fused128_kernel(AttnArgs a, float* __restrict__ dbias) {
  constexpr int SMAX = 128, NPB = SMAX / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qimg = smem;
  char* kimg = smem + SMAX * 128;
  char* vimg = smem + 2 * SMAX * 128;
  char* dimg = smem + 3 * SMAX * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 4 * SMAX * 128);
  float* del_s = lse_s + SMAX;
  float* mbias = del_s + SMAX;
  float* cs = mbias + SMAX;  // [3][64]
  uint32_t* seed_s = reinterpret_cast<uint32_t*>(cs + 192);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.A, h = bh % a.A;
  const int S = a.S, ld = 3 * a.H;
  const T* base = (const T*)a.qkv + (size_t)b * S * ld + h * HD;
  const T* dob = (const T*)a.dctx + (size_t)b * S * a.H + h * HD;
  const T* ob = (const T*)a.octx + (size_t)b * S * a.H + h * HD;
  const int kvl = min(a.kv_len[b], S);
  const int npb = (kvl + 31) >> 5, nqb = (S + 31) >> 5;
  const DropKey dk = drop_key(a.rng);
  const uint32_t thr_hi = attn_thr_hi(a.drop_thr);
  load_image<T>(qimg, base, ld, S, nqb * 32, tid);
  load_image<T>(kimg, base + a.H, ld, S, npb * 32, tid);
  load_image<T>(vimg, base + 2 * a.H, ld, S, npb * 32, tid);
  load_image<T>(dimg, dob, a.H, S, nqb * 32, tid);
  for (int i = tid; i < SMAX; i += ATT_THREADS) {
    mbias[i] = (i < S && a.mask[(size_t)b * S + i] != 0) ? 0.f : -INFINITY;
    lse_s[i] = i < S ? a.lse[(size_t)bh * S + i] : 0.f;
    seed_s[i] = attn_row_seed(dk, bh, i);
  }
  if (tid < 192) cs[tid] = 0.f;
  const int lm = lane & 15, g = lane >> 4;
  const int r0 = wave * 16;  // this wave's query rows in the dQ phase, its keys in the dK/dV phase
  {
    // delta = rowsum(dO * O) of the wave's 16 query rows
    float d = 0.f;
    const int q = r0 + lm;
    if (q < S) {
      const short8 df0 = gfrag<T>(dob, q, a.H, 0, lane), df1 = gfrag<T>(dob, q, a.H, 1, lane);
      const short8 of0 = gfrag<T>(ob, q, a.H, 0, lane), of1 = gfrag<T>(ob, q, a.H, 1, lane);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        d += Half16<T>::to_f(df0[j]) * Half16<T>::to_f(of0[j]) + Half16<T>::to_f(df1[j]) * Half16<T>::to_f(of1[j]);
    }
    d = quad_sum(d);
    if (g == 0) del_s[q] = d;
  }
  __syncthreads();

  // sum over the 16 rows a wave holds (lane&15), result valid in lanes with (lane&15)==0
  auto rows16 = [](float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
  };

  // ---- dQ for query rows r0..r0+15
  if (r0 < S) {
    const int q = r0 + lm;
    const short8 qf0 = row_frag(qimg, r0, 0, lane), qf1 = row_frag(qimg, r0, 1, lane);
    const short8 df0 = row_frag(dimg, r0, 0, lane), df1 = row_frag(dimg, r0, 1, lane);
    const float lse = lse_s[q], delta = del_s[q];
    const uint32_t rseed = seed_s[q], kmul0 = (uint32_t)(4 * g) * ATT_GOLD;
    float4v dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
      if (pb < npb) {
        float4v ds[2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          const int kb16 = pb * 32 + hb * 16;
          float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = Half16<T>::mfma(row_frag(kimg, kb16, 0, lane), qf0, sc);
          sc = Half16<T>::mfma(row_frag(kimg, kb16, 1, lane), qf1, sc);
          dp = Half16<T>::mfma(row_frag(vimg, kb16, 0, lane), df0, dp);
          dp = Half16<T>::mfma(row_frag(vimg, kb16, 1, lane), df1, dp);
          const float4v mb = *reinterpret_cast<const float4v*>(mbias + kb16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(sc[r] * a.scale + mb[r] - lse);
            float d = dp[r];
            if (a.drop_thr != 0) d = attn_keep(rseed, kmul0 + (uint32_t)(kb16 + r) * ATT_GOLD, thr_hi) ? d * a.drop_scale : 0.f;
            ds[hb][r] = q < S ? p * (d - delta) * a.scale : 0.f;
          }
        }
        const short8 dsf = pack_pair<T>(ds[0], ds[1]);
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = Half16<T>::mfma(tr_frag(kimg, pb * 32, db, lane), dsf, dq[db]);
      }
    }
    if (q < S) {
      T* out = (T*)a.dqkv + ((size_t)b * S + q) * ld + h * HD;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        short4v pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk[r] = Half16<T>::from_f(dq[db][r]);
        *reinterpret_cast<short4v*>(out + db * 16 + 4 * g) = pk;
      }
    }
    if (dbias) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = rows16(dq[db][r]);
          if (lm == 0) atomicAdd(cs + db * 16 + 4 * g + r, v);
        }
    }
  }

  // ---- dK, dV for keys r0..r0+15 (keys at or beyond kv_len receive exactly zero gradient)
  {
    const int key = r0 + lm;
    float4v dk_[4], dv_[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk_[db] = float4v{0.f, 0.f, 0.f, 0.f}; dv_[db] = float4v{0.f, 0.f, 0.f, 0.f}; }
    if (r0 < kvl) {
      const short8 kf0 = row_frag(kimg, r0, 0, lane), kf1 = row_frag(kimg, r0, 1, lane);
      const short8 vf0 = row_frag(vimg, r0, 0, lane), vf1 = row_frag(vimg, r0, 1, lane);
      const float mb = mbias[key];
      const uint32_t kmul = (uint32_t)key * ATT_GOLD;
#pragma unroll
      for (int qb = 0; qb < NPB; ++qb) {
        if (qb < nqb) {
          float4v pp[2], ds[2];
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            const int q16 = qb * 32 + hb * 16;
            float4v sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            sc = Half16<T>::mfma(row_frag(qimg, q16, 0, lane), kf0, sc);
            sc = Half16<T>::mfma(row_frag(qimg, q16, 1, lane), kf1, sc);
            dp = Half16<T>::mfma(row_frag(dimg, q16, 0, lane), vf0, dp);
            dp = Half16<T>::mfma(row_frag(dimg, q16, 1, lane), vf1, dp);
            const float4v ls = *reinterpret_cast<const float4v*>(lse_s + q16 + 4 * g);
            const float4v de = *reinterpret_cast<const float4v*>(del_s + q16 + 4 * g);
            const uint4v sd = *reinterpret_cast<const uint4v*>(seed_s + q16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int q = q16 + 4 * g + r;
              float p = __expf(sc[r] * a.scale + mb - ls[r]);
              if (q >= S) p = 0.f;
              float d = dp[r], pd = p;
              if (a.drop_thr != 0) {
                const bool keep = attn_keep(sd[r], kmul, thr_hi);
                d = keep ? d * a.drop_scale : 0.f;
                pd = keep ? p * a.drop_scale : 0.f;
              }
              pp[hb][r] = pd;
              ds[hb][r] = p * (d - de[r]) * a.scale;
            }
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
    if (key < S) {
      T* outk = (T*)a.dqkv + ((size_t)b * S + key) * ld + a.H + h * HD;
      T* outv = outk + a.H;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        short4v pk, pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pk[r] = Half16<T>::from_f(dk_[db][r]); pv[r] = Half16<T>::from_f(dv_[db][r]); }
        *reinterpret_cast<short4v*>(outk + db * 16 + 4 * g) = pk;
        *reinterpret_cast<short4v*>(outv + db * 16 + 4 * g) = pv;
      }
    }
    if (dbias && r0 < kvl) {
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vk = rows16(key < S ? dk_[db][r] : 0.f), vv = rows16(key < S ? dv_[db][r] : 0.f);
          if (lm == 0) {
            atomicAdd(cs + 64 + db * 16 + 4 * g + r, vk);
            atomicAdd(cs + 128 + db * 16 + 4 * g + r, vv);
          }
        }
    }
  }
  if (dbias) {
    __syncthreads();
    if (tid < 192) atomicAdd(dbias + (tid >> 6) * a.H + h * HD + (tid & 63), cs[tid]);
  }
}

template <typename KernT>
static int launch_attn(KernT kern, const AttnArgs& a, int smax, hipStream_t stream) {
  const size_t lds = (size_t)2 * smax * 128 + (size_t)3 * smax * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(a.B * a.A, (a.S + ATT_ROWS - 1) / ATT_ROWS), dim3(ATT_THREADS), lds, stream, a);
  return (int)hipGetLastError();
}

template <typename T>
static int attn_dispatch(int which, const AttnArgs& a, hipStream_t s) {
#define PDT_CASE(SM)                                                              \
  if (a.S <= SM) {                                                                \
    if (which == 0) return launch_attn(attn_fwd_kernel<T, SM>, a, SM, s);         \
    if (which == 1) return launch_attn(attn_bwd_dq_kernel<T, SM>, a, SM, s);      \
    return launch_attn(attn_bwd_dkv_kernel<T, SM>, a, SM, s);                     \
  }
  PDT_CASE(128) PDT_CASE(256) PDT_CASE(384) PDT_CASE(512)
#undef PDT_CASE
  return -3;
}

}  // namespace pdt

using namespace pdt;

static int check_attn(int B, int S, int A) { return (B <= 0 || S <= 0 || S > 512 || A <= 0) ? -3 : 0; }

extern "C" int pdt_attn_fwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, void* ctx,
                            float* lse, int B, int S, int A, float drop_scale, uint32_t drop_thr, PdtRng rng,
                            void* stream) {
  if (int e = check_attn(B, S, A)) return e;
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.kv_len = kv_len; a.ctx = ctx; a.lse = lse;
  a.B = B; a.S = S; a.A = A; a.H = A * HD; a.scale = 0.125f;
  a.drop_scale = drop_scale; a.drop_thr = drop_thr; a.rng = rng;
  return dtype == PDT_F16 ? attn_dispatch<_Float16>(0, a, (hipStream_t)stream) : attn_dispatch<__bf16>(0, a, (hipStream_t)stream);
}

template <typename T>
static int launch_fused128(const AttnArgs& a, float* dbias, hipStream_t stream) {
  const size_t lds = (size_t)4 * 128 * 128 + (size_t)(4 * 128 + 192) * sizeof(float);
  auto kern = attn_bwd_fused128_kernel<T>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(a.B * a.A), dim3(ATT_THREADS), lds, stream, a, dbias);
  return (int)hipGetLastError();
}

extern "C" int pdt_attn_bwd_fuses_dbias(int S) { return S <= 128 ? 1 : 0; }

extern "C" int pdt_attn_bwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, const void* ctx,
                            const void* dctx, const float* lse, float* delta, void* dqkv, float* dbias, int B, int S,
                            int A, float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream) {
  if (int e = check_attn(B, S, A)) return e;
  if (dbias && !pdt_attn_bwd_fuses_dbias(S)) return -3;
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.kv_len = kv_len; a.octx = ctx; a.dctx = dctx; a.lse = const_cast<float*>(lse);
  a.delta = delta; a.dqkv = dqkv;
  a.B = B; a.S = S; a.A = A; a.H = A * HD; a.scale = 0.125f;
  a.drop_scale = drop_scale; a.drop_thr = drop_thr; a.rng = rng;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 128) return dtype == PDT_F16 ? launch_fused128<_Float16>(a, dbias, s) : launch_fused128<__bf16>(a, dbias, s);
  int e = dtype == PDT_F16 ? attn_dispatch<_Float16>(1, a, s) : attn_dispatch<__bf16>(1, a, s);
  if (e) return e;
  return dtype == PDT_F16 ? attn_dispatch<_Float16>(2, a, s) : attn_dispatch<__bf16>(2, a, s);
}
