// MFMA GEMM for gfx950: C[M,N] = sum_k A(m,k) * B(n,k) with fused epilogues.
//
// Operand storage (per operand, compile-time):
//   K-contiguous : X stored [rows, ld], element (x,k) at X[x*ld + k]   (activations, Linear weights in fwd)
//   K-strided    : X stored [K, ld],    element (x,k) at X[k*ld + x]   (weights in dgrad, both operands in wgrad)
// so forward (X*W^T), dgrad (dY*W) and wgrad (dY^T*X) all run on this one kernel with no transposed copies.
//
// Structure: BK=64 K-tiles, direct-to-LDS loads (buffer_load ... lds, 16 B/lane) into two LDS stages,
// one barrier per K-tile; 16x16x32 MFMA; XOR-swizzled LDS images so that ds_read_b128 (K-contiguous)
// and ds_read_b64_tr_b16 (K-strided) fragment reads are bank-conflict free; the swizzle is applied on the
// per-lane SOURCE address because an LDS-DMA write is lane-linear.  The accumulator tile leaves through a
// wave-private fp32 LDS transpose so that every global store / residual load is 16 B per lane on full
// 128-B row segments.
#pragma once
#include "common.hpp"

// PDT_EPI_ABLATE (bench/micro/gemm8_ablate.hip only; results are wrong by construction): bit 0 = no global stores,
// bit 1 = no side-input loads, bit 2 = no LDS transpose of the accumulators
#ifndef PDT_EPI_ABLATE
#define PDT_EPI_ABLATE 0
#endif

namespace pdt {

enum GemmEpilogue : int {
  EPI_PLAIN = PDT_EPI_PLAIN,                  // C = acc
  EPI_BIAS = PDT_EPI_BIAS,                    // C = acc + bias[n]
  EPI_BIAS_GELU = PDT_EPI_BIAS_GELU,          // u = acc + bias[n] ; C = gelu(u) ; aux = gelu'(u)
  EPI_BIAS_DROP_RES = PDT_EPI_BIAS_DROP_RES,  // C = dropout(acc + bias[n]) + res[m,n]
  EPI_MUL_DGELU = PDT_EPI_MUL_DGELU,          // C = acc * aux[m,n]   (aux = gelu'(u) saved by BIAS_GELU)
  EPI_ADD_RES = PDT_EPI_ADD_RES,              // C = acc + res[m,n]
  EPI_SLAB_F32 = PDT_EPI_SLAB_F32,            // split-K partial: fp32 slab[split][m,n] = acc
  EPI_BIAS_TANH = PDT_EPI_BIAS_TANH,          // C = tanh(acc + bias[n])
};
using GemmArgs = PdtGemm;

// STAGES: depth of the LDS ring of K-tiles.  2 = double buffer (the next K-tile is fetched while this one is
// multiplied; small enough for two workgroups per CU with 128x128 tiles).  4 = three K-tiles in flight and one counted
// wait per K-tile: for launches with about one workgroup per CU or fewer (short micro-batches), where nothing else
// on the CU hides the ~2 us a K-tile takes from HBM / L2 to LDS.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool AKS, bool BKS, int STAGES = 2>
struct GemmKernel {
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int BK = 64;
  static constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  static constexpr int FM = WTM / 16, FN = WTN / 16;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
  static_assert(STAGES == 2 || STAGES == 4, "ring depth");
  static constexpr int A_PASSES = A_BYTES / (NT * 16), B_PASSES = B_BYTES / (NT * 16);
  // (the staged epilogue assumes 64 output columns per wave: other wave shapes only write fp32 slabs)
  static_assert(A_BYTES % (NT * 16) == 0 && B_BYTES % (NT * 16) == 0, "tile/threads mismatch");
  static_assert(WAVES_M * WAVES_N * 32 * 256 <= LDS_BYTES, "epilogue staging does not fit");

  // swizzle key of a K-strided image row k (32-B chunk index XOR): distinct for the 8 rows one
  // half-wave touches in a transposed read {4 consecutive k} x {two k-groups 8 apart}
  static __device__ __forceinline__ int ks_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

  // ---- per-lane source offsets for one operand -------------------------------------------------
  template <bool KS, int BX, int PASSES>
  struct Loader {
    uint32_t off[PASSES];    // byte offset of this lane's 16 B inside the operand's buffer window
    uint32_t kcol[PASSES];   // (K-contiguous only) k index of the chunk inside the K-tile
    int4v rsrc;
    uint32_t kstep;          // bytes to advance per K-tile

    __device__ __forceinline__ void init(const T* X, int ld, int xdim, int x0, int kb, int klen, int tid) {
      const T* base;
      uint32_t records;
      const int xavail = min(BX, xdim - x0);
      if (KS) {
        base = X + (size_t)kb * ld + x0;
        records = (uint32_t)(((size_t)(klen - 1) * ld + xavail) * 2);
        kstep = (uint32_t)ld * 2u * BK;
      } else {
        base = X + (size_t)x0 * ld + kb;
        records = (uint32_t)(((size_t)(xavail - 1) * ld + klen) * 2);
        kstep = BK * 2;
      }
      if (xavail <= 0) {  // the whole window lies outside the operand: every load must read as zero
        base = X;
        records = 0;
      }
      rsrc = pdt_make_rsrc(base, records);
      offsets(ld, xavail, tid);
    }
    // per-lane offsets of the window (they depend on the operand's pitch and on how much of the window is inside
    // the matrix, not on where the window starts)
    __device__ __forceinline__ void offsets(int ld, int xavail, int tid) {
      constexpr int pitch = BX * 2;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int pos = p * NT * 16 + tid * 16;
        if (KS) {
          const int k = pos / pitch, w = pos % pitch;
          const int chunk = (w >> 5) ^ ks_key(k);
          const int col = chunk * 16 + ((w >> 4) & 1) * 8;
          const bool ok = (col + 8 <= xavail);
          off[p] = ok ? (uint32_t)k * ld * 2u + col * 2u : 0x80000000u;
          kcol[p] = 0;
        } else {
          const int row = pos >> 7, cpos = (pos >> 4) & 7;
          const int c = cpos ^ (row & 7);
          const bool ok = row < xavail;
          off[p] = ok ? (uint32_t)row * ld * 2u + c * 16u : 0x80000000u;
          kcol[p] = c * 8;
        }
      }
    }
    // point the loader at another FULL window of the same operand (same lane offsets, new base): what a
    // persistent kernel does between output tiles
    __device__ __forceinline__ void retarget(const T* X, int ld, int x0, int kb, int klen) {
      const T* base = KS ? X + (size_t)kb * ld + x0 : X + (size_t)x0 * ld + kb;
      const uint32_t records = KS ? (uint32_t)(((size_t)(klen - 1) * ld + BX) * 2) : (uint32_t)(((size_t)(BX - 1) * ld + klen) * 2);
      rsrc = pdt_make_rsrc(base, records);
    }
    // L2 prefetch: one dword per lane from 64 different 128-B lines of K-tile `kt` of this window - K-contiguous: rows
    // part*64 + lane; K-strided: the 64 k-rows, line `part` (columns part*64 ..) of each.  The 256 B land at `junk`.
    // Offsets past the window are dropped by the buffer unit (no access).
    __device__ __forceinline__ void touch(char* junk, int kt, int part, int ld, int lane) {
      const uint32_t o = KS ? (uint32_t)lane * (uint32_t)ld * 2u + (uint32_t)part * 128u + (uint32_t)kt * kstep
                            : (uint32_t)(part * 64 + lane) * (uint32_t)ld * 2u + (uint32_t)kt * kstep;
      pdt_lds_dma4(rsrc, junk, o);
    }
    // issue this operand's LDS-DMA loads for K-tile `kt` into `dst` (wave-uniform LDS base of the image)
    __device__ __forceinline__ void issue(char* dst, int kt, int klen, int wave) {
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        uint32_t o = off[p] + (uint32_t)kt * kstep;
        if (!KS) o = ((int)(kcol[p] + kt * BK) < klen) ? o : 0x80000000u;
        pdt_lds_dma16(rsrc, dst + p * NT * 16 + wave * 1024, o);
      }
    }
  };

  // ---- fragment reads ----------------------------------------------------------------------------
  // K-contiguous image [BX][64] (128-B rows): 8 k-values of row x0+(lane&15), k-group (lane>>4) of k-step ks
  static __device__ __forceinline__ short8 frag_kc(const char* img, int x0, int ks, int lane) {
    const int row = x0 + (lane & 15);
    const int chunk = (ks * 4 + (lane >> 4)) ^ (row & 7);
    return *reinterpret_cast<const short8*>(img + row * 128 + chunk * 16);
  }
  // K-strided image [64][BX]: two transposed 4x16 reads give the same 8 k-values for column x0+(lane&15)
  template <int BX>
  static __device__ __forceinline__ short8 frag_ks(const char* img, int x0, int ks, int lane) {
    constexpr int pitch = BX * 2;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int key = q | ((g & 1) << 2);
    const int chunk = (x0 >> 4) ^ key;
    const int k = ks * 32 + g * 8 + q;
    const char* a = img + k * pitch + chunk * 32 + p * 8;
    short4r lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4r __attribute__((address_space(3)))*)PDT_LDS_PTR(a));
    short4r hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4r __attribute__((address_space(3)))*)PDT_LDS_PTR(a + 4 * pitch));
    short8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  // Math on 8 consecutive columns of one row.  `bias` holds the 8 bias values of these columns (zeros when the
  // epilogue has none), `ext` the 8 side-input values of this row: aux for MUL_DGELU, res for *_RES.
  // `drow`: the row the dropout counters of row m derive from (drop_row(g.rng, m); only read by BIAS_DROP_RES)
  template <int epi>
  static __device__ __forceinline__ void epilogue8(const GemmArgs& g, int m, int n, float (&v)[8], const float (&bias)[8],
                                                   const short8& ext, const DropKey& dk, uint32_t drow) {
    if (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_TANH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += bias[i];
    }
    if (epi == EPI_BIAS_GELU) {
      // aux = gelu'(pre-activation): all the backward pass needs of it (MUL_DGELU is then a plain multiply, no
      // exp / rcp in the data-gradient epilogue).  Not read again before backward: keep it out of the caches.
      float d[8];
      gelu8(v, d);
      if (PDT_EPI_ABLATE & 1) { const short8 pd = pack8<T>(d); asm volatile("" :: "v"(pd)); }
      else __builtin_nontemporal_store(pack8<T>(d), reinterpret_cast<short8*>((T*)g.aux + (size_t)m * g.ldaux + n));
    } else if (epi == EPI_BIAS_TANH) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = tanhf(v[i]);
    } else if (epi == EPI_MUL_DGELU) {
      float d[8];
      unpack8<T>(ext, d);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= d[i];
    }
    if (epi == EPI_BIAS_DROP_RES && g.drop_thr != 0) {
      const uint32_t keep = drop_keep8(dk, ((uint64_t)drow * g.N + n) >> 3, g.drop_thr);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = ((keep >> i) & 1u) ? v[i] * g.drop_scale : 0.0f;
    }
    if (epi == EPI_BIAS_DROP_RES || epi == EPI_ADD_RES) {
      float r[8];
      unpack8<T>(ext, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += r[i];
    }
    const short8 packed = pack8<T>(v);
    if (PDT_EPI_ABLATE & 1) asm volatile("" :: "v"(packed));
    else *reinterpret_cast<short8*>((T*)g.C + (size_t)m * g.ldc + n) = packed;
    if (g.colsum) unpack8<T>(packed, v);  // column sums are taken over exactly what was stored
  }

  // store the 8 per-column partial sums a lane collected (over the rows it handled) into row `prow` of the
  // partial-sum matrix g.colsum: the 8 lanes that share a column block (lane>>3 = row inside a pass) are
  // combined first, then one of them stores 32 bytes
  static __device__ __forceinline__ void flush_colsum(const GemmArgs& g, int prow, int n, bool n_ok, float (&cs)[8], int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = cs[i];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      cs[i] = v;
    }
    if ((lane >> 3) == 0 && n_ok) {
      float* dst = g.colsum + (size_t)prow * g.N + n;
      *reinterpret_cast<float4v*>(dst) = float4v{cs[0], cs[1], cs[2], cs[3]};
      *reinterpret_cast<float4v*>(dst + 4) = float4v{cs[4], cs[5], cs[6], cs[7]};
    }
  }

  // The accumulators of a wave (2*NIP row blocks of 16 x 64 columns) leave through a wave-private 32-row x
  // 64-col fp32 LDS staging area (16-B chunks XOR-swizzled by row&15), one pair of row blocks ("ip") at a time,
  // so that every global access is 16 B per lane on full 128-B row segments.  Lane -> (row it*8 + lane>>3,
  // column block lane&7); `n` is the first of the lane's 8 global columns (the same for every row) and
  // mbase(ip) the global row of staged row 0.  The side inputs (aux / res) of row-block pair ip+1 are fetched
  // before pair ip is processed, so their latency hides under the math and the LDS transpose; the bias is
  // read once; the column sums are kept in registers over the whole wave tile and stored once, into row `prow`
  // (tile row * 2 + wave row) of the partial-sum matrix.
  // KEYMODE (BIAS_DROP_RES only): how a row's dropout row is found - 0: the row index itself (no sample ids, or
  // dropout off), 1: row + a per-block offset from ONE scalar sample-id load (sequence length a multiple of 32),
  // 2: a lookup per row (unpadded batches, other lengths).  A compile-time property of the copy: with the three ways
  // behind run-time branches of one copy, the vector loads of mode 2 left `s_waitcnt vmcnt(0)` at the joins of every
  // row iteration - which, executed on the paths of modes 0 and 1, waited for the output stores just issued, 16
  // store round trips per tile.
  template <int NIP, int epi, typename MBase, int KEYMODE = 0>
  static __device__ __forceinline__ void staged_epilogue_as(const GemmArgs& g, float4v (&acc)[2 * NIP][4], char* stg, int n,
                                                            MBase mbase, int prow, int lane) {
    const int lm = lane & 15, lg = lane >> 4, jc = lane & 7, rsub = lane >> 3;
    const bool n_ok = n < g.N;
    const DropKey dk = drop_key(g.rng);
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_TANH) && n_ok)
      unpack8<T>(*reinterpret_cast<const short8*>((const T*)g.bias + n), bias);
    const bool dgelu = epi == EPI_MUL_DGELU;
    const bool has_ext = dgelu || epi == EPI_BIAS_DROP_RES || epi == EPI_ADD_RES;
    const T* ext = dgelu ? (const T*)g.aux : (const T*)g.res;
    const int ldx = dgelu ? g.ldaux : g.ldr;
    short8 xin[2][4];
    // sample-keyed dropout (PdtRng::sample_ids): when the sequence length is a multiple of 32 the 32 rows of a staged
    // block pair belong to ONE sample, so the id lookup is wave-uniform (a scalar load) and a row's dropout row is
    // m + offset; other lengths look the id up per row
    constexpr bool keyed = epi == EPI_BIAS_DROP_RES && KEYMODE != 0;
    constexpr bool keyed_uniform = epi == EPI_BIAS_DROP_RES && KEYMODE == 1;
    auto fetch = [&](int ip, short8(&dst)[4]) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int m = mbase(ip) + it * 8 + rsub;
        dst[it] = short8{0, 0, 0, 0, 0, 0, 0, 0};
        if (!(PDT_EPI_ABLATE & 2) && has_ext && n_ok && m < g.M) dst[it] = *reinterpret_cast<const short8*>(ext + (size_t)m * ldx + n);
      }
    };
    fetch(0, xin[0]);
    uint32_t drow_offs[NIP];  // (wave-uniform: the id loads are issued together, ahead of the loop that uses them)
#pragma unroll
    for (int ip = 0; ip < NIP; ++ip) {
      drow_offs[ip] = 0;
      if (keyed_uniform) {
        const uint32_t m32 = (uint32_t)__builtin_amdgcn_readfirstlane(mbase(ip));
        if (m32 < (uint32_t)g.M) {
          const uint32_t b = __umulhi(m32, g.rng.seq_magic);
          drow_offs[ip] = (uint32_t)uniform_load_i32(g.rng.sample_ids, b) * 512u - b * g.rng.seq;
        }
      }
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ip = 0; ip < NIP; ++ip) {
      if (ip + 1 < NIP) fetch(ip + 1, xin[(ip + 1) & 1]);
      const uint32_t drow_off = drow_offs[ip];
      if (!(PDT_EPI_ABLATE & 4)) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 16 + lm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int chunk = (j * 4 + lg) ^ lm;
          *reinterpret_cast<float4v*>(stg + row * 256 + chunk * 16) = acc[ip * 2 + h][j];
        }
      }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        const int key = row & 15;
        const float4v lo = (PDT_EPI_ABLATE & 4) ? acc[ip * 2 + (it >> 1)][(it & 1) * 2] : *reinterpret_cast<const float4v*>(stg + row * 256 + (((2 * jc) ^ key) << 4));
        const float4v hi = (PDT_EPI_ABLATE & 4) ? acc[ip * 2 + (it >> 1)][(it & 1) * 2 + 1] : *reinterpret_cast<const float4v*>(stg + row * 256 + (((2 * jc + 1) ^ key) << 4));
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int m = mbase(ip) + row;
        if (m < g.M && n_ok) {
          uint32_t drow = (uint32_t)m;
          if (epi == EPI_BIAS_DROP_RES && keyed) drow = keyed_uniform ? (uint32_t)m + drow_off : drop_row(g.rng, m);
          epilogue8<epi>(g, m, n, v, bias, xin[ip & 1][it], dk, drow);
          if (g.colsum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[i] += v[i];
          }
        }
      }
    }
    if (g.colsum) flush_colsum(g, prow, n, n_ok, cs, lane);
  }

  // The epilogue kind is a run-time argument of the kernel but a compile-time parameter of the (fully unrolled)
  // epilogue code: one specialised copy per kind the operand layout is used with, selected by one uniform
  // branch, so a tile only ever fetches the instructions of its own epilogue.
  //   NT (forward):        PLAIN, BIAS, BIAS_GELU, BIAS_DROP_RES, BIAS_TANH
  //   NN (data gradient):  PLAIN, MUL_DGELU, ADD_RES
  //   TN (weight gradient) only ever writes split-K slabs (handled by the caller); PLAIN is kept for tests.
  static constexpr bool epilogue_supported(int epi) {
    return epi == EPI_PLAIN || epi == EPI_SLAB_F32 ||
           (!AKS && !BKS && (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_DROP_RES || epi == EPI_BIAS_TANH)) ||
           (!AKS && BKS && (epi == EPI_MUL_DGELU || epi == EPI_ADD_RES));
  }
  template <int NIP, typename MBase>
  static __device__ __forceinline__ void staged_epilogue(const GemmArgs& g, float4v (&acc)[2 * NIP][4], char* stg, int n,
                                                         MBase mbase, int prow, int lane) {
    if (!AKS && !BKS) {
      switch (g.epi) {
        case EPI_BIAS: return staged_epilogue_as<NIP, EPI_BIAS>(g, acc, stg, n, mbase, prow, lane);
        case EPI_BIAS_GELU: return staged_epilogue_as<NIP, EPI_BIAS_GELU>(g, acc, stg, n, mbase, prow, lane);
        case EPI_BIAS_DROP_RES: {
          const bool keyed = g.drop_thr != 0 && (g.rng.sample_ids != nullptr || g.rng.row_table != nullptr);
          if (!keyed) return staged_epilogue_as<NIP, EPI_BIAS_DROP_RES, MBase, 0>(g, acc, stg, n, mbase, prow, lane);
          if (g.rng.row_table == nullptr && (g.rng.seq & 31u) == 0)
            return staged_epilogue_as<NIP, EPI_BIAS_DROP_RES, MBase, 1>(g, acc, stg, n, mbase, prow, lane);
          return staged_epilogue_as<NIP, EPI_BIAS_DROP_RES, MBase, 2>(g, acc, stg, n, mbase, prow, lane);
        }
        case EPI_BIAS_TANH: return staged_epilogue_as<NIP, EPI_BIAS_TANH>(g, acc, stg, n, mbase, prow, lane);
        default: break;
      }
    } else if (!AKS && BKS) {
      switch (g.epi) {
        case EPI_MUL_DGELU: return staged_epilogue_as<NIP, EPI_MUL_DGELU>(g, acc, stg, n, mbase, prow, lane);
        case EPI_ADD_RES: return staged_epilogue_as<NIP, EPI_ADD_RES>(g, acc, stg, n, mbase, prow, lane);
        default: break;
      }
    }
    staged_epilogue_as<NIP, EPI_PLAIN>(g, acc, stg, n, mbase, prow, lane);
  }

  static __device__ __forceinline__ void run(const GemmArgs& g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int ntiles = g.tiles_m * g.tiles_n;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles, t = id % ntiles;
    int tm, tn;
    tile_coords(t, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kb = split * g.k_per_split;
    const int klen = min(g.K - kb, g.k_per_split);
    const int nkt = (klen + BK - 1) / BK;

    Loader<AKS, BM, A_PASSES> la;
    Loader<BKS, BN, B_PASSES> lb;
    la.init((const T*)g.A, g.lda, g.M, m0, kb, klen, tid);
    lb.init((const T*)g.B, g.ldb, g.N, n0, kb, klen, tid);

    float4v acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](const char* st) {  // one K-tile from the ring slot `st`
      const char* ia = st;
      const char* ib = st + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        short8 fa[FM], fb[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j)
          fb[j] = BKS ? frag_ks<BN>(ib, wn * WTN + j * 16, ks, lane) : frag_kc(ib, wn * WTN + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
          fa[i] = AKS ? frag_ks<BM>(ia, wm * WTM + i * 16, ks, lane) : frag_kc(ia, wm * WTM + i * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            // MFMA "A" operand = B-matrix rows (n), "B" operand = A-matrix rows (m):
            // D[n][m] => lane holds m = lane&15 and 4 consecutive n = 4*(lane>>4)+r
            acc[i][j] = Half16<T>::mfma(fb[j], fa[i], acc[i][j]);
      }
    };
    if (STAGES == 2) {
      la.issue(smem, 0, klen, wave);
      lb.issue(smem + A_BYTES, 0, klen, wave);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int cur = 0;
      for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) {
          char* nx = smem + (cur ^ 1) * STAGE_BYTES;
          la.issue(nx, kt + 1, klen, wave);
          lb.issue(nx + A_BYTES, kt + 1, klen, wave);
        }
        multiply(smem + cur * STAGE_BYTES);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
      }
    } else {
      // K-tiles kt .. kt+STAGES-2 are in flight when tile kt is needed.  Every K-tile costs a wave the same number of
      // DMA instructions - tiles past the end of K are issued too, with out-of-range offsets (they return zeros at
      // once) - so the wait for tile kt is always "all but the newest (STAGES-2) tiles' worth": one immediate count.
      constexpr int PER_TILE = A_PASSES + B_PASSES;
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s) {
        la.issue(smem + s * STAGE_BYTES, s, klen, wave);
        lb.issue(smem + s * STAGE_BYTES + A_BYTES, s, klen, wave);
      }
      for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_TILE) : "memory");
        __syncthreads();  // tile kt has landed for every wave, and every wave is done reading tile kt-1 ...
        char* nx = smem + ((kt + STAGES - 1) % STAGES) * STAGE_BYTES;  // ... whose slot takes tile kt+STAGES-1
        la.issue(nx, kt + STAGES - 1, klen, wave);
        lb.issue(nx + A_BYTES, kt + STAGES - 1, klen, wave);
        multiply(smem + (kt % STAGES) * STAGE_BYTES);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the trailing zero-fills: the epilogue reuses the ring)
      __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    const int lm = lane & 15, lg = lane >> 4;
    if (g.epi == EPI_SLAB_F32) {
      float* slab = (float*)g.C + (size_t)split * g.M * g.ldc;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + lm;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = n0 + wn * WTN + j * 16 + lg * 4;
          if (m < g.M && n < g.N) {
            float4v* dst = reinterpret_cast<float4v*>(slab + (size_t)m * g.ldc + n);
            *dst = g.slab_accumulate ? *dst + acc[i][j] : acc[i][j];
          }
        }
      }
      return;
    }
    if constexpr (FN == 4 && FM % 2 == 0 && WAVES_M == 2)  // (2 partial column-sum rows per tile row)
      staged_epilogue<FM / 2>(g, acc, smem + wave * (32 * 256), n0 + wn * WTN + (lane & 7) * 8,
                              [&](int ip) { return m0 + wm * WTM + ip * 32; }, tm * WAVES_M + wm, lane);
  }
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool AKS, bool BKS, int STAGES = 2>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_kernel(GemmArgs g) {
  GemmKernel<T, BM, BN, WAVES_M, WAVES_N, AKS, BKS, STAGES>::run(g);
}

}  // namespace pdt
