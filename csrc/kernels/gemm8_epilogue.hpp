// Epilogue of the 256x256 ping-pong GEMM for output tiles that lie completely inside the matrix (every tile of a
// persistent launch, and all but the edge tiles otherwise).  The general epilogue (GemmKernel::staged_epilogue) checks
// every row against the matrix edge, which puts every global access inside its own exec-masked branch: the compiler
// then waits for ALL outstanding memory operations at every join (one store round trip per row group), rebuilds 64-bit
// addresses per access and re-reads its arguments per row group.  Measured with in-kernel time stamps at 131072
// tokens (bench/micro/gemm8_ablate.hip): 4-12 us per tile next to a K loop of 25 us at K = 1024.
//
// This one is straight-line code per epilogue kind:
//   * global accesses go through buffer descriptors (bounds-checked by the hardware against the exact extent of the
//     tensor: a wrong offset is dropped, never a fault) with 32-bit offsets = one per-lane constant + one scalar;
//   * the side inputs (residual, gelu') are requested one row-block pair (4 loads per lane) ahead of their use;
//   * kinds whose math needs no side input (plain, bias, bias + GELU) do the math in the accumulator layout and
//     transpose 16-bit values through LDS (half the LDS bytes of the fp32 transpose: LDS stores run at ~80 B/clk per
//     CU, the slowest data path of an epilogue); the others transpose fp32 and work on rows, as before;
//   * the staging area is wave-private, so no barrier is involved and LDS operations of one wave execute in order: a
//     row read issued behind the transposed writes sees them without a wait in between.
// Results are bit-identical to the general epilogue for the fp32-staged kinds; the 16-bit-staged kinds round the same
// fp32 value once, so they are bit-identical too.
#pragma once
#include "gemm.hpp"

namespace pdt {

typedef uint32_t uint2v __attribute__((ext_vector_type(2)));

template <typename T, bool AKS, bool BKS>
struct FastEpilogue8 {
  // kinds with a fast path, per operand layout (forward: bias family; data gradient: GELU-grad multiply / residual add)
  static __device__ __forceinline__ bool supported(int epi) {
    if (epi == EPI_PLAIN) return true;
    if (!AKS && !BKS) return epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_BIAS_DROP_RES;
    if (!AKS && BKS) return epi == EPI_MUL_DGELU || epi == EPI_ADD_RES;
    return false;
  }

  static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, int rows, int ld, int cols) {
    // exact extent in bytes of a 16-bit [rows, ld] tensor whose rows hold `cols` values
    const uint32_t bytes = (uint32_t)(((int64_t)(rows - 1) * ld + cols) * 2);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
  }
  // first staged row of row-block pair ip inside the wave's rows: the wave owns rows [0,64) and [128,192) of its tile half
  static constexpr int pair_row(int ip) { return (ip >> 1) * 128 + (ip & 1) * 32; }

  static __device__ __forceinline__ uint2v pack4(const float4v& v) {
    short4v s;
    s[0] = Half16<T>::from_f(v[0]); s[1] = Half16<T>::from_f(v[1]);
    s[2] = Half16<T>::from_f(v[2]); s[3] = Half16<T>::from_f(v[3]);
    return __builtin_bit_cast(uint2v, s);
  }

  // ---- plain / bias / bias + GELU: math on the accumulators, 16-bit transpose ----------------------------------
  // Staging image of one row-block pair: 32 rows x 64 columns x 2 B, 128-B rows, 16-B chunks XOR-swizzled by row & 7, and
  // the two 8-B halves of every chunk swapped in rows 8-15 of each 16-row block.
  // Accumulator layout: lane (lm = lane & 15, lg = lane >> 4) holds columns lg*4 .. +3 of row lm of every 16 x 16 block.
  // Row layout: lane (rsub = lane >> 3, jc = lane & 7) holds the 8 columns of chunk jc of row it*8 + rsub.
  // The 16 lanes that write together (one lg, rows lm = 0 .. 15) hit 8 distinct chunks through the XOR; without the half
  // swap rows lm and lm + 8 would write the same 8 bytes' banks (a 2-way conflict on every transposed write: measured
  // as 7-12 % of the kernel's LDS cycles), with it the 16 lanes cover all 32 banks once.  A row read fetches whole
  // chunks, so the reader just swaps the halves back in registers - for odd `it`, known at compile time.
  template <int EPI>
  static __device__ __forceinline__ void run16(const PdtGemm& g, float4v (&acc)[8][4], char* stg, int m_w, int n_w, int lane) {
    constexpr bool has_bias = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU;
    constexpr bool gelu = EPI == EPI_BIAS_GELU;
    const int lm = lane & 15, lg = lane >> 4, rsub = lane >> 3, jc = lane & 7;
    float4v bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bias4[j] = float4v{0.f, 0.f, 0.f, 0.f};
      if (has_bias) {
        const short4v b = *reinterpret_cast<const short4v*>((const T*)g.bias + n_w + (j >> 1) * 128 + (j & 1) * 16 + lg * 4);
        bias4[j] = float4v{Half16<T>::to_f(b[0]), Half16<T>::to_f(b[1]), Half16<T>::to_f(b[2]), Half16<T>::to_f(b[3])};
      }
    }
    const int key_w = (lg >> 1) ^ (lm & 7);
    char* wbase = stg + lm * 128 + ((lg & 1) ^ (lm >> 3)) * 8;
    const char* rbase = stg + rsub * 128 + ((jc ^ rsub) << 4);
    const __amdgpu_buffer_rsrc_t rc = rsrc_of(g.C, g.M, g.ldc, g.N);
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(gelu ? g.aux : g.C, g.M, gelu ? g.ldaux : g.ldc, g.N);
    const uint32_t col_l = (uint32_t)((jc >> 2) * 128 + (jc & 3) * 8);
    const uint32_t lane_c = ((uint32_t)rsub * g.ldc + col_l) * 2u, lane_a = ((uint32_t)rsub * g.ldaux + col_l) * 2u;
    const uint32_t base_c = ((uint32_t)m_w * g.ldc + n_w) * 2u, base_a = ((uint32_t)m_w * g.ldaux + n_w) * 2u;
    const uint32_t step_c = (uint32_t)g.ldc * 2u, step_a = (uint32_t)g.ldaux * 2u;
    // without GELU the two 4-KiB halves of the wave's staging area alternate between row-block pairs (the writes of
    // pair ip+1 do not have to queue behind the reads of pair ip); with GELU they hold the activation and its derivative
    auto stage_write = [&](int ip) {
      char* w = wbase + ((!gelu && (ip & 1)) ? 4096 : 0);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          float4v v0 = acc[ip * 2 + h][j] + bias4[j], v1 = acc[ip * 2 + h][j + 1] + bias4[j + 1];
          char* d0 = w + h * 2048 + (((j * 2) ^ key_w) << 4);
          char* d1 = w + h * 2048 + ((((j + 1) * 2) ^ key_w) << 4);
          if (gelu) {
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}, d[8];
            gelu8(v, d);
            *reinterpret_cast<uint2v*>(d0 + 4096) = pack4(float4v{d[0], d[1], d[2], d[3]});
            *reinterpret_cast<uint2v*>(d1 + 4096) = pack4(float4v{d[4], d[5], d[6], d[7]});
            v0 = float4v{v[0], v[1], v[2], v[3]};
            v1 = float4v{v[4], v[5], v[6], v[7]};
          }
          *reinterpret_cast<uint2v*>(d0) = pack4(v0);
          *reinterpret_cast<uint2v*>(d1) = pack4(v1);
        }
    };
    auto stage_store = [&](int ip) {
      const char* r = rbase + ((!gelu && (ip & 1)) ? 4096 : 0);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        uint4v c = *reinterpret_cast<const uint4v*>(r + it * 1024);
        if (it & 1) c = uint4v{c[2], c[3], c[0], c[1]};  // rows 8-15 of a block: halves were stored swapped
        const uint32_t row = (uint32_t)(pair_row(ip) + it * 8);
        __builtin_amdgcn_raw_buffer_store_b128(c, rc, (int)(lane_c + base_c + row * step_c), 0, 0);
        if (gelu) {
          // gelu'(pre-activation): not read again before the backward pass, keep it out of the caches
          uint4v a = *reinterpret_cast<const uint4v*>(r + 4096 + it * 1024);
          if (it & 1) a = uint4v{a[2], a[3], a[0], a[1]};
          __builtin_amdgcn_raw_buffer_store_b128(a, ra, (int)(lane_a + base_a + row * step_a), 0, 2);
        }
      }
    };
    stage_write(0);
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      if (!gelu && ip + 1 < 4) stage_write(ip + 1);
      stage_store(ip);
      if (gelu && ip + 1 < 4) stage_write(ip + 1);
    }
  }

  // ---- bias + dropout + residual / GELU-grad multiply (+ column sums) / residual add: fp32 transpose, math on rows ----
  // KEYMODE as in GemmKernel::staged_epilogue_as: how a row's dropout row is found (0: the row index, 1: row + an offset
  // per 32-row block from one scalar sample-id load, 2: a lookup per row); -1: no dropout (a compile-time property: the
  // mask generator's constants would otherwise occupy registers in every variant)
  template <int EPI, int KEYMODE>
  static __device__ __forceinline__ void run32(const PdtGemm& g, float4v (&acc)[8][4], char* stg, int m_w, int n_w, int prow, int lane) {
    constexpr bool dgelu = EPI == EPI_MUL_DGELU;
    constexpr bool drop = EPI == EPI_BIAS_DROP_RES;
    const int lm = lane & 15, lg = lane >> 4, rsub = lane >> 3, jc = lane & 7;
    const int ldx = dgelu ? g.ldaux : g.ldr;
    const __amdgpu_buffer_rsrc_t rc = rsrc_of(g.C, g.M, g.ldc, g.N);
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(dgelu ? g.aux : g.res, g.M, ldx, g.N);
    const uint32_t col_l = (uint32_t)((jc >> 2) * 128 + (jc & 3) * 8);
    const int n = n_w + (int)col_l;  // first of the lane's 8 columns in row layout
    const uint32_t lane_c = ((uint32_t)rsub * g.ldc + col_l) * 2u, lane_x = ((uint32_t)rsub * ldx + col_l) * 2u;
    const uint32_t base_c = ((uint32_t)m_w * g.ldc + n_w) * 2u, base_x = ((uint32_t)m_w * ldx + n_w) * 2u;
    const uint32_t step_c = (uint32_t)g.ldc * 2u, step_x = (uint32_t)ldx * 2u;
    // side inputs: the loads of the next row-block pair (4 x 16 B per lane) are in flight under the math of this one
    // (two pairs ahead, the compiler ran out of registers in the dropout variants and spilled the loaded values)
    uint4v xin[4][4];
    auto request = [&](int ip) {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        xin[ip][it] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(lane_x + base_x + (uint32_t)(pair_row(ip) + it * 8) * step_x), 0, 0);
    };
    request(0);
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (drop) unpack8<T>(*reinterpret_cast<const short8*>((const T*)g.bias + n), bias);
    const DropKey dk = drop_key(g.rng);
    constexpr bool dropping = drop && KEYMODE >= 0;
    uint32_t drow_off[4] = {0u, 0u, 0u, 0u};  // (wave-uniform)
    if (drop && KEYMODE == 1) {
#pragma unroll
      for (int ip = 0; ip < 4; ++ip) {
        const uint32_t m32 = (uint32_t)__builtin_amdgcn_readfirstlane(m_w + pair_row(ip));
        const uint32_t b = __umulhi(m32, g.rng.seq_magic);
        drow_off[ip] = (uint32_t)uniform_load_i32(g.rng.sample_ids, b) * 512u - b * g.rng.seq;
      }
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    char* wbase = stg + lm * 256;
    const char* rbase = stg + rsub * 256;
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4v*>(wbase + h * 4096 + (((j * 4 + lg) ^ lm) << 4)) = acc[ip * 2 + h][j];
      float4v lo[4], hi[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int key = (it * 8 + rsub) & 15;
        lo[it] = *reinterpret_cast<const float4v*>(rbase + it * 2048 + (((2 * jc) ^ key) << 4));
        hi[it] = *reinterpret_cast<const float4v*>(rbase + it * 2048 + (((2 * jc + 1) ^ key) << 4));
      }
      if (ip + 1 < 4) request(ip + 1);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t row = (uint32_t)(pair_row(ip) + it * 8);
        const int m = m_w + (int)row + rsub;
        float v[8] = {lo[it][0], lo[it][1], lo[it][2], lo[it][3], hi[it][0], hi[it][1], hi[it][2], hi[it][3]};
        float x[8];
        unpack8<T>(__builtin_bit_cast(short8, xin[ip][it]), x);
        if (drop) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += bias[i];
          if (dropping) {
            uint32_t drow = (uint32_t)m;
            if (KEYMODE == 1) drow += drow_off[ip];
            if (KEYMODE == 2) drow = drop_row(g.rng, (uint32_t)m);
            const uint32_t keep = drop_keep8(dk, ((uint64_t)drow * g.N + n) >> 3, g.drop_thr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ((keep >> i) & 1u) ? v[i] * g.drop_scale : 0.0f;
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = dgelu ? v[i] * x[i] : v[i] + x[i];
        const short8 packed = pack8<T>(v);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, packed), rc, (int)(lane_c + base_c + row * step_c), 0, 0);
        if (g.colsum) {  // column sums are taken over exactly what was stored
          unpack8<T>(packed, v);
#pragma unroll
          for (int i = 0; i < 8; ++i) cs[i] += v[i];
        }
      }
    }
    if (g.colsum) GemmKernel<T, 256, 256, 2, 4, AKS, BKS>::flush_colsum(g, prow, n, true, cs, lane);
  }

  // m_w / n_w: first row / column of the wave inside the output (tile origin + wave row * 64 / wave column * 32);
  // stg: the wave's private 8 KiB of LDS; prow: row of the partial column-sum matrix (tile row * 2 + wave row)
  static __device__ __forceinline__ void run(const PdtGemm& g, float4v (&acc)[8][4], char* stg, int m_w, int n_w, int prow, int lane) {
    if (!AKS && !BKS) {
      switch (g.epi) {
        case EPI_BIAS: return run16<EPI_BIAS>(g, acc, stg, m_w, n_w, lane);
        case EPI_BIAS_GELU: return run16<EPI_BIAS_GELU>(g, acc, stg, m_w, n_w, lane);
        case EPI_BIAS_DROP_RES: {
          if (g.drop_thr == 0) return run32<EPI_BIAS_DROP_RES, -1>(g, acc, stg, m_w, n_w, prow, lane);
          const bool keyed = g.rng.sample_ids != nullptr || g.rng.row_table != nullptr;
          if (!keyed) return run32<EPI_BIAS_DROP_RES, 0>(g, acc, stg, m_w, n_w, prow, lane);
          if (g.rng.row_table == nullptr && (g.rng.seq & 31u) == 0) return run32<EPI_BIAS_DROP_RES, 1>(g, acc, stg, m_w, n_w, prow, lane);
          return run32<EPI_BIAS_DROP_RES, 2>(g, acc, stg, m_w, n_w, prow, lane);
        }
        default: break;
      }
    } else if (!AKS && BKS) {
      switch (g.epi) {
        case EPI_MUL_DGELU: return run32<EPI_MUL_DGELU, -1>(g, acc, stg, m_w, n_w, prow, lane);
        case EPI_ADD_RES: return run32<EPI_ADD_RES, -1>(g, acc, stg, m_w, n_w, prow, lane);
        default: break;
      }
    }
    run16<EPI_PLAIN>(g, acc, stg, m_w, n_w, lane);
  }
};

}  // namespace pdt
