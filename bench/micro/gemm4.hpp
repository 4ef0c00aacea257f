// EXPERIMENT (bench/micro/gemm8_ablate.hip -DPDT_KERNEL4=1; not part of the library): 256x256x64 MFMA GEMM for gfx950
// with FOUR waves, one per SIMD, 128 x 128 outputs per wave.  Correct on every shape tried, and measured 4-8 % SLOWER
// than the 8-wave ping-pong kernel (K loop of 64 K-tiles: 101 vs 97 us; 16 K-tiles: 25.0 vs 23.0 us): with one wave per
// SIMD every fragment read (~15 cycles) and every LDS-DMA piece (~28 cycles) that does not fit into the 8 issue cycles an
// MFMA leaves free extends the loop - 3040 cycles per K-tile against 2600 - and on a power-limited chip the extra
// cycles have to be bought with a higher clock and voltage (1.93 against 1.71 GHz), which costs more energy than the
// smaller fragment traffic saves.  v_mfma_f32_32x32x16 (longer gaps to fill) needs 7 % fewer cycles but holds 1.68 GHz:
// slower again.  Leaving barriers staggered by 8 or 16 cycles per wave: slower (the reads do not queue behind each other).
//
// Why a second large-tile kernel.  Under these GEMMs the chip is power-limited, not cycle-limited: in-kernel stamps
// (bench/micro/gemm8_ablate.hip) show the K loop of the 8-wave ping-pong kernel taking the same ~2700 shader cycles per
// K-tile whether 16 or 256 CUs run it, while the clock falls from 2.39 to 1.7-1.9 GHz; saving cycles (an L2 prefetch:
// -2 %) returns as a lower clock, reading a third fewer fragment bytes from LDS returns as a higher one (1.70 ->
// 1.92 GHz, -14 % time).  What a K-tile costs is therefore energy, and after the MFMAs the largest consumer a kernel
// controls is the LDS fragment traffic: a wave that owns 128 x 64 outputs reads (128 + 64) x 64 x 2 B = 24 KiB per
// K-tile, eight of them 192 KiB; a wave that owns 128 x 128 reads 32 KiB, four of them 128 KiB - a third less for the
// same MFMAs.  The price is the register file: 128 x 128 fp32 accumulators are 256 registers per lane, so a wave needs
// the whole 512-register budget of its SIMD (accumulators in AGPRs), there is one wave per SIMD and no partner wave to
// hide its waits behind - but idle cycles are cheap on a power-limited chip.
//
// K loop, per K-tile t (ring of two 64-KiB buffers, four 16-KiB half tiles A0 A1 B0 B1 each; wave (wr, wc) reads A[wr]
// and B[wc] only; F0 / F1 = the fragment registers of the tile's k-steps 0-31 / 32-63, 64 registers each):
//   top:  F1 <- LDS (tile t), issued between the first 16 MFMAs of k-step 0 (F0 was read during the previous K-tile)
//         wait for those reads; barrier "everyone is done with buffer t & 1"
//         the 16 LDS-DMA pieces of tile t+2 into that buffer, one per three MFMAs of k-step 0
//   mid:  wait for this wave's pieces of tile t+1 (counted: the 16 just issued stay in flight); barrier "tile t+1 landed"
//         F0 <- LDS (tile t+1), issued between the first 16 MFMAs of k-step 1; remaining MFMAs of k-step 1
// so every LDS read has half a K-tile (~1000 cycles) to return, up to two K-tiles (128 KiB) of LDS-DMA are in flight per
// CU, and the only instructions not paired with an MFMA are the two waits + barriers.  A piece is three instructions
// (m0, a hazard nop, buffer_load ... lds): its per-lane offset never changes, the K-tile's position travels in the
// scalar offset (K is a multiple of 64 here, so no lane needs a per-tile bounds decision).  Tiles past the end of K (the
// last two pieces rounds of a workgroup without a next tile) are issued with an out-of-range lane offset instead - the
// buffer unit writes zeros - so the wait counts are the same in every iteration.  A persistent workgroup stages the first
// two K-tiles of its next output tile under the last two K-tiles of the current one, and reads that tile's first
// fragments before its epilogue.
#pragma once
#include "../../csrc/kernels/gemm.hpp"

#ifndef PDT_GROUP_M4
#define PDT_GROUP_M4 8  // tile rows per rasterisation group (see tile_coords)
#endif
#ifndef PDT_ABLATE4
#define PDT_ABLATE4 0  // (bench/micro only) 4: no epilogue; 5: + no LDS-DMA in the K loop; 6: + no fragment reads in the K loop; 7: + neither
#endif
#ifndef PDT_STAMP
#define PDT_STAMP(id) do { } while (0)
#endif
// MFMAs of a K-tile issued before the wave joins the barrier that frees the buffer (its fragment reads must have
// returned by then: the barrier should find them done rather than wait for them)
#ifndef PDT_G4_EARLY
#define PDT_G4_EARLY 16
#endif
// placement of the 16 LDS-DMA pieces of a K-tile among its MFMA slots 16 .. 127 (see the K loop)
#ifndef PDT_G4_FIRST
#define PDT_G4_FIRST 16
#endif
#ifndef PDT_G4_GAP
#define PDT_G4_GAP 7
#endif
// cycles by which consecutive waves leave a barrier apart (0: together).  The four waves run the same instruction stream
// on four SIMDs but share one LDS and one address unit: requests issued in the same cycle queue behind each other, and
// with one wave per SIMD a queued request holds up the MFMAs behind it.
#ifndef PDT_G4_SKEW
#define PDT_G4_SKEW 8
#endif
// MFMA slots per fragment read in the two read sections (1: the 16 reads of a section go with its first 16 MFMAs; R: with
// its first 16 R, one read every R slots).  The four waves issue their reads in the same window and share one LDS port
// (128 B/clk): at one 1-KiB read per wave and 16-cycle slot they ask for twice what it delivers, the reads queue, and
// with one wave per SIMD a queued read holds up the MFMAs behind it.  The barrier that frees the buffer follows the last
// read, so PDT_G4_FIRST must be at least 16 R.
#ifndef PDT_G4_RSPREAD
#define PDT_G4_RSPREAD 1
#endif
// MFMA slot in front of which a wave waits for its fragment reads of the top section and joins the barrier that frees
// the buffer (default: right behind the last read, which then costs a full LDS latency; later slots give the reads the
// MFMAs in between to return)
#ifndef PDT_G4_BARRIER
#define PDT_G4_BARRIER (16 * PDT_G4_RSPREAD)
#endif
static_assert(PDT_G4_RSPREAD >= 1 && PDT_G4_RSPREAD <= 4 && PDT_G4_BARRIER >= 16 * PDT_G4_RSPREAD && PDT_G4_BARRIER <= 64 &&
              PDT_G4_FIRST >= PDT_G4_BARRIER, "read spread / barrier slot / first DMA slot");
static_assert(PDT_G4_FIRST + 15 * PDT_G4_GAP < 128, "the 16 DMA pieces of a K-tile must fit into its MFMA slots");
// pieces issued before the mid-tile wait (slots < 64): ceil((64 - FIRST) / GAP), at most 16
#define PDT_G4_PIECES_BEFORE_MID ((64 - PDT_G4_FIRST + PDT_G4_GAP - 1) / PDT_G4_GAP > 16 ? 16 : (64 - PDT_G4_FIRST + PDT_G4_GAP - 1) / PDT_G4_GAP)

namespace pdt {

template <typename T, bool AKS, bool BKS>
struct Gemm4 {
  using Base = GemmKernel<T, 256, 256, 2, 2, AKS, BKS>;
  static constexpr int NT = 256, BK = 64;
  static constexpr int HALF_BYTES = 128 * 128;
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;
  static constexpr int LDS_RING_BYTES = 2 * BUF_BYTES;
  static constexpr int STAGE_BYTES = 8 * 1024;                       // per wave, behind the ring
  static constexpr int LDS_BYTES = LDS_RING_BYTES + 4 * STAGE_BYTES;  // 160 KiB
  static constexpr int PIECES = 16;                                   // LDS-DMA instructions per wave and K-tile
  static constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;

  // Per-lane source offsets of one operand's 128-wide half-tile window (as GemmKernel::Loader with 256 threads and four
  // passes, without the per-tile K check: K % 64 == 0).  Piece p of a K-tile lands at image + p * 4096 + wave * 1024.
  template <bool KS>
  struct Loader4 {
    uint32_t off[4];
    int4v rsrc;
    uint32_t kstep;
    __device__ __forceinline__ void init(const T* X, int ld, int xdim, int x0, int kb, int klen, int tid) {
      const T* base;
      uint32_t records;
      const int xavail = min(128, xdim - x0);
      if (KS) {
        base = X + (size_t)kb * ld + x0;
        records = (uint32_t)(((size_t)(klen - 1) * ld + xavail) * 2);
        kstep = (uint32_t)ld * 2u * BK;
      } else {
        base = X + (size_t)x0 * ld + kb;
        records = (uint32_t)(((size_t)(xavail - 1) * ld + klen) * 2);
        kstep = BK * 2;
      }
      if (xavail <= 0) { base = X; records = 0; }
      rsrc = pdt_make_rsrc(base, records);
      offsets(ld, xavail, tid);
    }
    __device__ __forceinline__ void offsets(int ld, int xavail, int tid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int pos = p * NT * 16 + tid * 16;
        if (KS) {
          const int k = pos >> 8, w = pos & 255;
          const int chunk = (w >> 5) ^ Base::ks_key(k);
          const int col = chunk * 16 + ((w >> 4) & 1) * 8;
          off[p] = (col + 8 <= xavail) ? (uint32_t)k * ld * 2u + col * 2u : 0x80000000u;
        } else {
          const int row = pos >> 7, cpos = (pos >> 4) & 7;
          const int c = cpos ^ (row & 7);
          off[p] = row < xavail ? (uint32_t)row * ld * 2u + c * 16u : 0x80000000u;
        }
      }
    }
    __device__ __forceinline__ void retarget(const T* X, int ld, int x0, int kb, int klen) {
      const T* base = KS ? X + (size_t)kb * ld + x0 : X + (size_t)x0 * ld + kb;
      const uint32_t records = KS ? (uint32_t)(((size_t)(klen - 1) * ld + 128) * 2) : (uint32_t)(((size_t)(128 - 1) * ld + klen) * 2);
      rsrc = pdt_make_rsrc(base, records);
    }
    // piece p of a K-tile: `r` = this loader's descriptor and `soff` the K-tile's byte position (kt * kstep, kt inside the
    // window), or - for a piece of zeros - an empty descriptor (every lane out of range) and 0
    __device__ __forceinline__ void piece(char* image, int p, int4v r, uint32_t soff, int wave) {
      pdt_lds_dma16_s(r, image + p * 4096 + wave * 1024, off[p], soff);
    }
    __device__ __forceinline__ int4v descriptor(bool real) const {
      return int4v{rsrc[0], rsrc[1], real ? rsrc[2] : 0, rsrc[3]};
    }
  };
  using LoaderA = Loader4<AKS>;
  using LoaderB = Loader4<BKS>;

  static __device__ __forceinline__ short8 fragA(const char* img, int x0, int ks, int lane) {
    return AKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }
  static __device__ __forceinline__ short8 fragB(const char* img, int x0, int ks, int lane) {
    return BKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }

  static __device__ __forceinline__ void skew(int wave) {
    if (PDT_G4_SKEW > 0) {
      if (wave & 1) asm volatile("s_nop %0" ::"n"(PDT_G4_SKEW - 1));
      if (wave & 2) asm volatile("s_nop %0\n\ts_nop %0" ::"n"(PDT_G4_SKEW - 1));
    }
  }
  // (PDT_ABLATE4 == 8, bench/micro only: the other MFMA shape on the same operand registers - wrong results)
  typedef float float16v __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ void mfma32_acc(const short8& a, const short8& b, float16v& c) {
    asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void mfma_acc(const short8& a, const short8& b, float4v& c) {
    if (Half16<T>::code == DT_F16) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }

  static __device__ __forceinline__ void run(const PdtGemm& g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // tile walk: as in Gemm8::run (one tile per workgroup, or a persistent launch with per-XCD ticket counters)
    const int ntiles = g.tiles_m * g.tiles_n;
    const int total = ntiles * g.splits;
    const bool persistent = (int)gridDim.x < total;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int q_ = total >> 3, r_ = total & 7;
    const int xcd_count = q_ + (xcd < r_ ? 1 : 0);
    const int xcd_start = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
    int cur = xcd_remap(blockIdx.x, total);
    int nxt = (persistent && (int)(blockIdx.x >> 3) + per_xcd < xcd_count) ? cur + per_xcd : -1;
    auto locate = [&](int id, int& tm_, int& tn_, int& split_) {
      split_ = id / ntiles;
      tile_coords<PDT_GROUP_M4>(id - split_ * ntiles, g.tiles_m, g.tiles_n, tm_, tn_);
    };
    LoaderA la0, la1;
    LoaderB lb0, lb1;
    int klen;
    {
      int tm, tn, split;
      locate(cur, tm, tn, split);
      const int kb = split * g.k_per_split;
      klen = min(g.K - kb, g.k_per_split);
      la0.init((const T*)g.A, g.lda, g.M, tm * 256, kb, klen, tid);
      la1.init((const T*)g.A, g.lda, g.M, tm * 256 + 128, kb, klen, tid);
      lb0.init((const T*)g.B, g.ldb, g.N, tn * 256, kb, klen, tid);
      lb1.init((const T*)g.B, g.ldb, g.N, tn * 256 + 128, kb, klen, tid);
    }
    const int nkt = klen / BK;  // (host-checked: every K slice is a multiple of 64)

    float4v acc[8][8];
    float16v acc32[16];
    (void)acc32;
    short8 fa[2][8], fb[2][8];

    // piece Q (0..15: half tile Q >> 2 = A0 A1 B0 B1, pass Q & 3) of K-tile TILE into ring buffer TILE & 1; tiles at or
    // beyond nkt belong to the next output tile (the loaders have been re-pointed by then) or, without one, are zeros
    // (PDT_PIECES4_BEGIN sets up the four descriptors and the two scalar offsets of the round: no per-piece decisions)
#define PDT_PIECES4_BEGIN(TILE)                                                              \
    char* pb_ = smem + ((TILE)&1) * BUF_BYTES;                                               \
    const bool real_ = (TILE) < nkt || has_next;                                             \
    const uint32_t kt_ = real_ ? (uint32_t)((TILE) >= nkt ? (TILE) - nkt : (TILE)) : 0u;     \
    const int4v ra0_ = la0.descriptor(real_), ra1_ = la1.descriptor(real_);                  \
    const int4v rb0_ = lb0.descriptor(real_), rb1_ = lb1.descriptor(real_);                  \
    const uint32_t sa_ = kt_ * la0.kstep, sb_ = kt_ * lb0.kstep
#define PDT_PIECE4(Q)                                                                        \
  do {                                                                                       \
    if ((PDT_ABLATE4 == 5 || PDT_ABLATE4 == 7) && in_loop_) break;                           \
    if (((Q) >> 2) == 0) la0.piece(pb_ + OFF_A0, (Q)&3, ra0_, sa_, wave);                    \
    else if (((Q) >> 2) == 1) la1.piece(pb_ + OFF_A1, (Q)&3, ra1_, sa_, wave);               \
    else if (((Q) >> 2) == 2) lb0.piece(pb_ + OFF_B0, (Q)&3, rb0_, sb_, wave);               \
    else lb1.piece(pb_ + OFF_B1, (Q)&3, rb1_, sb_, wave);                                    \
  } while (0)
    // fragment read C (0..7: B columns C*16.., 8..15: A rows (C-8)*16..) of k-step KS from the wave's half tiles of BUF
#define PDT_FRAG4(C, KS, BUF)                                                                \
  do {                                                                                       \
    if ((PDT_ABLATE4 == 6 || PDT_ABLATE4 == 7) && in_loop_) break;                           \
    if ((C) < 8) fb[KS][(C)] = fragB((BUF) + (wc ? OFF_B1 : OFF_B0), (C) * 16, KS, lane);    \
    else fa[KS][(C)-8] = fragA((BUF) + (wr ? OFF_A1 : OFF_A0), ((C)-8) * 16, KS, lane);      \
  } while (0)
    // MFMA "A" operand = B-matrix rows (n), "B" operand = A-matrix rows (m): D[n][m], a lane owns 4 consecutive n
    // (issued through inline asm with the accumulator constrained to an AGPR tuple, updated in place: with the builtin
    // the register allocator moved accumulators through a[0:3] around every MFMA and spilled fragments to scratch)
#define PDT_MFMA4(C, KS) mfma_acc(fb[KS][(C)&7], fa[KS][(C) >> 3], acc[(C) >> 3][(C)&7])

    bool in_loop_ = false;
    {
      const bool has_next = false;
      {
        PDT_PIECES4_BEGIN(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) PDT_PIECE4(q);
      }
      {
        PDT_PIECES4_BEGIN(1);
#pragma unroll
        for (int q = 0; q < 16; ++q) PDT_PIECE4(q);
      }
    }
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int c = 0; c < 16; ++c) PDT_FRAG4(c, 0, smem);

    int stamp_seq = 0;
    (void)stamp_seq;
    for (;;) {
      const bool has_next = nxt >= 0;
      PDT_STAMP(0);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
      if (PDT_ABLATE4 == 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
      }

      in_loop_ = true;
      for (int t = 0; t < nkt; ++t) {
        const char* buf = smem + (t & 1) * BUF_BYTES;
        const char* nbuf = smem + ((t + 1) & 1) * BUF_BYTES;
        // ---- top: F1 <- tile t, paired with the first MFMAs of k-step 0
#pragma unroll
        for (int c = 0; c < PDT_G4_BARRIER; ++c) {
          if (c < 16 * PDT_G4_RSPREAD && c % PDT_G4_RSPREAD == 0) PDT_FRAG4(c / PDT_G4_RSPREAD, 1, buf);
          if (PDT_ABLATE4 == 8) { if ((c & 1) == 0) mfma32_acc(fb[0][(c >> 1) & 7], fa[0][0], acc32[(c >> 1) & 15]); }
          else PDT_MFMA4(c, 0);
          if ((PDT_G4_RSPREAD > 1 || PDT_G4_BARRIER > 16) && c % PDT_G4_RSPREAD == PDT_G4_RSPREAD - 1) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has read all it needs from buffer t & 1
        skew(wave);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next && t == nkt - 2) {
          int ntm_, ntn_, ns_;
          locate(nxt, ntm_, ntn_, ns_);
          la0.retarget((const T*)g.A, g.lda, ntm_ * 256, ns_ * g.k_per_split, klen);
          la1.retarget((const T*)g.A, g.lda, ntm_ * 256 + 128, ns_ * g.k_per_split, klen);
          lb0.retarget((const T*)g.B, g.ldb, ntn_ * 256, ns_ * g.k_per_split, klen);
          lb1.retarget((const T*)g.B, g.ldb, ntn_ * 256 + 128, ns_ * g.k_per_split, klen);
        }
        PDT_PIECES4_BEGIN(t + 2);
        // MFMA slots 16 .. 127 of the K-tile (k-step 0: 16 .. 63, k-step 1: 64 .. 127); piece q goes in front of slot
        // PDT_G4_FIRST + PDT_G4_GAP * q (+ PDT_G4_SKEW * wave: the four waves share one address unit, pieces issued in the
        // same cycle queue behind each other)
#define PDT_SLOT4(S)                                                                         \
  do {                                                                                       \
    if ((S) >= PDT_G4_FIRST && ((S) - PDT_G4_FIRST) % PDT_G4_GAP == 0 && ((S) - PDT_G4_FIRST) / PDT_G4_GAP < 16) \
      PDT_PIECE4(((S) - PDT_G4_FIRST) / PDT_G4_GAP);                                         \
    if (PDT_ABLATE4 == 8) {                                                                  \
      if (((S)&1) == 0) mfma32_acc(fb[(S) >> 6][((S) >> 1) & 7], fa[(S) >> 6][((S) >> 4) & 7], acc32[((S) >> 1) & 15]); \
    } else if ((S) < 64) PDT_MFMA4((S), 0); else PDT_MFMA4((S) - 64, 1);                     \
  } while (0)
#pragma unroll
        for (int sl = PDT_G4_BARRIER; sl < 64; ++sl) {
          PDT_SLOT4(sl);
          if ((sl & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- mid: tile t+1 has landed for this wave (the pieces of tile t+2 issued so far may still be in flight) ...
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PDT_G4_PIECES_BEFORE_MID) : "memory");
        __builtin_amdgcn_s_barrier();  // ... and for everybody else
        skew(wave);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 16 * PDT_G4_RSPREAD; ++c) {
          if (c % PDT_G4_RSPREAD == 0) PDT_FRAG4(c / PDT_G4_RSPREAD, 0, nbuf);
          PDT_SLOT4(64 + c);
          if (PDT_G4_RSPREAD > 1 && c % PDT_G4_RSPREAD == PDT_G4_RSPREAD - 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sl = 64 + 16 * PDT_G4_RSPREAD; sl < 128; ++sl) {
          PDT_SLOT4(sl);
          if ((sl & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#undef PDT_SLOT4
      }
      PDT_STAMP(1);
      // (the MFMAs are opaque to the compiler's hazard recognizer: keep their results away from the first read)
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      PDT_STAMP(2);

      // ------------------------------------------------------------------ epilogue
      auto kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(kargs));
      const PdtGemm& ge = *(const PdtGemm*)kargs;
      int tm, tn, split;
      locate(cur, tm, tn, split);
      const int m0 = tm * 256, n0 = tn * 256;
      uint32_t ticket = 0;
      const bool draw = has_next && tid == 0;
      if (draw) ticket = __hip_atomic_fetch_add(ge.sync + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      if (PDT_ABLATE4 == 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" :: "a"(acc32[i]));
      } else if (PDT_ABLATE4 >= 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(acc[i][j]));
      } else {
        epilogue(ge, acc, smem + LDS_RING_BYTES + wave * STAGE_BYTES, m0 + wr * 128, n0 + wc * 128, split, lane_e);
      }
      PDT_STAMP(3);
      // the pieces staged for the next tile were issued before the epilogue's own memory operations
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PDT_STAMP(4);
      ++stamp_seq;
      if (!has_next) break;
      int* mailbox = reinterpret_cast<int*>(smem + LDS_RING_BYTES);
      if (draw) {
        *mailbox = (int)ticket < xcd_count ? xcd_start + (int)ticket : -1;
        if ((int)ticket == xcd_count + per_xcd - 1) ge.sync[xcd] = 2u * per_xcd;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cur = nxt;
      __builtin_amdgcn_s_barrier();
      nxt = __builtin_amdgcn_readfirstlane(*mailbox);
      {
        int tid2 = tid;
        asm volatile("" : "+v"(tid2));
        la0.offsets(g.lda, 128, tid2); la1.offsets(g.lda, 128, tid2);
        lb0.offsets(g.ldb, 128, tid2); lb1.offsets(g.ldb, 128, tid2);
      }
    }
#undef PDT_PIECE4
#undef PDT_PIECES4_BEGIN
#undef PDT_FRAG4
#undef PDT_MFMA4
  }

  // Bring-up epilogue: 16-bit C straight from the accumulator layout (8 B per lane), or the fp32 slab of a split.
  // m_w / n_w: first row / column of the wave's 128 x 128 outputs.
  static __device__ __forceinline__ void epilogue(const PdtGemm& g, float4v (&acc)[8][8], char* stg, int m_w, int n_w, int split, int lane) {
    (void)stg;
    const int lm = lane & 15, lg = lane >> 4;
    if (g.epi == EPI_SLAB_F32) {
      const int64_t rows = g.M;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (float*)g.C + (size_t)split * g.M * g.ldc, 0, (int)(uint32_t)(((rows - 1) * g.ldc + g.N) * 4), 0x00020000);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t off = ((uint32_t)(m_w + i * 16 + lm) * g.ldc + (uint32_t)(n_w + j * 16 + lg * 4)) * 4u;
          float4v v = acc[i][j];
          if (g.slab_accumulate) v += __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rs, (int)off, 0, 0);
        }
      return;
    }
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        g.C, 0, (int)(uint32_t)(((int64_t)(g.M - 1) * g.ldc + g.N) * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        short4v s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = Half16<T>::from_f(acc[i][j][r]);
        const uint32_t off = ((uint32_t)(m_w + i * 16 + lm) * g.ldc + (uint32_t)(n_w + j * 16 + lg * 4)) * 2u;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, s), rc, (int)off, 0, 0);
      }
  }
};

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(256) void gemm4_kernel(PdtGemm g) {
  Gemm4<T, AKS, BKS>::run(g);
}

}  // namespace pdt
