// 256x256x64 "ping-pong" MFMA GEMM for gfx950: the large-shape kernel of this library.
//
// 8 waves = 2 groups of 4 (one wave of each group per SIMD).  Each K-tile is cut into four 16-KiB half
// tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255); a wave owns output rows
// {mh*128 + wr*64 .. +63 | mh=0,1} x cols {nh*128 + wc*32 .. +31 | nh=0,1}, so output quadrant (mh,nh) needs
// exactly half tile A[mh] and B[nh].  A K-tile is two phases: X reads A0,B0,B1 and computes quadrants
// (0,0),(0,1); Y reads A1 and computes (1,1),(1,0).  Per phase a wave does: LDS fragment reads, TWO half-tile
// LDS-DMA prefetches (4 x buffer_load ... lds), lgkmcnt(0), barrier, 32 MFMAs, barrier.  The second wave
// group runs one barrier behind the first, so on every SIMD one wave is in its MFMA section while its
// partner is in its load section: the matrix pipe stays busy and the LDS/VMEM traffic hides under it.
// (A four-phase variant, one quadrant per phase, is kept behind PDT_PHASES=4; it measured ~3 % slower:
// twice as many barriers.)
//
// Prefetch distance is 6 half tiles (1.5 K-tiles) over a 2-deep LDS ring (128 KiB).  A wave retires its
// LDS reads before the barrier that closes its load section, so a slot may be re-staged in the very next
// phase: X(t) issues (t+1: B1, A1) into slots last read in X(t-1)/Y(t-1), Y(t) issues (t+2: A0, B0) into
// slots last read in X(t).  LDS-DMA completion is tracked with ONE counted wait per K-tile
// (s_waitcnt vmcnt(4) in Y: everything but the two newest half tiles has landed) placed before a barrier;
// the data is first read in the NEXT phase, i.e. after a barrier every issuing wave has passed behind its
// wait.  Out-of-range half tiles (beyond K or the matrix edge) are still issued with an out-of-bounds
// offset - the buffer unit writes zeros - so the wait count is the same in every iteration.
#pragma once
#include "gemm.hpp"

// PDT_ABLATE (bench/micro/gemm8_ablate.hip only): compile parts of a phase out to see what bounds the loop
#ifndef PDT_ABLATE
#define PDT_ABLATE 0
#endif
// PDT_PRIO (experiments): 0 = raise priority around the MFMA cluster, 1 = no priority changes,
// 2 = raise priority in the load section instead
#ifndef PDT_PRIO
#define PDT_PRIO 0
#endif
// PDT_PHASES: phases per K-tile.  4 = one output quadrant (16 MFMAs) per phase; 2 = two quadrants (32 MFMAs)
// per phase, half as many barriers.
#ifndef PDT_PHASES
#define PDT_PHASES 2
#endif

namespace pdt {

template <typename T, bool AKS, bool BKS>
struct Gemm8 {
  using Base = GemmKernel<T, 256, 256, 2, 4, AKS, BKS>;
  static constexpr int NT = 512, BK = 64;
  static constexpr int HALF_BYTES = 128 * 128;         // one half tile
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;     // A0 | B0 | B1 | A1
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  using LoaderA = typename Base::template Loader<AKS, 128, 2>;
  using LoaderB = typename Base::template Loader<BKS, 128, 2>;

  // slot order inside a K-tile buffer = staging order = order of first use
  static constexpr int OFF_A0 = 0, OFF_B0 = HALF_BYTES, OFF_B1 = 2 * HALF_BYTES, OFF_A1 = 3 * HALF_BYTES;

  static __device__ __forceinline__ short8 fragA(const char* img, int x0, int ks, int lane) {
    if (PDT_ABLATE == 2) { short8 r; for (int i = 0; i < 8; ++i) r[i] = (short)(x0 + ks + lane); return r; }
    return AKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }
  static __device__ __forceinline__ short8 fragB(const char* img, int x0, int ks, int lane) {
    if (PDT_ABLATE == 2) { short8 r; for (int i = 0; i < 8; ++i) r[i] = (short)(x0 * 3 + ks + lane); return r; }
    return BKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }

  static __device__ __forceinline__ void run(const PdtGemm& g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int ntiles = g.tiles_m * g.tiles_n;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / ntiles, t_ = id % ntiles;
    int tm, tn;
    tile_coords(t_, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const int kb = split * g.k_per_split;
    const int klen = min(g.K - kb, g.k_per_split);
    const int nkt = (klen + BK - 1) / BK;

    LoaderA la0, la1;
    LoaderB lb0, lb1;
    la0.init((const T*)g.A, g.lda, g.M, m0, kb, klen, tid);
    la1.init((const T*)g.A, g.lda, g.M, m0 + 128, kb, klen, tid);
    lb0.init((const T*)g.B, g.ldb, g.N, n0, kb, klen, tid);
    lb1.init((const T*)g.B, g.ldb, g.N, n0 + 128, kb, klen, tid);

    float4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    // stage(i, tile): issue half tile i (0:A0 1:B0 2:B1 3:A1) of K-tile `tile` into ring buffer tile&1.
    // Tiles at or beyond nkt read out of bounds (kt*kstep runs past `records`, or the K check fails): zeros.
#define PDT_STAGE(I, TILE)                                                                   \
  do {                                                                                       \
    if (PDT_ABLATE == 1 && (TILE) > 1) break;                                                \
    char* b_ = smem + ((TILE)&1) * BUF_BYTES;                                                \
    if ((I) == 0) la0.issue(b_ + OFF_A0, (TILE), klen, wave);                                \
    else if ((I) == 1) lb0.issue(b_ + OFF_B0, (TILE), klen, wave);                           \
    else if ((I) == 2) lb1.issue(b_ + OFF_B1, (TILE), klen, wave);                           \
    else la1.issue(b_ + OFF_A1, (TILE), klen, wave);                                         \
  } while (0)

    // prologue: tile 0 completely + the first two half tiles of tile 1
    PDT_STAGE(0, 0); PDT_STAGE(1, 0); PDT_STAGE(2, 0); PDT_STAGE(3, 0);
    PDT_STAGE(0, 1); PDT_STAGE(1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the second wave group runs one barrier behind the first from here on
    if (wr == 1) __builtin_amdgcn_s_barrier();

    short8 fa[4][2], fb0[2][2], fb1[2][2];
    const int arow = wr * 64, bcol = wc * 32;

#define PDT_MFMA_QUADRANT(MH, FB, NH)                                                        \
  do {                                                                                       \
    if (PDT_PRIO == 0) __builtin_amdgcn_s_setprio(1);                                        \
    if (PDT_PRIO == 2) __builtin_amdgcn_s_setprio(0);                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                         \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                         \
      if (PDT_ABLATE == 3) { asm volatile("" :: "v"(FB[ni][ks]), "v"(fa[mi][ks])); }         \
      else acc[(MH)*4 + mi][(NH)*2 + ni] = Half16<T>::mfma(FB[ni][ks], fa[mi][ks], acc[(MH)*4 + mi][(NH)*2 + ni]); \
    if (PDT_PRIO == 0) __builtin_amdgcn_s_setprio(0);                                        \
    if (PDT_PRIO == 2) __builtin_amdgcn_s_setprio(1);                                        \
  } while (0)

#define PDT_SYNC_LOADS()                                                                     \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define PDT_END_PHASE()                                                                      \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)

#if PDT_PHASES == 4
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
      // ---- phase 0: quadrant (A0, B0) -----------------------------------------------------------
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb0[ni][ks] = fragB(buf + OFF_B0, bcol + ni * 16, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = fragA(buf + OFF_A0, arow + mi * 16, ks, lane);
      PDT_STAGE(2, t + 1);
      PDT_SYNC_LOADS();
      PDT_MFMA_QUADRANT(0, fb0, 0);
      PDT_END_PHASE();
      // ---- phase 1: quadrant (A0, B1) -----------------------------------------------------------
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb1[ni][ks] = fragB(buf + OFF_B1, bcol + ni * 16, ks, lane);
      PDT_STAGE(3, t + 1);
      PDT_SYNC_LOADS();
      PDT_MFMA_QUADRANT(0, fb1, 1);
      PDT_END_PHASE();
      // ---- phase 2: quadrant (A1, B1) -----------------------------------------------------------
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = fragA(buf + OFF_A1, arow + mi * 16, ks, lane);
      PDT_STAGE(0, t + 2);
      PDT_SYNC_LOADS();
      PDT_MFMA_QUADRANT(1, fb1, 1);
      PDT_END_PHASE();
      // ---- phase 3: quadrant (A1, B0); the K-tile's only DMA wait ---------------------------------
      PDT_STAGE(1, t + 2);
      // everything but the two newest half tiles (tile t+2: A0, B0) has landed => tile t+1 is complete
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PDT_SYNC_LOADS();
      PDT_MFMA_QUADRANT(1, fb0, 0);
      PDT_END_PHASE();
    }
#else
    // Two phases per K-tile.  The LDS reads of a phase are retired (lgkmcnt(0)) BEFORE the barrier that ends
    // its load section - the wave would wait at that barrier for its partner's MFMA section anyway - so a
    // slot may be re-staged one phase after its last read:
    //   X(t): read A0,B0,B1 of tile t ; issue (t+1: B1, A1) ; MFMA (A0,B0),(A0,B1)
    //   Y(t): read A1 of tile t       ; issue (t+2: A0, B0) ; wait vmcnt(4) => tile t+1 landed ; MFMA (A1,B1),(A1,B0)
#define PDT_SYNC_LOADS2()                                                                    \
  do {                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    __builtin_amdgcn_s_barrier();                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
    for (int t = 0; t < nkt; ++t) {
      const char* buf = smem + (t & 1) * BUF_BYTES;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb0[ni][ks] = fragB(buf + OFF_B0, bcol + ni * 16, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = fragA(buf + OFF_A0, arow + mi * 16, ks, lane);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb1[ni][ks] = fragB(buf + OFF_B1, bcol + ni * 16, ks, lane);
      PDT_STAGE(2, t + 1);
      PDT_STAGE(3, t + 1);
      PDT_SYNC_LOADS2();
      PDT_MFMA_QUADRANT(0, fb0, 0);
      PDT_MFMA_QUADRANT(0, fb1, 1);
      PDT_END_PHASE();
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = fragA(buf + OFF_A1, arow + mi * 16, ks, lane);
      PDT_STAGE(0, t + 2);
      PDT_STAGE(1, t + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PDT_SYNC_LOADS2();
      PDT_MFMA_QUADRANT(1, fb1, 1);
      PDT_MFMA_QUADRANT(1, fb0, 0);
      PDT_END_PHASE();
    }
#undef PDT_SYNC_LOADS2
#endif
    // re-align the two wave groups, drain the (zero-filling) DMA still in flight, then reuse LDS
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#undef PDT_STAGE
#undef PDT_MFMA_QUADRANT
#undef PDT_SYNC_LOADS
#undef PDT_END_PHASE

    // ------------------------------------------------------------------ epilogue
    const int lm = lane & 15, lg = lane >> 4;
    if (g.epi == EPI_SLAB_F32) {
      float* slab = (float*)g.C + (size_t)split * g.M * g.ldc;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + lm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + lg * 4;
          if (m < g.M && n < g.N) *reinterpret_cast<float4v*>(slab + (size_t)m * g.ldc + n) = acc[i][j];
        }
      }
      return;
    }
    // staged column block jc = lane&7: n-half jc>>2, 8 columns at (jc&3)*8; staged row block pair ip:
    // m-half ip>>1, rows (ip&1)*32.. of the wave's 64
    Base::template staged_epilogue<4>(g, acc, smem + wave * (32 * 256),
                                      n0 + ((lane & 7) >> 2) * 128 + wc * 32 + (lane & 3) * 8,
                                      [&](int ip) { return m0 + (ip >> 1) * 128 + wr * 64 + (ip & 1) * 32; }, tm * 2 + wr, lane);
  }
};

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(512) void gemm8_kernel(PdtGemm g) {
  Gemm8<T, AKS, BKS>::run(g);
}

}  // namespace pdt
