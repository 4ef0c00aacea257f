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
#include "gemm8_epilogue.hpp"

// PDT_ABLATE (bench/micro/gemm8_ablate.hip only): compile parts of a phase out to see what bounds the loop
// (5: the K loop on v_mfma_f32_32x32x16 fed with the same fragment registers, no epilogue - wrong results, the same
// operand traffic: what the other MFMA shape does to cycles and clock)
#ifndef PDT_GROUP_M8
#define PDT_GROUP_M8 8  // tile rows per rasterisation group of this kernel (see tile_coords)
#endif
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
// PDT_L2_TOUCH = D > 0: wave 0 of every workgroup pulls a share of the operand panels of K-tile t + 2 + D into the
// XCD's L2 while K-tile t is multiplied (see the K loop); 0 = off
#ifndef PDT_L2_TOUCH
#define PDT_L2_TOUCH 0
#endif
// PDT_STAMP(id) (bench/micro/gemm8_ablate.hip only): in-kernel time stamps at the tile / K loop / epilogue boundaries
#ifndef PDT_STAMP
#define PDT_STAMP(id) do { } while (0)
#endif

namespace pdt {

template <typename T, bool AKS, bool BKS>
struct Gemm8 {
  using Base = GemmKernel<T, 256, 256, 2, 4, AKS, BKS>;
  static constexpr int NT = 512, BK = 64;
  static constexpr int HALF_BYTES = 128 * 128;         // one half tile
  static constexpr int BUF_BYTES = 4 * HALF_BYTES;     // A0 | B0 | B1 | A1
  static constexpr int LDS_RING_BYTES = 2 * BUF_BYTES;
  static constexpr int LDS_BYTES = LDS_RING_BYTES + 4 * (32 * 256);  // ring + epilogue staging of waves 0-3 = 160 KiB
  using LoaderA = typename Base::template Loader<AKS, 128, 2>;
  using LoaderB = typename Base::template Loader<BKS, 128, 2>;

  // slot order inside a K-tile buffer = staging order = order of first use
  static constexpr int OFF_A0 = 0, OFF_B0 = HALF_BYTES, OFF_B1 = 2 * HALF_BYTES, OFF_A1 = 3 * HALF_BYTES;

  static __device__ __forceinline__ short8 fragA(const char* img, int x0, int ks, int lane) {
    if (PDT_ABLATE == 2) { short8 r; for (int i = 0; i < 8; ++i) r[i] = (short)(x0 + ks + lane); return r; }
    return AKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }
  static __device__ __forceinline__ short8 fragB(const char* img, int x0, int ks, int lane) {
    // (6: no B fragment reads - a third of the LDS read bytes, what a 128 x 128 wave tile would save)
    if (PDT_ABLATE == 2 || PDT_ABLATE == 6) { short8 r; for (int i = 0; i < 8; ++i) r[i] = (short)(x0 * 3 + ks + lane); return r; }
    return BKS ? Base::template frag_ks<128>(img, x0, ks, lane) : Base::frag_kc(img, x0, ks, lane);
  }

  static __device__ __forceinline__ void run(const PdtGemm& g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // One output tile per workgroup (gridDim.x == number of tiles), or a persistent launch: fewer workgroups than
    // tiles (host-checked: every tile full, an even number of K-tiles, no K split, >= 2 tiles per workgroup).  A
    // persistent workgroup keeps the LDS ring running across output tiles - the last phases of a tile stage the
    // first K-tiles of the next one, so its first loads fly under the epilogue instead of in front of an idle
    // matrix pipe.  Tiles are handed out per XCD (each XCD owns the contiguous run of tile ids xcd_remap gives
    // it): a workgroup's first two tiles are fixed, further ones are drawn from the XCD's ticket counter in
    // g.sync, one tile ahead (the draw is issued at the start of an epilogue and is back when it ends).  A
    // workgroup that is scheduled late (CUs busy with another stream's kernels) therefore costs at most its two
    // fixed tiles, not a fixed 1/256 of the problem.
    const int ntiles = g.tiles_m * g.tiles_n;
    const int total = ntiles * g.splits;
    const bool persistent = (int)gridDim.x < total;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int q_ = total >> 3, r_ = total & 7;
    const int xcd_count = q_ + (xcd < r_ ? 1 : 0);                                            // tiles of this XCD
    const int xcd_start = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;        // first tile id
    int cur = xcd_remap(blockIdx.x, total);  // == xcd_start + (blockIdx.x >> 3)
    int nxt = (persistent && (int)(blockIdx.x >> 3) + per_xcd < xcd_count) ? cur + per_xcd : -1;
    auto locate = [&](int id, int& tm_, int& tn_, int& split_) {
      split_ = id / ntiles;
      tile_coords<PDT_GROUP_M8>(id - split_ * ntiles, g.tiles_m, g.tiles_n, tm_, tn_);
    };
    LoaderA la0, la1;
    LoaderB lb0, lb1;
    int klen;
    int qa = 0, qb = 0;  // which quarter of its A / B panel this workgroup prefetches into L2 (PDT_L2_TOUCH)
    {
      int tm, tn, split;
      locate(cur, tm, tn, split);
      qa = tn & 3; qb = tm & 3;
      const int kb = split * g.k_per_split;
      klen = min(g.K - kb, g.k_per_split);  // (persistent launches: the same for every tile)
      la0.init((const T*)g.A, g.lda, g.M, tm * 256, kb, klen, tid);
      la1.init((const T*)g.A, g.lda, g.M, tm * 256 + 128, kb, klen, tid);
      lb0.init((const T*)g.B, g.ldb, g.N, tn * 256, kb, klen, tid);
      lb1.init((const T*)g.B, g.ldb, g.N, tn * 256 + 128, kb, klen, tid);
    }
    const int nkt = (klen + BK - 1) / BK;

    float4v acc[8][4];
    typedef float float16v __attribute__((ext_vector_type(16)));
    float16v acc32[8];
    (void)acc32;

    // stage(i, tile): issue half tile i (0:A0 1:B0 2:B1 3:A1) of K-tile `tile` into ring buffer tile&1.
    // Tiles at or beyond nkt belong to the next output tile of a persistent workgroup (the loaders have been
    // retargeted by then, kt_bias = nkt); without a next tile they read out of bounds (kt*kstep runs past
    // `records`, or the K check fails) and the buffer unit writes zeros.
#define PDT_STAGE(I, TILE)                                                                   \
  do {                                                                                       \
    if (PDT_ABLATE == 1 && (TILE) > 1) break;                                                \
    char* b_ = smem + ((TILE)&1) * BUF_BYTES;                                                \
    const int kt_ = ((TILE) >= nkt && has_next) ? (TILE) - nkt : (TILE);                     \
    if ((I) == 0) la0.issue(b_ + OFF_A0, kt_, klen, wave);                                   \
    else if ((I) == 1) lb0.issue(b_ + OFF_B0, kt_, klen, wave);                              \
    else if ((I) == 2) lb1.issue(b_ + OFF_B1, kt_, klen, wave);                              \
    else la1.issue(b_ + OFF_A1, kt_, klen, wave);                                            \
  } while (0)

    // prologue: tile 0 completely + the first two half tiles of tile 1
    {
      const bool has_next = false;
      PDT_STAGE(0, 0); PDT_STAGE(1, 0); PDT_STAGE(2, 0); PDT_STAGE(3, 0);
      PDT_STAGE(0, 1); PDT_STAGE(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");

    int stamp_seq = 0;
    (void)stamp_seq;
    for (;;) {
    const bool has_next = nxt >= 0;
    PDT_STAMP(0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    if (PDT_ABLATE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    }
    // K-tile 0 has landed (every wave waited for its own share before this barrier)
    __builtin_amdgcn_s_barrier();
    // the second wave group runs one barrier behind the first from here on
    if (wr == 1) __builtin_amdgcn_s_barrier();

    short8 fa[4][2], fb0[2][2], fb1[2][2];
    const int arow = wr * 64, bcol = wc * 32;

#define PDT_MFMA_QUADRANT(MH, FB, NH)                                                        \
  do {                                                                                       \
    if (PDT_PRIO == 0) __builtin_amdgcn_s_setprio(1);                                        \
    if (PDT_PRIO == 2) __builtin_amdgcn_s_setprio(0);                                        \
    if (PDT_ABLATE == 5) {                                                                   \
      _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                       \
      _Pragma("unroll") for (int b_ = 0; b_ < 2; ++b_)                                       \
        acc32[((MH)*2 + (NH)) * 2 + b_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(            \
            __builtin_bit_cast(half8, FB[b_][s_ & 1]), __builtin_bit_cast(half8, fa[b_ * 2 + (s_ >> 1)][s_ & 1]), \
            acc32[((MH)*2 + (NH)) * 2 + b_], 0, 0, 0);                                       \
    } else                                                                                   \
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

    // persistent launch: the last K-tiles of this output tile stage the first ones of the next
#define PDT_RETARGET_A0B0()                                                                  \
  do {                                                                                       \
    if (has_next && t == nkt - 2) {                                                          \
      int ntm_, ntn_, ns_;                                                                   \
      locate(nxt, ntm_, ntn_, ns_);                                                \
      la0.retarget((const T*)g.A, g.lda, ntm_ * 256, ns_ * g.k_per_split, klen);             \
      lb0.retarget((const T*)g.B, g.ldb, ntn_ * 256, ns_ * g.k_per_split, klen);             \
    }                                                                                        \
  } while (0)
#define PDT_RETARGET_B1A1()                                                                  \
  do {                                                                                       \
    if (has_next && t == nkt - 1) {                                                          \
      int ntm_, ntn_, ns_;                                                                   \
      locate(nxt, ntm_, ntn_, ns_);                                                \
      lb1.retarget((const T*)g.B, g.ldb, ntn_ * 256 + 128, ns_ * g.k_per_split, klen);       \
      la1.retarget((const T*)g.A, g.lda, ntm_ * 256 + 128, ns_ * g.k_per_split, klen);       \
    }                                                                                        \
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
      PDT_RETARGET_B1A1();
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
      PDT_RETARGET_A0B0();
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
      if (PDT_ABLATE == 7) {  // a third MORE LDS read bytes (8 extra fragment reads per K-tile, results discarded)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const short8 x0_ = fragB(buf + OFF_B0, ((bcol + 64) & 127) + ni * 16, ks, lane);
            const short8 x1_ = fragB(buf + OFF_B1, ((bcol + 64) & 127) + ni * 16, ks, lane);
            asm volatile("" :: "v"(x0_), "v"(x1_));
          }
      }
      PDT_RETARGET_B1A1();
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
      PDT_RETARGET_A0B0();
      PDT_STAGE(0, t + 2);
      PDT_STAGE(1, t + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      // L2 prefetch, issued behind the wait so that it has a whole K-tile to return before the next one: the four
      // workgroups of an XCD that share an A panel (eight: a B panel) ask for the same lines at the same time, so every
      // one of them sees the latency of the miss; a quarter of each panel touched D K-tiles early by each of them turns
      // the real transfers into L2 hits.  The junk dword per lane goes where this wave's next real transfer (B1 of
      // K-tile t+2, issued in the next phase, returned in order behind it) writes anyway.
      if (PDT_L2_TOUCH > 0 && wave == 0 && t + 2 + PDT_L2_TOUCH < nkt) {
        char* junk = smem + (t & 1) * BUF_BYTES + OFF_B1;
        if (qa & 2) la1.touch(junk, t + 2 + PDT_L2_TOUCH, qa & 1, g.lda, lane);
        else la0.touch(junk, t + 2 + PDT_L2_TOUCH, qa & 1, g.lda, lane);
        if (qb & 2) lb1.touch(junk, t + 2 + PDT_L2_TOUCH, qb & 1, g.ldb, lane);
        else lb0.touch(junk, t + 2 + PDT_L2_TOUCH, qb & 1, g.ldb, lane);
      }
      PDT_SYNC_LOADS2();
      PDT_MFMA_QUADRANT(1, fb1, 1);
      PDT_MFMA_QUADRANT(1, fb0, 0);
      PDT_END_PHASE();
    }
#undef PDT_SYNC_LOADS2
#endif
    // re-align the two wave groups: every wave has retired its last fragment reads.  The DMA still in flight
    // (the next tile's first K-tiles, or zero fill) targets buffer 0 and the A0/B0 slots of buffer 1; the epilogue
    // stages through the B1/A1 slots of buffer 1 and the 32 KiB behind the ring, so nothing has to drain here.
    PDT_STAMP(1);
    if (wr == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    PDT_STAMP(2);

    // ------------------------------------------------------------------ epilogue
    // The epilogue re-reads its (many) arguments from the kernel-argument segment for every tile, through a
    // pointer the optimiser cannot see through: hoisted out of the tile loop they would sit in ~40 SGPRs during
    // the main loop.  (PdtGemm is the kernel's only argument: offset 0 of the segment.)
    auto kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kargs));
    const PdtGemm& ge = *(const PdtGemm*)kargs;
    int tm, tn, split;
    locate(cur, tm, tn, split);
    const int m0 = tm * 256, n0 = tn * 256;
    // draw the ticket of the tile after next: the atomic's round trip hides under the epilogue
    uint32_t ticket = 0;
    const bool draw = has_next && tid == 0;
    if (draw) ticket = __hip_atomic_fetch_add(ge.sync + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (likewise the epilogue's per-lane address arithmetic: recomputed per tile from an opaque copy of the lane id
    // instead of living in registers through the main loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lm = lane_e & 15, lg = lane_e >> 4;
    if (PDT_ABLATE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(acc32[i]));
    } else if (PDT_ABLATE == 4) {  // no epilogue: keep the accumulators alive, store nothing
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else if (ge.epi == EPI_SLAB_F32) {  // (weight gradients: never persistent)
      float* slab = (float*)ge.C + (size_t)split * ge.M * ge.ldc;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + lm;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + lg * 4;
          if (m < ge.M && n < ge.N) {
            float4v* dst = reinterpret_cast<float4v*>(slab + (size_t)m * ge.ldc + n);
            *dst = ge.slab_accumulate ? *dst + acc[i][j] : acc[i][j];
          }
        }
      }
    } else {
      // staged column block jc = lane&7: n-half jc>>2, 8 columns at (jc&3)*8; staged row block pair ip:
      // m-half ip>>1, rows (ip&1)*32.. of the wave's 64
      char* stg = wave < 4 ? smem + LDS_RING_BYTES + wave * (32 * 256)
                           : smem + BUF_BYTES + OFF_B1 + (wave - 4) * (32 * 256);
      if (ge.fast_epi != 0 && m0 + 256 <= ge.M && n0 + 256 <= ge.N)
        FastEpilogue8<T, AKS, BKS>::run(ge, acc, stg, m0 + wr * 64, n0 + wc * 32, tm * 2 + wr, lane_e);
      else
        Base::template staged_epilogue<4>(ge, acc, stg, n0 + ((lane_e & 7) >> 2) * 128 + wc * 32 + (lane_e & 3) * 8,
                                          [&](int ip) { return m0 + (ip >> 1) * 128 + wr * 64 + (ip & 1) * 32; }, tm * 2 + wr, lane_e);
    }
    // the loads staged for the next tile (K-tile 0 and half of K-tile 1) were issued before the epilogue's own
    // memory operations: an empty counter means they have landed (and, at the end, that no zero fill is in flight)
    PDT_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PDT_STAMP(4);
    ++stamp_seq;
    if (!has_next) break;
    // hand the drawn tile id (or -1: the XCD's tiles are all taken) to the other waves through the first word of
    // wave 0's staging area; they read it behind the barrier at the top of the next tile
    int* mailbox = reinterpret_cast<int*>(smem + LDS_RING_BYTES);
    if (draw) {
      *mailbox = (int)ticket < xcd_count ? xcd_start + (int)ticket : -1;
      // every workgroup of the XCD draws exactly one ticket beyond the end: the last of them re-arms the counter
      // (the first 2 * per_xcd tickets are the fixed tiles) for the next launch on this stream
      if ((int)ticket == xcd_count + per_xcd - 1) ge.sync[xcd] = 2u * per_xcd;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur = nxt;
    if (PDT_L2_TOUCH > 0) {
      int ntm_, ntn_, ns_;
      locate(cur, ntm_, ntn_, ns_);
      qa = ntn_ & 3; qb = ntm_ & 3;
    }
    __builtin_amdgcn_s_barrier();
    nxt = __builtin_amdgcn_readfirstlane(*mailbox);
    {
      // the lane offsets are the same for every (full) tile: recomputed here from an opaque copy of the thread
      // id so that they do not occupy registers across the epilogue
      int tid2 = tid;
      asm volatile("" : "+v"(tid2));
      la0.offsets(g.lda, 128, tid2); la1.offsets(g.lda, 128, tid2);
      lb0.offsets(g.ldb, 128, tid2); lb1.offsets(g.ldb, 128, tid2);
    }
    }  // persistent tile loop
#undef PDT_STAGE
#undef PDT_MFMA_QUADRANT
#undef PDT_SYNC_LOADS
#undef PDT_END_PHASE
#undef PDT_RETARGET_A0B0
#undef PDT_RETARGET_B1A1
  }
};

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(512) void gemm8_kernel(PdtGemm g) {
  Gemm8<T, AKS, BKS>::run(g);
}

}  // namespace pdt
