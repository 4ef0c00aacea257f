// Plain-C launch interface of the gfx950 kernel library.  Included by the device translation units
// and by the host-only torch binding, so it contains no HIP device types.
// Every launcher enqueues on `stream` (a hipStream_t passed as void*), never synchronises and never
// allocates, so a caller may capture it into a hipGraph.  Return value: 0 on success, otherwise a
// negative argument-check code or a positive hipError_t.
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PDT_F16 = 0, PDT_BF16 = 1 };

typedef struct PdtRng {
  const uint64_t* state;  // device memory: state[0] = seed, state[1] = optimizer-step offset
  uint32_t micro;         // micro-batch counter (or, with sample ids, the optimizer-step counter)
  uint32_t stream;        // dropout site id
  // Optional global sample ids (device, int32 [B]).  When given, the dropout counters of row r of a [B*seq, H]
  // activation are derived from (sample_ids[r / seq], r % seq) instead of r, those of a [B, H] tensor from
  // sample_ids[b], and the attention masks from (sample_ids[b], head, query, key): a mask then belongs to the
  // SAMPLE, whichever rank, micro-batch or padded length the sample is processed with - data-parallel and
  // pipelined runs draw exactly the masks of a single-GPU run over the same global batch.
  const int32_t* sample_ids;
  uint32_t seq;           // tokens per sample of the activation the kernel works on (1 for [B, H] tensors)
  uint32_t seq_magic;     // ceil(2^32 / seq): row / seq == umulhi(row, seq_magic) for row * seq < 2^32
  // Unpadded (packed) activations [T, H], where a row's sample and position do not follow from its index: the
  // dropout row of every row, precomputed (sample id * 512 + position; device, uint32 [T]).  Takes precedence over
  // the (sample_ids, seq) rule for row-wise tensors; per-sample tensors and attention still use sample_ids.
  const uint32_t* row_table;
} PdtRng;

// ---- GEMM -------------------------------------------------------------------------------------
enum { PDT_EPI_PLAIN = 0, PDT_EPI_BIAS = 1, PDT_EPI_BIAS_GELU = 2, PDT_EPI_BIAS_DROP_RES = 3,
       PDT_EPI_MUL_DGELU = 4, PDT_EPI_ADD_RES = 5, PDT_EPI_SLAB_F32 = 6, PDT_EPI_BIAS_TANH = 7 };
// operand layouts: NT: A[M,K] B[N,K];  NN: A[M,K] B[K,N];  TN: A[K,M] B[K,N]
enum { PDT_GEMM_NT = 0, PDT_GEMM_NN = 1, PDT_GEMM_TN = 2 };

typedef struct PdtGemm {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int epi;
  const void* bias;
  const void* res;
  int ldr;
  void* aux;
  int ldaux;
  float drop_scale;
  uint32_t drop_thr;
  PdtRng rng;
  int k_per_split;
  int tiles_m, tiles_n, splits;
  // optional fp32 [2 * ceil(M / 128)][N] partial column sums of the stored C (the bias gradient of the layer
  // that produced C's input): every wave row of every tile row stores the sums of its rows into a row of its
  // own (plain stores, no atomics, deterministic); the launcher reports in colsum_rows how many rows were
  // written (2 per tile row), and pdt_colsum_finish() adds them up.
  float* colsum;
  int colsum_rows;
  // optional fp32 workspace [ws_splits][M][N] for split-K forward / dgrad GEMMs: when non-null (and the epilogue
  // has no column sums) the K range is cut into ws_splits slices whose fp32 partial tiles are summed by a
  // second kernel that also applies the epilogue.  pdt_gemm_splits() says how many slices a shape wants.
  float* ws;
  int ws_splits;
  // (set by the launcher) per-XCD ticket counters of a persistent launch of the 256x256 ping-pong kernel
  uint32_t* sync;
  // EPI_SLAB_F32 with ONE slab: add the tile to what C holds instead of overwriting it - a weight gradient whose K
  // is not split accumulates straight into the gradient arena, with no slab buffer and no reduction kernel
  int slab_accumulate;
  // (set by the launcher) the 256x256 ping-pong kernel may use its straight-line epilogue for tiles that lie completely
  // inside the output: the epilogue kind has one and every tensor it touches is addressable with 31-bit byte offsets
  int fast_epi;
} PdtGemm;

// tile: 0 = 128x128 (4 waves, double-buffered, 2 workgroups per CU), 1 = 256x256 (8 waves, one barrier per K-tile),
// 2 = 256x256 ping-pong (8 waves in two staggered groups, counted LDS-DMA waits), 3 = 128x128 with a 4-deep K-tile
// ring (1 workgroup per CU; for launches of about one workgroup per CU or fewer), 4 = 64x128 and 5 = 64x64 on the same
// ring (NT / NN resp. NT only: outputs with fewer 128x128 tiles than CUs), -1 = choose from the shape.
// For TN, C is the fp32 slab buffer [splits][M][ldc] and g->splits / k_per_split select the K split
// (0 = choose); the chosen values are written back into *g.
int pdt_gemm(int dtype, int layout, int tile, PdtGemm* g, void* stream);
// out[i] = (accumulate ? out[i] : 0) + sum_s slab[s][i]   (out fp32; n multiple of 4)
int pdt_splitk_reduce(const float* slab, int splits, int64_t n, float* out, int accumulate, void* stream);
int pdt_gemm_max_splits(void);
// Persistent launches of the 256x256 ping-pong kernel (256 workgroups walk the output tiles of a GEMM):
// 0 never, 1 from 4 tiles per workgroup (default), 2 from 2 tiles per workgroup.  The default comes from the
// PDT_GEMM_PERSIST environment variable; the setter overrides it (e.g. off while another stream's kernels -
// gradient all-reduces, pipeline send/recv - share the CUs: one workgroup per tile lets them in at tile
// granularity instead of at the end of a whole GEMM).
int pdt_gemm_persist_mode(void);
void pdt_gemm_set_persist_mode(int mode);
// Straight-line epilogue of the ping-pong kernel for full tiles (csrc/kernels/gemm8_epilogue.hpp): 1 on (default; the
// PDT_GEMM_FAST_EPI environment variable), 0 = the general epilogue everywhere (A/B switch)
int pdt_gemm_fast_epi_mode(void);
void pdt_gemm_set_fast_epi_mode(int on);
// K split of forward / dgrad GEMMs with few output tiles (see pdt_gemm_splits): 1 on (default), 0 off
void pdt_gemm_set_split_mode(int on);
int pdt_gemm_split_mode(void);
// slabs a TN (weight-gradient) GEMM of this shape writes (g->splits > 0: that many); size C accordingly
int pdt_gemm_tn_splits(int tile, const PdtGemm* g);
// Number of K slices the dispatcher would like for this NT / NN problem (1 = none): small outputs (few tiles for
// 256 CUs) with a long K.  The caller allocates ws = fp32 [splits][M][N] and sets g->ws / g->ws_splits.
int pdt_gemm_splits(int layout, int tile, const PdtGemm* g);

// ---- LayerNorm ----------------------------------------------------------------------------------
// y = LN(x) * gamma + beta ; stats[row] = {mean, rstd}.  H multiple of 8, H <= 8192.
int pdt_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* stats,
                      int rows, int H, float eps, void* stream);
// dx (16-bit) = LN backward wrt x;  ddrop (optional, 16-bit) = dx with the dropout mask of `rng` applied
// (gradient wrt the pre-dropout tensor);  partial[nblk][3][H] fp32 column partial sums of
// {dy*xhat, dy, ddrop-or-dx}.  Returns the number of partial blocks through *nblk.
int pdt_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const void* gamma,
                      void* dx, void* ddrop, float* partial, int* nblk, int rows, int H,
                      float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream);
int pdt_layernorm_bwd_max_blocks(void);
// partial is [nblk][nseg][H]; out_j[c] (+)= sum_b partial[b][j][c] for each non-null out_j (nseg <= 4)
int pdt_colsum_finish(const float* partial, int nblk, int H, int nseg, float* o0, float* o1, float* o2, float* o3,
                      int accumulate, void* stream);
// Several finishing jobs in ONE launch (a layer's backward produces four sets of partial sums; with short micro-batches
// the launches, not the sums, are what they cost).  Job j is pdt_colsum_finish(partial, nblk, H, nseg, out..., accumulate).
typedef struct PdtColsumJob {
  const float* partial;
  float* out[4];
  int nblk, H, nseg, accumulate;
} PdtColsumJob;
#define PDT_COLSUM_MAX_JOBS 6
int pdt_colsum_finish_multi(const PdtColsumJob* jobs, int njobs, void* stream);
// partial[nblk][ncols] column sums of a 16-bit [rows, ld] matrix (bias gradients)
int pdt_colsum_partial(int dtype, const void* x, int rows, int ncols, int ld, float* partial, int* nblk, void* stream);

// ---- Embeddings -----------------------------------------------------------------------------------
// y = dropout(LN(word[ids] + pos[pos_ids] + type[type_ids]));  pre = the LN input (saved for backward)
// (V, P, R = rows of the word / position / token-type tables: an index outside its table is clamped into it, so
// a corrupt or mismatched id tensor gives wrong embeddings, not a GPU memory fault)
int pdt_embed_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                  const void* inputs_embeds, const void* word, const void* pos, const void* type,
                  const void* gamma, const void* beta, void* pre, void* y, float* stats, int tokens, int H,
                  int V, int P, int R, float eps, float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream);
// table[idx[t]] += d[t] for the 16-bit rows d[0..n) (an embedding-table gradient), as a deterministic segmented sum:
// `sorted` = the indices in ascending order (stable sort), `perm[i]` = the token of sorted position i, `scratch` =
// fp32 [pdt_embed_seg_chunks(n)][2][H] work space.  Indices equal to skip_id or outside [0, rows) are ignored.
int pdt_embed_seg_chunks(int n);
int pdt_embed_scatter_sorted(int dtype, const void* d, const int64_t* sorted, const int64_t* perm, float* table,
                             float* scratch, int n, int H, int rows, int64_t skip_id, void* stream);
// token-type table gradient (R <= 4 rows) as deterministic masked column partial sums partial[nblk][R][H]
// (finish with pdt_colsum_finish, nseg = R)
int pdt_embed_type_partial(int dtype, const void* d, const int64_t* type_ids, float* partial, int* nblk, int tokens,
                           int H, int R, void* stream);
// apply the dropout mask of `rng` to a 16-bit [n / H, H] tensor: y = keep ? x*scale : 0 (the mask the embedding
// forward drew for the same rng)
int pdt_dropout_apply(int dtype, const void* x, void* y, int64_t n, int H, float drop_scale, uint32_t drop_thr,
                      PdtRng rng, void* stream);

// ---- Attention -------------------------------------------------------------------------------------
// qkv: 16-bit [B*S, 3H] (Q | K | V, head h at columns h*64 of each third); mask int32 [B,S];
// kv_len int32 [B]; ctx 16-bit [B*S, H]; lse fp32 [B, A, S].  head_dim == 64, S <= 512.
// cu (nullable): int32 [B+1] first rows of the samples in a PACKED qkv / ctx [T, ...] (unpadded batch); S is then the
// longest sequence, mask and kv_len are not read.
int pdt_attn_fwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, const int32_t* cu, void* ctx,
                 float* lse, int B, int S, int A, float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream);
// dqkv 16-bit [B*S, 3H];  delta fp32 [B,A,S] scratch.  dbias (nullable, fp32 [rows][3H] with rows =
// pdt_attn_bwd_dbias_rows(B, S), one per sample and 128-row block): partial column sums of dqkv (finish with
// pdt_colsum_finish(dbias, rows, 3H, 1, ...) to get the QKV bias gradient).
int pdt_attn_bwd_fuses_dbias(int S);
int pdt_attn_bwd_dbias_rows(int B, int S);
int pdt_attn_bwd(int dtype, const void* qkv, const int32_t* mask, const int32_t* kv_len, const int32_t* cu,
                 const void* ctx, const void* dctx, const float* lse, float* delta, void* dqkv, float* dbias, int B, int S,
                 int A, float drop_scale, uint32_t drop_thr, PdtRng rng, void* stream);

// ---- Classification head ------------------------------------------------------------------------------
// logits[b][c] = dot(dropout(pooled[b]), W[c]) + bias[c]; loss[0] = mean CE.  pooled 16-bit [B,H];
// W, bias, logits, loss fp32; C <= 8; labels may be null (no loss)
int pdt_head_fwd(int dtype, const void* pooled, const float* W, const float* bias, const int64_t* labels,
                 float* logits, float* loss, int B, int H, int C, float drop_scale, uint32_t drop_thr, PdtRng rng,
                 void* stream);
// gloss[0] = upstream gradient of the mean loss (null: no CE term); dlogits_in [B,C] = explicit upstream
// gradient of the logits (null: none).  dpooled 16-bit [B,H] (times 1-pooled^2 when tanh_bwd);
// dW [C,H], dbias [C] fp32 (+= when accumulate).  ws: fp32 workspace of pdt_head_bwd_ws_floats(B, H, C) elements (0 for
// small batches: then ws may be null) - larger batches are cut into slices whose shares a finishing launch adds up.
int64_t pdt_head_bwd_ws_floats(int B, int H, int C);
int pdt_head_bwd(int dtype, const void* pooled, const float* W, const float* logits, const int64_t* labels,
                 const float* gloss, const float* dlogits_in, void* dpooled, float* dW, float* dbias, int B, int H,
                 int C, float drop_scale,
                 uint32_t drop_thr, PdtRng rng, int accumulate, int tanh_bwd, float* ws, void* stream);

// ---- Optimizer ----------------------------------------------------------------------------------------
typedef struct PdtAdamW {
  float* p;          // fp32 master parameters
  float* m;
  float* v;
  const float* g;    // fp32 gradients (possibly loss-scaled)
  void* p16;         // 16-bit compute copy (may be null)
  int64_t n;
  float beta1, beta2, eps, weight_decay;
  int correct_bias;
  // device scalars: lr[0]; step[0] (int32, incremented by pdt_scaler_update); scale_state = {scale,
  // growth_tracker(as float), found_inf}
  const float* lr;
  const int32_t* step;
  float* scale_state;
} PdtAdamW;
// found_inf |= any(!isfinite(g))  (scale_state[2])
int pdt_grad_check(const float* g, int64_t n, float* scale_state, void* stream);
// p,m,v,p16 update with g/scale, skipped entirely when found_inf is set
int pdt_adamw_step(int dtype, const PdtAdamW* a, void* stream);
// dynamic loss-scale update + step counter: on overflow scale*=backoff, tracker=0, step unchanged;
// else tracker++ (scale*=growth every `interval` clean steps), step++.  Clears found_inf.
int pdt_scaler_update(float* scale_state, int32_t* step, float growth, float backoff, int interval,
                      int dynamic, void* stream);
// 16-bit copy of fp32 values (initial cast of the master arena)
int pdt_cast_f32_to_16(int dtype, const float* src, void* dst, int64_t n, void* stream);
int pdt_fill_f32(float* dst, float value, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
