// Internal launcher interface between the C-ABI layer (asr_api*.hip, asr_ctx.h) and the
// gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

namespace asr {

// Train-mode BatchNorm output of one raw element, y = (v - mu) * (gamma * inv_std) + beta, as ONE explicit sequence
// (subtract, fused multiply-add): the pooling forward (bn_apply_elu_pool_kernel), both passes of the backward
// (bn_bwd_*_kernel) and the debug export (pool_mask_kernel) decide "is this element a maximum of its window" by comparing
// these values for EQUALITY, so all of them must round the same way whatever the compiler would have contracted.
__device__ __forceinline__ float bn_affine(float v, float mu, float sc, float be) { return __fmaf_rn(v - mu, sc, be); }

// A data-gradient convolution can carry the REDUCE pass of the BatchNorm backward of the block whose output gradient it
// writes (round 5): per output element dA it reads that block's raw conv output z at the same index (pooled blocks: the
// selected raw value zsel, which has the pooled shape of dA) and accumulates sum dy and sum dy*xhat with dy = dA *
// ELU'(y) (* the tie multiplicity), y by bn_affine - into the same per-wave float64 partial table the forward kernels
// use for the statistics.  bn_bwd_reduce_kernel then does not run for that block: one read of dA and one launch less.
// Round 6: compiled OUT by default.  The form was built for the three Winograd families, is parity-green under both
// pooling rules and runs the batch-512 update 1.8 ms SLOWER (12.36 against 10.57 ms: the epilogue's loads are consumed at
// once by waves that have nothing else to issue) - and the four per-channel constants it keeps live across the M-tile loop
// pushed the RAW builds of conv3x3_winog / conv3x3_wino / conv3x3_wino4s over their register budget (68-136 bytes of
// scratch per lane, spill stores in every prologue).  -DASR_BNB_FUSE_BUILD=1 (tools/build_variant.sh) brings the path and
// its switch ASR_TRAIN_BNB_FUSE=1 back for experiments.
#ifndef ASR_BNB_FUSE_BUILD
#define ASR_BNB_FUSE_BUILD 0
#endif
struct BnBwdFuse {
    const float *z;          // (N,H,W,C) raw output, or (N,H/2,W/2,C) selected raw values of a pooled block
    const uint8_t *tie;      // pooled + "every tied element": (N,H/2,W/2,C/4) two bits per channel = ties - 1; else null
    const float *cst;        // the block's statistics buffer: [mu | inv_std | gamma*inv_std | beta] x C
};
__device__ __forceinline__ void bnb_acc(float dA, float zz, float mult, float mu, float sc, float be, float istd, float &t1,
                                        float &t2) {
    const float y = bn_affine(zz, mu, sc, be);
    const float dact = y <= 0.0f ? __expf(y) : 1.0f;           // ELU'(y) = exp(y) for y <= 0 (blocks 1..8 all carry ELU)
    const float dy = dA * dact * mult;
    t1 += dy;
    t2 = fmaf(dy, (zz - mu) * istd, t2);
}
__device__ __forceinline__ float bnb_mult(const uint8_t *tie, size_t e) {
    return tie ? (float)(((tie[e >> 2] >> (2 * (e & 3))) & 3u) + 1u) : 1.0f;
}

// ---- first block: conv3x3 (C_in = 1) + BN + ELU, prepare() folded in -------
// in_mode: ASR_IN_* of asr_hip.h.  Hraw/Wraw: raw sheet size; H/W: network
// resolution (= raw, or raw/2 when rsz).  w: [COUT][9] correlation-form taps,
// bnp: [3][COUTP] (mean, gamma*inv_std, beta).
hipError_t launch_conv1(hipStream_t s, const void *in, int in_mode, int rsz,
                        const float *w, const float *bnp, float *out,
                        int N, int Hraw, int Wraw, int H, int W, int cout);

const char *conv1_symbol(int cout, int in_mode, int rsz, int N, int H, int W, int Hraw, int Wraw);

// ---- blocks 2..8: conv3x3 (C_in >= 12) as implicit GEMM on fp32 MFMA -------
struct ConvPlan {           // chosen on the host per layer geometry
    int cin, cout, pool;
    int H, W, OH, OW;
    int TH, TW, NI;         // tile: NI images x TH x TW output pixels (pre-pool)
    int tiles_y, tiles_x;
    int threads, lds_bytes, blocks_per_cu;
    int tile_floats;        // v2: floats of one LDS tile buffer
    double cost;            // planner's model cost (relative)
    int fuse1;              // block 1 evaluated inside this (block 2) kernel
    int variant;            // index into the instantiation table
    const char *symbol;     // kernel symbol as rocprofv3 prints it
    int c4p, rp;            // Winograd schedule: LDS patch layout (chunks per pixel / per row)
};
// Returns false when no instantiation exists for (cin, cout, pool).
// raw = 1: plain convolution output (no BN/ELU/pool) - train-mode forward and data gradients
bool plan_conv(int cin, int cout, int pool, int H, int W, ConvPlan *plan, int raw = 0, int fuse1 = 0);
struct Fuse1Args {          // what a fuse1 plan needs instead of the block-1 activation
    const void *raw; const float *w1; const float *bn1; int in_mode, rsz, Hraw, Wraw;
};
void conv_candidates_v1(int cin, int cout, int pool, int H, int W, int raw, int max_count, std::vector<ConvPlan> *out,
                        int fuse1 = 0);
void conv_candidates_v2(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out);
// second-generation schedule (conv_v2_kernels.hip); plan.variant >= 1000 marks a v2 plan
bool plan_conv_v2(int cin, int cout, int pool, int H, int W, ConvPlan *plan);
hipError_t launch_conv_v2(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk,
                          const float *bnp, float *out, int N, int num_cus);
// third schedule for the small-K blocks (conv_v3_kernels.hip); plan.variant >= 2000 marks a v3 plan
void conv_candidates_v3(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out,
                        int fuse1 = 0);
bool plan_conv_v3_raw(int cin, int cout, int H, int W, ConvPlan *plan);
hipError_t launch_conv_v3(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk,
                          const float *bnp, float *out, int N, int num_cus, const Fuse1Args *f1 = nullptr);
// Winograd F(2x2,3x3) schedule (conv_wino_kernels.hip); plan.variant >= 3000 marks such a plan (plan.NI = MY).
// Its transformed weights live behind the direct-form fragments in the same buffer: wpk + conv_wpack_floats().
void conv_candidates_wino(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out);
// stats / stats_rows: RAW (training) plans only - the kernel also writes the partial table of its outputs'
// per-channel sums for the BatchNorm statistics ([rows][2][C_out] float64; launch_bn_stats_final reduces it)
// f1: plans with fuse1 set (block 1 evaluated inside by producer waves) read the raw input instead of `in`
hipError_t launch_conv_wino(hipStream_t s, const ConvPlan &p, const float *in, const float *wino_wpk,
                            const float *bnp, float *out, int N, int num_cus, double *stats = nullptr,
                            int *stats_rows = nullptr, const Fuse1Args *f1 = nullptr, bool stats_clean = false,
                            const BnBwdFuse *bf = nullptr);
void conv_candidates_wino_fused(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out);
const char *conv_wino_symbol(const ConvPlan &p, int in_mode);
int conv_wino_stats_rows_max(int num_cus);
size_t wino_wpack_floats(int cin, int cout);
// W: master weights [cout][cin][3][3] (Lasagne convolution form) on the device
// dgrad = 1: the data-gradient convolution's weights (contraction over cout, outputs cin)
hipError_t launch_wino_pack(hipStream_t s, const float *W, int cin, int cout, float *wino_wpk, int dgrad = 0);
bool plan_conv_wino_raw(int cin, int cout, int H, int W, ConvPlan *plan);
void conv_candidates_wino_raw(int cin, int cout, int H, int W, int max_count, std::vector<ConvPlan> *out);
// Winograd F(4x4,3x3), global-A form (conv_wino4_kernels.hip); plan.variant >= 4000.  Its weights follow the
// F(2x2,3x3) ones in the layer's weight buffer: wpk + conv_wpack_floats() + wino_wpack_floats()
void conv_candidates_wino4(int cin, int cout, int pool, int H, int W, std::vector<ConvPlan> *out);
hipError_t launch_conv_wino4(hipStream_t s, const ConvPlan &p, const float *in, const float *wino4_wpk,
                             const float *bnp, float *out, int N, int num_cus, double *stats = nullptr,
                             int *stats_rows = nullptr, bool stats_clean = false, const BnBwdFuse *bf = nullptr);
// RAW (training) builds of conv3x3_wino4s: candidates of the training step's tuner for a block's forward convolution
// (cin, cout; dgrad = 0: only with ASR_TRAIN_WINO4=1, see conv_wino4_kernels.hip) or data gradient (cout, cin; dgrad =
// 1); their statistics table has one row per workgroup
void conv_candidates_wino4_raw(int cin, int cout, int H, int W, std::vector<ConvPlan> *out, int dgrad);
bool conv_wino4_is_raw(const ConvPlan &p);
int conv_wino4_stats_rows_max(int num_cus);
size_t wino4_wpack_floats(int cin, int cout);
hipError_t launch_wino4_pack(hipStream_t s, const float *W, int cin, int cout, float *wino4_wpk, int dgrad = 0);
size_t conv_wpack_floats(int cin, int cout);
// Wcorr: [9][cin][cout] correlation-form taps -> MFMA fragment order.
void pack_conv_weights(const float *wcorr, int cin, int cout, float *wpk);
hipError_t launch_conv(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk,
                       const float *bnp, float *out, int N, int num_cus, const Fuse1Args *f1 = nullptr);

// ---- tail: 1x1 conv + BN + global mean (+ CCA projection + length norm) ----
// a8: [N, h, w, c8] NHWC; w9: [32][c8]; bnp9: [3][32]; cca_mean: [32];
// cca_proj: [32][32] (U or V, row-major, rows = input dim).
// features (N,32) and/or latent (N,32) may be null.
hipError_t launch_tail(hipStream_t s, const float *a8, int N, int h, int w, int c8,
                       const float *w9, const float *bnp9, const float *cca_mean,
                       const float *cca_proj, float *features, float *latent);

// ---- ranking ---------------------------------------------------------------
hipError_t launch_row_norms(hipStream_t s, const float *x, int64_t n, int64_t ld, int dim, double *norms);
// rn2_pre (may be null): (float)(1 / norm2[j]) of every candidate, padded to a multiple of 4 - a resident data base
// brings them along, otherwise they are derived into the workspace
hipError_t launch_rank(hipStream_t s, const float *lv1, const double *norm1, int64_t n1, int64_t ld1,
                       const float *lv2, const double *norm2, int64_t n2, int64_t ld2, int dim,
                       int64_t query_offset, int64_t k, int64_t h,
                       int32_t *ranks, double *dstar, int32_t *ties, void *workspace = nullptr,
                       const float *rn2_pre = nullptr);
// workspace of the MFMA counting path of launch_rank (candidate sets >= 2048)
size_t rank_workspace_bytes(int64_t n1, int64_t n2);

// resident code data base (32-d packed rows): float64 norms [n], fp32 reciprocal norms [n rounded up to 4, zero
// padded], unit-length fp32 copy of the rows [n][32] - one pass over the pool
hipError_t launch_db_prepare(hipStream_t s, const float *x, int64_t n, double *norms, float *rn, float *unit);

// top-k smallest cosine distances per query (exact float64, stable index order); k <= 128
// workspace (topk_workspace_bytes; may be null): enables the fp32-MFMA filter stage in front of the exact scan
// unit / rn_db_pre (may be null): the resident data base's unit-length rows / reciprocal norms
size_t topk_workspace_bytes(int64_t n_db, int64_t n_q, int k, bool unit, bool fuse_rank);
hipError_t launch_topk(hipStream_t s, const float *db, const double *norm_db, int64_t n_db, int64_t ld_db,
                       const float *q, const double *norm_q, int64_t n_q, int64_t ld_q, int dim, int k,
                       int64_t idx_offset, int32_t *idx_out, double *dist_out, void *workspace,
                       const float *unit = nullptr, const float *rn_db_pre = nullptr, double *norm_q_pending = nullptr,
                       unsigned *tickets = nullptr);
// tickets (may be null): 4096 zero-initialised ints that stay zero between calls (1024 tickets, then the sort-free refine's
// per-query counts and flags) - the small dependent launches of
// the few-queries shape (threshold select, merge of the partial lists) are then done by the last workgroup to arrive in
// the kernel in front of them
// (norm_q_pending == norm_q: the query norms are NOT computed yet - launch_topk does it, inside the seeding kernel where
// that path runs, by launch_row_norms otherwise)
// a shard of a larger pool (rows [item_offset, ...) of n2_global): d* / global j* of the queries whose correct candidates
// it holds; the fused pass with d* / j* given, leaving the rank COUNTERS [n_q][3] (summed over the shards by the caller);
// the merge of top-k lists gathered from the shards ([part][n_q_total][k]); counters -> ranks
hipError_t launch_rank_dstar(hipStream_t s, const float *q, const double *norm_q, int64_t n_q, const float *db,
                             const double *norm_db, int64_t n_db, int64_t item_offset, int64_t n2_global,
                             int64_t query_offset, int64_t kk, int64_t hh, double *dstar, int64_t *jstar);
hipError_t launch_topk_count_db(hipStream_t s, const float *db, const float *unit, const double *norm_db, int64_t n_db,
                                const float *q, const double *norm_q, int64_t n_q, int k, int64_t idx_offset,
                                int32_t *idx_out, double *dist_out, const double *dstar, const int64_t *jstar,
                                int32_t *counts, void *workspace);
hipError_t launch_topk_merge(hipStream_t s, const int32_t *part_idx, const double *part_dist, int n_parts, int64_t n_q_total,
                             int64_t q_lo, int64_t n_q, int k, int32_t *idx_out, double *dist_out);
hipError_t launch_rank_finish(hipStream_t s, const int32_t *counts, const double *dstar, int64_t n, int32_t *ranks,
                              double *dstar_out, int32_t *ties);
// top-k and eval_retrieval ranks from one walk over the pool (needs topk_rank_fusable; 32-d packed rows)
bool topk_rank_fusable(int64_t n_db, int64_t kk);
hipError_t launch_topk_rank_db(hipStream_t s, const float *db, const float *unit, const double *norm_db, int64_t n_db,
                               const float *q, const double *norm_q, int64_t n_q, int k, int64_t idx_offset,
                               int32_t *idx_out, double *dist_out, int64_t query_offset, int64_t kk, int64_t hh,
                               int32_t *ranks, double *dstar, int32_t *ties, void *workspace);

// ---- alignment: cosine distance matrix + DTW (utils/alignment.py, utils/dtw_by_dist.py) ----
// D: (R+1)*(C+1) doubles workspace; dist_out: R*C doubles or null; path_*: R+C entries, reversed order
hipError_t launch_dtw(hipStream_t s, const float *a, const double *na, int64_t R, int64_t lda, const float *b,
                      const double *nb, int64_t C, int64_t ldb, int dim, double *D, double *dist_out, int32_t *path_i,
                      int32_t *path_j, int32_t *path_len, double *min_dist);

// ---- piece-identification vote (audio_sheet_server.py:213-300) -----------------
hipError_t launch_slice_windows(hipStream_t s, const float *src, int64_t T, int r0, int win_h, int win_w,
                                const int32_t *starts_dev, int n, float *out);
// counts_ws: n_pieces int32 workspace; out_piece/out_count: top_k entries (piece -1 / count 0 when fewer voted)
hipError_t launch_piece_vote(hipStream_t s, const int32_t *idx, int64_t n_idx, const int32_t *ids, int64_t n_db,
                             int32_t n_pieces, int top_k, int32_t *counts_ws, int32_t *out_piece, int32_t *out_count);

// autotuner self-check (ASR_TUNE_VERIFY=1): deterministic input pattern, max |a - b| as float bits
hipError_t launch_fill_pattern(hipStream_t s, float *p, int64_t n);
hipError_t launch_max_abs_diff(hipStream_t s, const float *a, const float *b, int64_t n, uint32_t *out_bits);
// audio front-end: framed, windowed |DFT| -> filterbank -> log10(mul * x + add); all pointers on the device
hipError_t launch_spectrogram(hipStream_t s, const float *samples, int64_t n_samples, const float *window, int frame_size,
                              double hop, int max_bin, const int32_t *fb_start, const int32_t *fb_len,
                              const int32_t *fb_off, const float *fb_w, int nf, float mul, float add, float *out,
                              int64_t n_frames, int transposed);
// batch assembly of the training pool: desc_dev holds n x 9 doubles (see piece_vote_kernels.hip)
hipError_t launch_gather_windows(hipStream_t s, const float *src, const double *desc_dev, int n, int out_h, int out_w,
                                 float *out);

// ---- CCA re-estimation (refine_cca.py / utils/cca.py 'svd') ------------------
size_t cca_workspace_bytes(int64_t n);
// H1,H2: [n][32] fp32 device; outputs device: U,V [32][32] fp32, means [64] fp32
// (m1 | m2), coeffs [32] fp64.
hipError_t launch_cca_fit(hipStream_t s, const float *H1, const float *H2, int64_t n, float r1, float r2,
                          void *workspace, float *U, float *V, float *means, double *coeffs);

// ---- data-parallel exchange points (null: single GPU) ------------------------------
// Training shards the batch over `world` ranks (shards may differ by one row: n_local of n_global samples live here);
// the per-channel BatchNorm sums (forward and
// backward) are all-reduced in place so that every rank normalises with the statistics of the FULL batch
// (SURVEY.md 8e).  allreduce sums `count` doubles in place on stream s (RCCL enqueues; a host callback synchronises).
struct Exchange {
    int (*allreduce_f64)(void *self, hipStream_t s, double *buf, int64_t count);
    void *self;
    int world;
    int n_local, n_global;      // this step's shard / whole batch, set by the training step before its first launch
    // 0: a launcher runs its local part, the all-reduce and the rest in one go.  The data-parallel step pairs the two
    // towers' exchanges - one all-reduce per block instead of two: 1 = local part only (the sums are left in `sums`),
    // 2 = only what follows the all-reduce (the caller has summed `sums` over the ranks in between)
    int phase;
};

// ---- training: forward with batch statistics --------------------------------
// stats / stats_rows (may be null): partial table of the outputs' per-channel sums, [rows][2][cout] float64
// mode 0: z + statistics; 1: statistics only (z unused); 2: BatchNorm (bn = [mu | inv_std], gamma, beta) + ELU of the
// recomputed z written to `z` (block 1's output)
hipError_t launch_conv1_raw(hipStream_t s, const float *x, const float *w, float *z, int N, int H, int W, int cout,
                            double *stats, int *stats_rows, int mode = 0, const float *bn = nullptr,
                            const float *gamma = nullptr, const float *beta = nullptr);
// long float64 partial tables [nb][cols] are pre-summed 32 rows per workgroup into the space BEHIND the table (allocate
// colsum_stage_extra(doubles) more); returns the table the single-workgroup finish reads and updates *nb
size_t colsum_stage_extra(size_t partial_doubles);
double *colsum_stage(hipStream_t s, double *partial, int *nb, int cols);
int bn_stats_blocks(int64_t rows);
// z: rows x C; partial: bn_stats_blocks(rows)*2*C doubles; stats: [mu | inv_std]; run_*: EMA targets or null
// ex != null: `sums` (2*C doubles) carries the local column sums through the all-reduce; rows counts the local shard
hipError_t launch_bn_stats(hipStream_t s, const float *z, int64_t rows, int C, double *partial, float *stats,
                           float *run_mean, float *run_istd, float eps, float ema, const Exchange *ex = nullptr,
                           double *sums = nullptr, unsigned *ticket = nullptr);
// the same finish from a partial table the convolution wrote itself (nb rows of [2][C]); rows = count behind the sums
// ticket (single GPU: ex == null): one zero-initialised counter -> the reduction runs as ONE launch (colsum_final_kernel:
// 32-row groups, the last workgroup to arrive finishes); zero_rows: that launch also clears the rows it consumed (the
// convolutions' own statistics table stays all-zero between uses, so they need no memset before they write their rows;
// `staged` then has to lie OUTSIDE that table: ceil(nb / 32) rows of 2*C doubles; default: behind the nb rows)
hipError_t launch_bn_stats_final(hipStream_t s, double *partial, int nb, int64_t rows, int C, float *stats,
                                 float *run_mean, float *run_istd, float eps, float ema, const Exchange *ex = nullptr,
                                 double *sums = nullptr, unsigned *ticket = nullptr, bool zero_rows = false,
                                 double *staged = nullptr);
// zsel (pooled blocks, may be null): (N,OH,OW,C) raw value of each window's selected element, for launch_bn_bwd.
// ztie (with zsel, may be null): (N,OH,OW,C/4) bytes, two bits per channel = (number of window elements whose y equals the
// maximum) - 1: what the reduce pass of launch_bn_bwd multiplies by under the "every tied element" pooling gradient.
// snap (pooled blocks, may be null): 2*C floats that receive [gamma * inv_std | beta] as this pass used them.
hipError_t launch_bn_apply(hipStream_t s, const float *z, const float *stats, const float *gamma, const float *beta,
                           float *out, int N, int H, int W, int C, int pool, int elu, float *zsel = nullptr,
                           uint8_t *ztie = nullptr, float *snap = nullptr);
// debug export: (N,OH,OW,C) floats holding the 4-bit set {rr : y(window element rr) == max y} (bit rr = 2 dy + dx) - the
// elements a pooled block's backward pass feeds under ASR_POOL_TIES_ALL, bit for bit (same bn_affine on the same
// inputs: snap = what launch_bn_apply kept)
hipError_t launch_pool_mask(hipStream_t s, const float *z, const float *stats, const float *snap, float *mask, int N, int H,
                            int W, int C);
hipError_t launch_conv1x1_raw(hipStream_t s, const float *a8, const float *w9, float *z9, int64_t rows, int C8);
hipError_t launch_bn_gpool(hipStream_t s, const float *z9, const float *stats, const float *gamma, const float *beta,
                           float *Hout, int N, int npix);
// CCALayer train branch + length norm + ranking loss, forward and backward (single workgroup, float64)
size_t cca_train_ws_bytes(int B);
hipError_t launch_cca_train(hipStream_t s, const float *H1, const float *H2, int B, const float *cca_in,
                            float *cca_out, float r1, float r2, float rT, float alpha, float gamma, void *ws,
                            float *loss_out, float *lv1, float *lv2, float *dH1, float *dH2, float weight = 1.0f,
                            int symmetric = 0);

// get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69) of given embeddings, no gradients
hipError_t launch_rank_loss(hipStream_t s, const float *lv1, const float *lv2, int B, float gamma, float *loss_out,
                            float weight = 1.0f, int symmetric = 0);

// ---- training: backward + update ------------------------------------------------
int bn_bwd_blocks(int64_t opix);
// dz must not alias z for pooled blocks (dz = null: reduce pass and batch sums only).  partial: bn_bwd_blocks*2*C doubles;
// sums: 2*C doubles.  zsel (pooled blocks, may be null): what launch_bn_apply wrote - the reduce pass then reads it
// instead of the four window elements of z (the same values, the same sums).
// ties_first = 0 (ASR_POOL_TIES_ALL): every window element whose y equals the window maximum receives the pooled gradient
// (Theano's CPU MaxPoolGrad); with zsel the reduce pass needs ztie (launch_bn_apply) for the multiplicities.
// ties_first = 1: only the first such element in row-major order.
hipError_t launch_bn_bwd(hipStream_t s, const float *z, float *dz, const float *dout, const float *stats,
                         const float *gamma, const float *beta, double *partial, double *sums, float *dbeta,
                         float *dgamma, int N, int H, int W, int C, int pool, int elu, const Exchange *ex = nullptr,
                         const float *zsel = nullptr, const uint8_t *ztie = nullptr, int ties_first = 0,
                         unsigned *ticket = nullptr, double *pre_partial = nullptr, int pre_rows = 0,
                         double *pre_staged = nullptr);
// pre_partial / pre_rows (> 0) / pre_staged: the reduce pass already ran inside the data-gradient convolution that wrote
// `dout` (BnBwdFuse): its per-wave partial table (the zero-between-uses statistics table) is reduced, the reduce kernel
// is skipped.  Needs ticket.
// One launch for "column sums of a partial table [nb][cols] (float64) + what follows": workgroup b sums rows 32b .. 32b+31
// into a staged row behind the table, the LAST workgroup to arrive (atomic ticket, reset to 0 for the next use) sums the
// staged rows in fixed order - deterministic for a given nb - and finishes.  mode 0: BatchNorm statistics (mu, inv_std,
// EMA of both); mode 1: sums[cols] only; mode 2: sums[cols] + dbeta / dgamma (BatchNorm backward).  Replaces
// colsum_stage_kernel + bn_stats_final_kernel / bn_bwd_final_kernel (two launches, and a memset before the producer).
struct ColsumFinalArgs {
    double *partial; int nb, cols; double *staged; unsigned *ticket; int zero_rows, mode, C;
    double count; float eps, ema; float *stats, *run_mean, *run_istd; double *sums; float *dbeta, *dgamma;
};
hipError_t launch_colsum_final(hipStream_t s, const ColsumFinalArgs &a);
struct WgradPlan {
    int cin, cout, H, W, TH, TW, tiles_y, tiles_x, lds_bytes, variant, grid_cap;
};
bool plan_wgrad(int cin, int cout, int H, int W, int num_cus, WgradPlan *p);
void wgrad_candidates(int cin, int cout, int H, int W, int num_cus, int max_count, std::vector<WgradPlan> *out);
size_t wgrad_partial_floats(const WgradPlan &p);
hipError_t launch_wgrad(hipStream_t s, const WgradPlan &p, const float *x, const float *dz, int N, float *partial,
                        float *dW);
int conv1_wgrad_blocks();
// z != null: block 1's BatchNorm / ELU backward fused (launch_bn_bwd was called with dz = null: reduce pass only;
// `sums` are its batch sums); dz is then unused
hipError_t launch_conv1_wgrad(hipStream_t s, const float *x, const float *dz, int N, int H, int W, int cout,
                              double *partial, float *dW, const float *z = nullptr, const float *dout = nullptr,
                              const float *stats = nullptr, const float *gamma = nullptr, const float *beta = nullptr,
                              const double *sums = nullptr, int n_global = 0, const float *w1 = nullptr);
// block 1 without its raw tensor: reduce pass + batch sums of its BatchNorm backward with z recomputed from the image
// (w1: block 1's [C][9] taps); launch_conv1_wgrad(..., z = null, w1) applies dz the same way
hipError_t launch_bn_bwd_conv1(hipStream_t s, const float *x, const float *w1, const float *dout, const float *stats,
                               const float *gamma, const float *beta, double *partial, double *sums, float *dbeta,
                               float *dgamma, int N, int H, int W, int C, const Exchange *ex = nullptr);
int tail_dw_blocks(int64_t rows);
// partial: tail_dw_blocks(N * npix) * 32 * C8 + N * 64 doubles
hipError_t launch_tail_bwd(hipStream_t s, const float *dH, float *z9, const float *a8, const float *w9,
                           const float *stats, const float *gamma, int N, int npix, int C8, double *sums,
                           double *partial, float *dbeta, float *dgamma, float *dW9, float *da8,
                           const Exchange *ex = nullptr);
hipError_t launch_adam(hipStream_t s, float *p, const float *g, float *m, float *v, const unsigned char *mask,
                       int64_t n, float a_t, double beta1, double beta2, float eps, float l2);
hipError_t launch_l2_penalty(hipStream_t s, const float *p, const unsigned char *mask, int64_t n, double *out);
// one conv block's re-layout work for repack_all_kernel (train_bwd_kernels.hip); null pointers = not wanted
struct RepackDesc {
    const float *W, *beta, *gamma, *mean, *istd;      // master parameters of the block
    float *wfwd, *wdgrad;                             // direct-form fragments (kind 0: the [co][9] taps; kind 2: copy target)
    float *wino_fwd, *wino_dgrad;                     // Winograd-domain copies
    float *wino4_fwd, *wino4_dgrad;                   // F(4x4) Winograd-domain copies (where the step's plans use them)
    float *bnp;                                       // deterministic-path BN fold
    int cin, cout;                                    // kind 2: cin * cout = elements to copy
    int kind;                                         // 0: block 1 (C_in = 1), 1: 3x3 block, 2: plain copy
};
hipError_t launch_repack_all(hipStream_t s, const RepackDesc *descs_dev, int n_descs);
hipError_t launch_repack_conv(hipStream_t s, const float *W, int cin, int cout, float *wfwd, float *wdgrad);
hipError_t launch_repack_conv1(hipStream_t s, const float *W, int cout, float *w1);
hipError_t launch_bn_fold(hipStream_t s, const float *beta, const float *gamma, const float *mean, const float *istd,
                          int cout, float *bnp);

}  // namespace asr
