/* asr_hip.h - C ABI of libasr_hip.so, the MI355X (gfx950) implementation of the
 * audio<->sheet retrieval hot path of CPJKU/audio_sheet_retrieval.
 *
 * The reference has no FFI; its seam is Python-level: Theano
 * "compiled callables" on C-contiguous NCHW NumPy arrays plus Lasagne's
 * get/set_all_param_values.  Every entry point below names the reference
 * interface it replaces (paths relative to audio_sheet_retrieval/).  The
 * Python host code binds these with ctypes (audio_sheet_retrieval_amd/_lib.py,
 * INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns ASR_OK (0) or an ASR_ERR_* code; no exceptions
 *     cross the ABI; asr_last_error() gives the message of the last failure;
 *   - host buffers are caller-owned, C-contiguous; "_dev" variants take
 *     device pointers obtained from asr_dev_alloc (plain void*);
 *   - one asr_ctx is used from one thread at a time (the reference's compiled
 *     functions are not re-entrant either); distinct contexts are independent;
 *   - all work is enqueued on the context's own HIP stream; host-buffer
 *     variants return after the results are in the caller's memory,
 *     "_dev" variants return after enqueueing (call asr_sync()).
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_OK           0
#define ASR_ERR_INVALID  1   /* bad argument                                  */
#define ASR_ERR_HIP      2   /* HIP runtime / kernel launch failure           */
#define ASR_ERR_STATE    3   /* call order (e.g. embed before set_params)     */
#define ASR_ERR_COMM     4   /* RCCL failure                                  */
#define ASR_ERR_NOMEM    5

typedef struct asr_ctx asr_ctx;

/* Model hyper-parameters = the module constants of
 * models/mutopia_ccal_cont.py:23-51 (and _rsz.py) + geometry from
 * exp_configs/mutopia_full_aug.yaml:1-4. */
typedef struct asr_config {
    int32_t struct_size;    /* sizeof(asr_config): ABI check                   */
    int32_t device;         /* HIP device ordinal (reference: THEANO_FLAGS)    */
    int32_t num_filters;    /* num_filters_1: 12 (cont :74) or 24 (rsz :77)    */
    int32_t resize_view1;   /* 1 = rsz prepare: halve the sheet (rsz :179-185) */
    int32_t h1, w1;         /* raw sheet snippet, 160 x 200                    */
    int32_t h2, w2;         /* spectrogram excerpt, 92 x 42                    */
    int32_t dim_latent;     /* DIM_LATENT = 32 (:35); only 32 is supported     */
    int32_t max_chunk;      /* samples per internal launch, 0 = default        */
    float r1, r2, rT;       /* CCALayer regularisers (:42-43)                  */
    float alpha;            /* CCALayer running-average factor ALPHA (:49)     */
    float gamma;            /* ranking-loss margin GAMMA (:51)                 */
    float l2;               /* weight decay L2 (:39)                           */
    /* Gradient of MaxPool2DLayer (models/mutopia_ccal_cont.py:79,83,87,91,104,108,
     * 112,116) at windows with several equal maxima.  ASR_POOL_TIES_ALL (0, the
     * default): every element equal to the window maximum receives the upstream
     * gradient - Theano's CPU MaxPoolGrad, the path the reference's CPU runs take.
     * ASR_POOL_TIES_FIRST (1): only the first one in row-major order (what a
     * cuDNN-style pooling backward does; unverified offline).  Added in round 5:
     * a caller that passes the 64-byte struct of the earlier ABI gets 0.
     * "Equal" is decided on the float32 BatchNorm output y, BEFORE the ELU; the
     * reference pools float32 elu(y).  ELU is monotone (same maximum) but not
     * injective in float32 for y < 0 (about five neighbouring floats share an
     * image near y = -3; everything below about -17 is -1.0f): distinct y that tie
     * only after the ELU receive the gradient in the reference and not here.
     * The additional gradient is scaled by ELU'(y) = exp(y) <= 1; tied windows of
     * blank paper are bit-identical patches and tie either way (0 windows differ
     * on the synthetic and mostly-blank pages: oracle/train.py:elu_tie_deviation,
     * tests/test_oracle_train.py::test_ties_decided_before_or_after_the_float32_elu). */
    int32_t pool_ties;
} asr_config;
#define ASR_POOL_TIES_ALL   0
#define ASR_POOL_TIES_FIRST 1
#define ASR_CONFIG_SIZE_V1  64   /* sizeof(asr_config) before pool_ties existed */

/* ---- life cycle ------------------------------------------------------- */
/* build_model() (models/mutopia_ccal_cont.py:61-149): allocates the network
 * state on the device.  On failure *out is NULL and asr_last_error(NULL)
 * holds the message. */
int asr_create(const asr_config *cfg, asr_ctx **out);
void asr_destroy(asr_ctx *ctx);
const char *asr_last_error(const asr_ctx *ctx);
const char *asr_version(void);
int asr_sync(asr_ctx *ctx);                      /* wait for the ctx stream     */
/* The reference network is shape-agnostic (GlobalPoolLayer,
 * models/mutopia_ccal_cont.py:96,121; INPUT_SHAPE_1 is declared 120x200 but fed
 * 160x200).  Changes the raw input size of `view` (1|2): launch
 * plans are re-derived, activation buffers re-allocated on the next embed. */
int asr_set_input_size(asr_ctx *ctx, int view, int h, int w);

/* ---- parameters --------------------------------------------------------
 * lasagne.layers.get_all_param_values / set_all_param_values on the layer
 * list (utils/train_dcca_pool.py:395-401, run_eval.py:74-82,
 * retrieval_wrapper.py:27-29): a flat list of 97 float32 arrays,
 *   tower 1 blocks 1..9: W (O,I,kh,kw), beta, gamma, mean, inv_std   [0..44]
 *   tower 2 likewise                                                 [45..89]
 *   CCALayer U, V, mean1, mean2, S12, S11, S22 (layers/cca.py:69-77) [90..96]
 * The library repacks W (filter flip + MFMA fragment order) internally. */
int asr_param_count(const asr_ctx *ctx);
int asr_param_size(const asr_ctx *ctx, int index, int64_t *n_elements);
int asr_set_params(asr_ctx *ctx, const float *const *arrays, const int64_t *sizes, int n_arrays);
int asr_get_params(asr_ctx *ctx, float *const *arrays, const int64_t *sizes, int n_arrays);
/* refine_cca.py:104-107: cca_layer.mean1/mean2/U/V.set_value(...) */
int asr_set_cca(asr_ctx *ctx, const float *U, const float *V, const float *mean1, const float *mean2);

/* ---- embedding (deterministic forward) ----------------------------------
 * compute_v1_latent / compute_v2_latent (run_eval.py:92-95,
 * retrieval_wrapper.py:33-38) when out_kind = ASR_OUT_LATENT: tower -> CCALayer
 * deterministic branch (layers/cca.py:185-201) -> LengthNormLayer (:39-40);
 * the tower-only functions of refine_cca.py:86-89 when ASR_OUT_FEATURES.
 * View-1 input modes fold model.prepare (models/mutopia_ccal_cont.py:170-190)
 * into the first kernel:
 *   ASR_IN_F32_PREPARED  float32 (n,1,H,W) already /255 (and halved for rsz):
 *                        exactly what the reference's compiled function takes;
 *   ASR_IN_F32_RAW       float32 (n,1,h1,w1) holding 0..255 (pool output);
 *   ASR_IN_U8_RAW        uint8   (n,1,h1,w1) (what the servers pass,
 *                        audio_sheet_server.py:331,472).
 * out: (n, 32) float32. */
#define ASR_IN_F32_PREPARED 0
#define ASR_IN_F32_RAW      1
#define ASR_IN_U8_RAW       2
#define ASR_OUT_LATENT      0
#define ASR_OUT_FEATURES    1
int asr_embed_view1(asr_ctx *ctx, const void *x, int in_mode, int64_t n, int out_kind, float *out);
int asr_embed_view2(asr_ctx *ctx, const float *z, int64_t n, int out_kind, float *out);
int asr_embed_view1_dev(asr_ctx *ctx, const void *x_dev, int in_mode, int64_t n, int out_kind, float *out_dev);
int asr_embed_view2_dev(asr_ctx *ctx, const float *z_dev, int64_t n, int out_kind, float *out_dev);
/* compute_output of create_iter_functions (utils/train_dcca_pool.py:158): both views of the same n pairs in one call
 * -> [v1 latent, v2 latent].  Equivalent to asr_embed_view1 + asr_embed_view2. */
int asr_embed_both(asr_ctx *ctx, const void *x, int in_mode, const float *z, int64_t n, int out_kind,
                   float *out1, float *out2);

/* ---- ranking --------------------------------------------------------------
 * eval_retrieval (utils/train_dcca_pool.py:28-82): float64 cosine distances
 * (scipy cdist order, see oracle/retrieval.py) of lv1 (n1,dim) vs lv2 (n2,dim)
 * and, per query i, the 1-based rank of its correct item in a stable ascending
 * sort, computed by counting instead of sorting:
 *    k = n2 / n1_global if n2 > n1_global else 1   (:35)
 *    h = n1_global / n2 if n1_global > n2 else 1   (:36)
 *    correct(i) = { j : j / k == (i + query_offset) / h }
 *    d*   = min_{j in correct(i)} d_ij , j* the first index attaining it
 *    rank = 1 + #{ j : d_ij < d* } + #{ j < j* : d_ij == d* }
 * dstar[i] = d* (= diag(dists) for equal list sizes, :77), ties[i] = number of
 * other candidates with d_ij == d*.  ld1/ld2: row strides in floats (>= dim;
 * run_eval.py:160-162 clips dimensions by slicing columns).
 * query_offset / n1_global describe a shard of the query list (multi-GPU);
 * pass 0 / n1 for the single-device case.  Any output pointer may be NULL. */
int asr_rank(asr_ctx *ctx, const float *lv1, int64_t n1, int64_t ld1,
             const float *lv2, int64_t n2, int64_t ld2, int dim,
             int64_t query_offset, int64_t n1_global,
             int32_t *ranks, double *dstar, int32_t *ties);
int asr_rank_dev(asr_ctx *ctx, const float *lv1_dev, int64_t n1, int64_t ld1,
                 const float *lv2_dev, int64_t n2, int64_t ld2, int dim,
                 int64_t query_offset, int64_t n1_global,
                 int32_t *ranks_dev, double *dstar_dev, int32_t *ties_dev);

/* ---- top-k retrieval ---------------------------------------------------------
 * _retrieve_sheet_snippet_ids / _retrieve_perform_excerpt_ids
 * (audio_sheet_server.py:530-563): dists = cdist(db_codes, query, "cosine");
 * sorted_idx = argsort(dists)[:n_candidates].  For every query row of q
 * (n_q, dim) the k smallest float64 cosine distances to db (n_db, dim) in
 * ascending order, ties by ascending index (= NumPy's stable argsort);
 * idx[i*k + r] = index + idx_offset (int32; -1 and dist = +inf when n_db < k).
 * idx_offset makes the indices global when db is one shard of a larger pool
 * (multi-GPU: per-shard top-k, then a k-way merge of the gathered lists).
 * k <= 128, dim <= 64. */
int asr_topk(asr_ctx *ctx, const float *db, int64_t n_db, int64_t ld_db,
             const float *q, int64_t n_q, int64_t ld_q, int dim, int k,
             int64_t idx_offset, int32_t *idx, double *dist);
int asr_topk_dev(asr_ctx *ctx, const float *db_dev, int64_t n_db, int64_t ld_db,
                 const float *q_dev, int64_t n_q, int64_t ld_q, int dim, int k,
                 int64_t idx_offset, int32_t *idx_dev, double *dist_dev);

/* ---- resident code data base ------------------------------------------------------
 * The reference's server loads its code data base ONCE (audio_sheet_server.py:496-522 load_*_db_file ->
 * self.sheet_snippet_codes / self.spec_codes) and queries it for every incoming frame (:530-563
 * _retrieve_*_ids_for_*: cdist against the whole data base -> argsort[:n_candidates]); eval_retrieval
 * (utils/train_dcca_pool.py:40-74) likewise derives the top ranks AND the rank of the correct item from one
 * distance row.  An asr_db holds what such a data base needs beyond its rows, computed once instead of per
 * call: float64 row norms, their fp32 reciprocals and (32-d packed rows) a unit-length fp32 copy the MFMA
 * filter stages read.  codes_dev (n, ld) float32 stays OWNED BY THE CALLER and must outlive the handle; after
 * changing rows in place (e.g. an all-gather landing new shards in the same buffer) call asr_db_refresh.
 *   asr_topk_db_dev       = asr_topk_dev against the data base (same results, bit for bit)
 *   asr_rank_db_dev       = asr_rank_dev with the data base as the candidate side
 *   asr_topk_rank_db_dev  = both for the same queries from ONE walk over the pool: every 16 x 16 tile of fp32
 *                           cosines feeds the top-k candidate buffers and the rank counters (n >= 16384, 32-d
 *                           packed rows; otherwise the two passes run back to back).  Results identical to
 *                           the two separate calls.
 * query_offset / n1_global as in asr_rank_dev.  A handle belongs to the context that created it. */
typedef struct asr_db asr_db;
int asr_db_create(asr_ctx *ctx, const float *codes_dev, int64_t n, int64_t ld, int dim, asr_db **out);
int asr_db_refresh(asr_ctx *ctx, asr_db *db);
int asr_db_destroy(asr_ctx *ctx, asr_db *db);
int asr_db_size(asr_ctx *ctx, const asr_db *db, int64_t *n, int *dim);
int asr_topk_db_dev(asr_ctx *ctx, const asr_db *db, const float *q_dev, int64_t n_q, int64_t ld_q, int k,
                    int64_t idx_offset, int32_t *idx_dev, double *dist_dev);
int asr_rank_db_dev(asr_ctx *ctx, const asr_db *db, const float *lv1_dev, int64_t n1, int64_t ld1,
                    int64_t query_offset, int64_t n1_global, int32_t *ranks_dev, double *dstar_dev, int32_t *ties_dev);
int asr_topk_rank_db_dev(asr_ctx *ctx, const asr_db *db, const float *q_dev, int64_t n_q, int64_t ld_q, int k,
                         int64_t idx_offset, int32_t *idx_dev, double *dist_dev, int64_t query_offset,
                         int64_t n1_global, int32_t *ranks_dev, double *dstar_dev, int32_t *ties_dev);

/* A data base that is one SHARD of a larger pool (rows [item_offset, item_offset + n) of n2_global): the pieces of a
 * QUERY-sharded retrieval.  BASELINE configs[4] ("2M-snippet candidate pool sharded, RCCL all-gather 32-d embeddings,
 * global top-k") can gather the pool's embeddings (256 MB per step) or the QUERIES' (0.5 MB): every rank then searches
 * its own shard for all queries, the k-lists (n_q x k keys) are all-gathered and merged, the rank counters all-reduced
 * (ASR_DTYPE_I32) - the same integers, two orders of magnitude less traffic (bench.py --workload pool2m --exchange queries).
 *   asr_rank_dstar_db_dev : d* and the GLOBAL index j* of queries whose correct candidates (utils/train_dcca_pool.py:
 *                           35-36, 52-55) lie in this shard
 *   asr_topk_count_db_dev : top-k of the queries against the shard (indices + item_offset) and, d* / j* given, their
 *                           rank counters counts[n_q][3] = (#d < d*, #d == d*, #d == d* before j*) over the shard
 *   asr_topk_merge_dev    : the k smallest (distance, index) keys of queries [q_lo, q_lo + n_q) among n_parts lists
 *                           laid out [part][n_q_total][k] (n_parts * k <= 2048)
 *   asr_rank_finish_dev   : ranks = 1 + less + equal-before, ties = equal - 1 from summed counters */
int asr_rank_dstar_db_dev(asr_ctx *ctx, const asr_db *db, const float *q_dev, int64_t n_q, int64_t ld_q, int64_t item_offset,
                          int64_t n2_global, int64_t query_offset, int64_t n1_global, double *dstar_dev, int64_t *jstar_dev);
int asr_topk_count_db_dev(asr_ctx *ctx, const asr_db *db, const float *q_dev, int64_t n_q, int64_t ld_q, int k,
                          int64_t item_offset, int32_t *idx_dev, double *dist_dev, const double *dstar_dev,
                          const int64_t *jstar_dev, int32_t *counts_dev);
int asr_topk_merge_dev(asr_ctx *ctx, const int32_t *part_idx_dev, const double *part_dist_dev, int n_parts,
                       int64_t n_q_total, int64_t q_lo, int64_t n_q, int k, int32_t *idx_dev, double *dist_dev);
int asr_rank_finish_dev(asr_ctx *ctx, const int32_t *counts_dev, const double *dstar_dev, int64_t n, int32_t *ranks_dev,
                        double *dstar_out_dev, int32_t *ties_dev);

/* ---- CCA re-estimation -----------------------------------------------------
 * CCA(method='svd').fit(H1, H2) (utils/cca.py:25-53, 199-211) as driven by
 * refine_cca.py:100-107: float32 means and centring, second moments / (n-1)
 * rounded to float32, + r1/r2 on the diagonal in float64 (regularisers from
 * asr_config), then S11^-1/2, S22^-1/2, T, svd(T) in float64; outputs cast to
 * float32 exactly like refine_cca.py:104-107.  H1, H2: (n,32) float32 pre-CCA
 * tower outputs (ASR_OUT_FEATURES).  U, V: (32,32) row-major; columns are
 * defined up to a JOINT sign per canonical dimension (U[:,j], V[:,j]) ->
 * (-U[:,j], -V[:,j]), which leaves every cross-view cosine score unchanged.
 * coeffs (32 canonical correlations, descending, float64) may be NULL.
 * The result is NOT installed into the context: call asr_set_cca. */
int asr_cca_fit(asr_ctx *ctx, const float *H1, const float *H2, int64_t n,
                float *U, float *V, float *mean1, float *mean2, double *coeffs);
/* device variant: means_dev receives mean1 | mean2 (64 floats) */
int asr_cca_fit_dev(asr_ctx *ctx, const float *H1_dev, const float *H2_dev, int64_t n,
                    float *U_dev, float *V_dev, float *means_dev, double *coeffs_dev);

/* ---- piece identification on top of top-k ------------------------
 * detect_score / detect_performance (audio_sheet_server.py:213-300): n_samples sliding windows of one recording
 * (or one unrolled score) are embedded, each retrieves its n_candidates nearest data-base codes (asr_topk_dev),
 * the piece ids of all retrieved entries are counted and the top_k pieces by votes returned.
 *   asr_slice_windows_dev: out[i,0,r,c] = src[r0 + r, starts[i] + c]; src (rows, T) float32 row-major on the
 *     device, starts: n host int32 (np.linspace(0, T - win_w, n).astype(int), :217-218), out (n,1,win_h,win_w).
 *   asr_piece_vote_dev: idx_dev = the (n_q * k) int32 indices asr_topk_dev wrote (-1 entries ignored), ids_dev[j]
 *     = piece id of data-base entry j (sheet_snippet_ids, :520); pieces/counts (host, top_k): pieces ordered by
 *     votes descending, equal votes: larger piece id first (np.unique + argsort(counts)[::-1], :235-238, whose
 *     tie order NumPy leaves open); *n_out = number of pieces returned (<= top_k, only pieces with votes). */
int asr_slice_windows_dev(asr_ctx *ctx, const float *src_dev, int64_t rows, int64_t T, int r0, int win_h, int win_w,
                          const int32_t *starts, int n, float *out_dev);
int asr_piece_vote_dev(asr_ctx *ctx, const int32_t *idx_dev, int64_t n_idx, const int32_t *ids_dev, int64_t n_db,
                       int32_t n_pieces, int top_k, int32_t *pieces, int32_t *counts, int32_t *n_out);

/* ---- audio front-end -------------------------------------------------
 * The madmom chain the reference feeds its spectrogram tower with (tutorials/Embedding Tutorial.ipynb cell 28,
 * msmd.midi_parser.processor; audio_sheet_server.py:632,678 `processor.process(audio_file).T`):
 *   FramedSignalProcessor(frame_size, fps, origin='future'): frame i = samples[int(i*hop) : int(i*hop)+frame_size],
 *     zero padded past the end, hop = sample_rate / fps (fractional);
 *   magnitude STFT with `window` (np.hanning(frame_size); divided by 32767 for int16 input - the caller's choice);
 *   filterbank: filter f = fb_len[f] weights starting at FFT bin fb_start[f] (weights concatenated in fb_weights);
 *   out = log10(mul * x + add).
 * samples_dev: float32 mono samples on the device; out_dev: (n_frames, n_filters), or (n_filters, n_frames) when
 * transposed != 0 (the layout detect_score / asr_slice_windows_dev take).  window / fb_*: host arrays. */
int asr_spectrogram_dev(asr_ctx *ctx, const float *samples_dev, int64_t n_samples, int frame_size, double hop,
                        const float *window, const int32_t *fb_start, const int32_t *fb_len, const float *fb_weights,
                        int n_filters, float mul, float add, int64_t n_frames, int transposed, float *out_dev);

/* ---- alignment: distance matrix + DTW -----------------------------
 * compute_alignment / align_pydtw (utils/alignment.py:120-186) on dtw_by_dist (utils/dtw_by_dist.py:5-34,76-91):
 * dists = cdist(a, b, "cosine") in float64 (same arithmetic as asr_rank), accumulated cost
 * D1[i,j] += min(D0[i,j], D0[i,j+1], D0[i+1,j]) as an anti-diagonal wavefront, traceback with argmin over
 * (diagonal, up, left), first minimum winning.  a_dev (n_a,dim), b_dev (n_b,dim) float32 on the device, rows of the
 * cost matrix = a.  The reference transposes when the matrix is wider than tall (:13-15) - the caller passes the
 * longer sequence as `a` (audio_sheet_retrieval_amd/alignment.py does).  Outputs (host): dists n_a*n_b doubles (may
 * be NULL), path_a / path_b (capacity n_a + n_b) = _traceback's (p, q), *path_len, *min_dist = D1[-1,-1] / (n_a+n_b). */
int asr_dtw_dev(asr_ctx *ctx, const float *a_dev, int64_t n_a, const float *b_dev, int64_t n_b, int dim,
                double *dists, int32_t *path_a, int32_t *path_b, int32_t *path_len, double *min_dist);

/* ---- training-pool batch assembly on the device ----------------------
 * AudioScoreRetrievalPool.__getitem__ (utils/data_pools.py:127-228): every sample is a window of one strip (unrolled
 * score image or spectrogram) of a pool that stays resident on the device, with the augmentations of
 * exp_configs/mutopia_full_aug.yaml - sheet_scaling (cv2.resize INTER_NEAREST), system_translation, onset_translation,
 * spec_padding (np.pad mode="edge").  The host draws the random numbers in the reference's order and reduces a sample
 * to 9 doubles desc[i] = {off, stride, y0, sy, ymax, x0, sx, xmax, xadd}; then
 *   out[i,0,y,x] = src[off + clamp(floor((y0+y)*sy), 0, ymax) * stride + xadd + clamp(floor((x0+x)*sx), 0, xmax)].
 * src_dev: the pool (src_floats float32, strips concatenated); out_dev: (n,1,out_h,out_w) float32. */
int asr_gather_windows_dev(asr_ctx *ctx, const float *src_dev, int64_t src_floats, const double *desc, int n,
                           int out_h, int out_w, float *out_dev);

/* ---- host-buffer pipeline ---------------------------------------------------------------------
 * The evaluation loop of run_eval.py:102-108,174 (and the per-call copies of utils/batch_iterators.py:90-109) for a
 * stream of host batches: batch k = n (sheet, spectrogram) pairs x[k] / z[k] in host memory (in_mode as in
 * asr_embed_view1) -> both towers -> all-pairs ranking of the batch's n queries against its n candidates ->
 * ranks[k] (n int32), and optionally dstar[k] (n float64), ties[k] (n int32), lv1[k] / lv2[k] (n x 32 float32) back
 * in host memory (the array pointers, or single entries, may be NULL).  Inputs are double-buffered on the device:
 * the host-to-device copy of batch k+1 and the device-to-host copy of batch k-1 run on their own copy streams while
 * batch k computes.  Host buffers from asr_host_alloc (page-locked) make the copies truly asynchronous; ordinary
 * memory works, slower.  Returns when every output is in place. */
int asr_host_alloc(asr_ctx *ctx, size_t bytes, void **hptr);
int asr_host_free(asr_ctx *ctx, void *hptr);
int asr_eval_batches(asr_ctx *ctx, const void *const *x, int in_mode, const float *const *z, int n_batches, int64_t n,
                     int32_t *const *ranks, double *const *dstar, int32_t *const *ties, float *const *lv1,
                     float *const *lv2);

/* Self-check of the kernel autotuner.  With ASR_TUNE_VERIFY=1 in the environment the first embed call of each tower
 * runs every candidate schedule / tiling of every conv block on one deterministic input and compares its output with
 * the first candidate's (all evaluate the same fp32 FMA chains, so they must agree to <= 1e-5).  Returns the number of
 * comparisons made, of mismatches, and the largest deviation seen. */
int asr_debug_tune_report(asr_ctx *ctx, int32_t *checked, int32_t *mismatches, float *max_diff);

/* ---- multi-GPU: one process and one context per GPU -------------------
 * The reference is single-device; these entry points are what a sharded deployment binds.  Pairs are sharded
 * by contiguous ranges, rank r of `world` holding [r*n_local, (r+1)*n_local).
 *   retrieval: embedding needs no communication; asr_rank_sharded_dev all-gathers the candidate-side
 *     embeddings (n_local x 32 floats per rank) and ranks this rank's queries against all of them - integer
 *     results identical to the single-GPU asr_rank on the concatenated data;
 *   training: asr_train_step / asr_burn_in shard the batch: each rank passes its rows - batch/world of them, or,
 *     after asr_train_set_global_batch(n), its contiguous share of n rows (the first n % world ranks hold one row
 *     more), so that the reference's BATCH_SIZE = 100 (models/mutopia_ccal_cont.py:26) trains on all 100 rows on 3 or
 *     8 GPUs.  The per-channel BatchNorm sums are all-reduced forward and backward, the 32-d tower outputs are
 *     all-gathered and every rank evaluates CCALayer + loss on the full batch, the parameter gradients are
 *     all-reduced before Adam: every rank ends the step with the same parameters the single-GPU step over the
 *     whole batch produces (float32 summation order aside).
 * Transport: RCCL (asr_comm_unique_id on rank 0, hand the 128 bytes to every rank, asr_comm_init on all; the
 * collectives are enqueued on the context's stream), or caller-supplied host callbacks (asr_comm_init_custom:
 * the library drains the stream, then calls the callback, which must return with the result in place - used by
 * the tests to run two ranks on one GPU, and by deployments with another transport).
 * Initialise the communicator before asr_train_begin. */
#define ASR_COMM_ID_BYTES 128
#define ASR_DTYPE_F32 0
#define ASR_DTYPE_F64 1
#define ASR_DTYPE_I32 2
typedef int (*asr_allreduce_fn)(void *user, void *buf_dev, int64_t count, int dtype);       /* in-place sum */
typedef int (*asr_allgather_fn)(void *user, const void *send_dev, void *recv_dev, int64_t bytes_per_rank);
int asr_comm_unique_id(void *id_out /* ASR_COMM_ID_BYTES */);
int asr_comm_init(asr_ctx *ctx, int rank, int world, const void *unique_id);
int asr_comm_init_custom(asr_ctx *ctx, int rank, int world, asr_allreduce_fn allreduce, asr_allgather_fn allgather,
                         void *user);
int asr_comm_destroy(asr_ctx *ctx);
int asr_comm_info(asr_ctx *ctx, int *rank, int *world);
/* Collectives this context has issued through its communicator since the last reset: counts[0..3] = all-reduce calls,
 * all-reduce payload bytes, all-gather calls, all-gather bytes sent per rank.  What `bench.py --workload train`
 * reports as the exchange cost of one data-parallel update (the reference's single-device step,
 * utils/train_dcca_pool.py:203-205, has none). */
int asr_comm_stats(asr_ctx *ctx, int64_t *counts, int reset);
/* Time spent in those collectives: while enabled every collective is bracketed by two HIP events on the stream it is
 * enqueued on (RCCL) or timed on the host clock (callbacks).  Each call waits for the context's streams, returns the
 * summed duration (ms) and the number of collectives since the previous call, resets both and sets the switch to
 * `enable`.  ms / calls may be NULL.  (New with the multi-GPU path; the reference has no collectives.) */
int asr_comm_timing(asr_ctx *ctx, int enable, double *ms, int64_t *calls);
/* File the RCCL entry points of this context's communicator were bound from ("" without an RCCL communicator).  The
 * library is looked up as ASR_RCCL_LIB, $ROCM_PATH/lib/librccl.so, /opt/rocm/lib/librccl.so, then by bare name. */
int asr_comm_library(asr_ctx *ctx, char *path, int cap);
/* The communicator's two collectives on caller-owned device buffers, enqueued on the context's stream like the
 * library's own exchange steps: in-place sum over all ranks of `count` values (ASR_DTYPE_F32 / _F64) and the
 * concatenation of every rank's bytes_per_rank bytes in rank order.  What the host code uses for the hit counters,
 * the max-over-ranks timing and the epoch decisions fit() takes on rank 0 (utils/train_dcca_pool.py:391-411,
 * 492-520 run in one process in the reference).  With no communicator (or a world of one) the all-reduce leaves
 * the buffer unchanged and the all-gather copies send -> recv. */
int asr_comm_allreduce_dev(asr_ctx *ctx, void *buf_dev, int64_t count, int dtype);
int asr_comm_allgather_dev(asr_ctx *ctx, const void *send_dev, void *recv_dev, int64_t bytes_per_rank);
/* lv1_dev, lv2_dev: (n_local,32) device; lv2_all_dev: (world*n_local,32) device workspace that receives the
 * gathered candidates; ranks/dstar/ties: n_local device outputs as in asr_rank_dev. */
int asr_rank_sharded_dev(asr_ctx *ctx, const float *lv1_dev, const float *lv2_dev, int64_t n_local,
                         float *lv2_all_dev, int32_t *ranks, double *dstar, int32_t *ties);

/* ---- training ------------------------------------------------------------------
 * The compiled functions of create_iter_functions (utils/train_dcca_pool.py:85-167):
 *   asr_train_step  = iter_funcs['train'](X1, X2) -> [loss, corr]      (:154)
 *   asr_valid_loss  = iter_funcs['valid'](X1, X2) -> [loss]            (:155)
 * asr_train_step runs, entirely on the device: both towers with batch statistics
 * (BatchNormLayer train branch + EMA of mean / inv_std), the CCALayer
 * train branch (layers/cca.py:91-182; the four eigh's and their EighGrad as
 * float64 Jacobi iterations in one workgroup; the layer's running U, V, means,
 * S12, S11, S22 are overwritten like its default_updates do), LengthNormLayer, the
 * contrastive cos loss (models/objectives.py:30-69, gamma from asr_config), the
 * full backward pass, the L2 penalty l2 * sum p^2 over W, beta, gamma (:141-142)
 * and lasagne.updates.adam (beta1 .9, beta2 .999, eps 1e-8, step
 * lr*sqrt(1-b2^t)/(1-b1^t); models/mutopia_ccal_cont.py:158-162).
 * x1: (batch,1,H1,W1) float32 ALREADY prepared (the reference applies model.prepare
 * in the batch iterator, utils/batch_iterators.py:220-221); x2: (batch,1,h2,w2).
 * loss = ranking loss + L2 penalty (as Theano reports it), corr: 32 canonical
 * correlations (ascending, layers/cca.py:161-164).  lr is passed per call
 * (fit() mutates the shared learning rate, :343,520,525).
 * asr_train_begin allocates the device state for batches up to batch_size and
 * starts Adam from zero moments; asr_get/set_opt_state expose (m, v, t) for the
 * refinement restarts of fit() (:396,515-516): flat float32 arrays over the
 * tower parameters in the order of asr_get_params (asr_opt_state_size values;
 * non-trainable slots are zero). */
int asr_train_begin(asr_ctx *ctx, int batch_size);
int asr_train_end(asr_ctx *ctx);
/* Data parallel (asr_comm_init*), batches that are not a multiple of the world size: the following asr_train_step /
 * asr_burn_in / asr_compute_gradients calls carry this rank's rows [lo, hi) of ONE batch of n_global rows, lo / hi by
 * the contiguous rule (rank r: lo = r*(n/world) + min(r, n%world); the first n%world ranks hold one row more); a call
 * whose `batch` is not that count fails.  BatchNorm means, the CCALayer covariances and the loss run over exactly
 * n_global rows - nothing is padded or dropped - which is what iter_funcs['train'](X1, X2) computes on one device
 * (utils/train_dcca_pool.py:154, :203-205).  n_global = 0 restores the default (batch * world, equal shards).  Needs
 * asr_train_begin with batch_size >= ceil(n_global / world). */
int asr_train_set_global_batch(asr_ctx *ctx, int64_t n_global);
int asr_train_step(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float lr,
                   float *loss, float *corr);
int asr_train_step_dev(asr_ctx *ctx, const float *x1_dev, const float *x2_dev, int64_t batch, float lr,
                       float *loss, float *corr);
int asr_valid_loss(asr_ctx *ctx, const float *x1, const float *x2, int64_t n, float *loss);
/* objectives() of the model module (models/mutopia_ccal_cont.py:152-155) =
 * get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69): the loss asr_train_step,
 * asr_compute_gradients and asr_valid_loss evaluate.  Default: weight 1, gamma = asr_config.gamma, symmetric 0 - what
 * both models bind.  symmetric = 1 adds the transposed direction (:53-65: the same hinge with the views swapped);
 * weight scales loss and gradients (:67).  Takes effect from the next call. */
int asr_set_objective(asr_ctx *ctx, float weight, float gamma, int symmetric);
/* iter_funcs['init_cca'](X1, X2) of create_iter_functions(init_cca=True) (utils/train_dcca_pool.py:160-162), the
 * burn-in pass of pretrain() (:170-182): one TRAIN-mode forward whose only side effects are the default updates of the
 * graph - BatchNorm running mean / inv_std and the CCALayer running means, covariances, U, V.  No gradients, no Adam
 * step, Adam's t unchanged.  lv1 / lv2 (batch,32) receive the train-mode outputs (may be NULL).  Needs
 * asr_train_begin. */
int asr_burn_in(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float *lv1, float *lv2);
/* iter_funcs['compute_gradients'](X1, X2) (utils/train_dcca_pool.py:164-167: theano.function(input_vars, all_grads)):
 * the gradient of the train loss (ranking loss + l2 * sum p^2) wrt every trainable parameter, WITHOUT the Adam step.
 * grads: flat float32 array of asr_opt_state_size values in the order of asr_get_params (non-trainable slots zero).
 * Like every function compiled from the train-mode graph it applies the graph's default updates (BatchNorm and
 * CCALayer running values move exactly as in asr_burn_in); Adam's moments and t stay untouched.  loss may be NULL. */
int asr_compute_gradients(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float *grads, int64_t n,
                          float *loss);
int asr_opt_state_size(asr_ctx *ctx, int64_t *n);
int asr_get_opt_state(asr_ctx *ctx, float *m, float *v, int64_t n, int32_t *t);
int asr_set_opt_state(asr_ctx *ctx, const float *m, const float *v, int64_t n, int32_t t);
/* test aids: intermediate tensors of the last training step
 * (kind: 0 raw conv output z, 1 block input x, 2 batch stats [mu|inv_std], 3 H,
 * 4 dL/dH, 5 train-mode embedding, 6 gradient of parameter `index`, 7 device
 * value of parameter `index`, 8 [loss | corr], 9 pooled block `index`: the raw conv output of the element every 2x2
 * pooling window selected - (batch, H/2, W/2, C), the FIRST maximal element; 10 pooled block `index`: the set of
 * window elements whose BatchNorm output equals the window maximum, as the backward pass decides it - (batch, H/2, W/2,
 * C) floats holding 4-bit sets, bit 2*dy+dx; derived on request from z, the batch statistics and the scale / shift
 * the forward pass kept); and the
 * CCALayer + loss stage
 * alone on host arrays (cca_in/cca_out: U V mean1 mean2 S12 S11 S22, 5184 floats). */
int asr_debug_train_tensor(asr_ctx *ctx, int kind, int view, int index, int64_t batch,
                           float *out, int64_t cap, int64_t *n_out);
int asr_cca_train_debug(asr_ctx *ctx, const float *H1, const float *H2, int64_t batch,
                        const float *cca_in, float *cca_out, float *loss_corr,
                        float *lv1, float *lv2, float *dH1, float *dH2);

/* ---- device memory (plain pointers; library-owned allocations) ---------- */
int asr_dev_alloc(asr_ctx *ctx, size_t bytes, void **dptr);
int asr_dev_free(asr_ctx *ctx, void *dptr);
int asr_dev_upload(asr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int asr_dev_download(asr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- per-kernel timing (HIP events on the ctx stream) -------------------
 * The reference only prints wall-clock "ups" (utils/train_dcca_pool.py:221-231);
 * bench.py needs per-kernel durations for the roofline line. */
int asr_profile_enable(asr_ctx *ctx, int on);
/* symbol != NULL / "": only launches of that kernel symbol are bracketed by events (the measurement then costs two
 * events per step instead of two per kernel); NULL or "" = every kernel */
int asr_profile_filter(asr_ctx *ctx, const char *symbol);
int asr_profile_reset(asr_ctx *ctx);
int asr_profile_count(asr_ctx *ctx);
int asr_profile_get(asr_ctx *ctx, int index, char *name, int name_cap,
                    int64_t *launches, double *total_ms, double *flops, double *bytes);
/* kernel symbol (as rocprofv3 --kernel-trace prints it) behind record `index`;
 * empty for labels that cover several kernels */
int asr_profile_symbol(asr_ctx *ctx, int index, char *symbol, int symbol_cap);

/* ---- debugging aid for the parity tests ---------------------------------
 * Copies the NHWC activation of conv block `block` (0..7, after BN/ELU and the
 * max-pool where the block has one) of tower `view` (1|2), as left by the last
 * embed call, for its first n samples; *h,*w,*c receive its geometry.  `out`
 * may be NULL to query the geometry only. */
int asr_debug_activation(asr_ctx *ctx, int view, int block, int64_t n,
                         float *out, int *h, int *w, int *c);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
