// C-ABI layer, retrieval side: eval_retrieval ranks, top-k, the resident code data base and its shard pieces, window
// slicing, DTW alignment, the audio front-end, window gathering, the piece vote (include/asr_hip.h for the contract and
// the reference interfaces each entry point replaces).  Context and shared helpers: asr_ctx.h.
#include "asr_ctx.h"

extern "C" {

int asr_rank_dev(asr_ctx *ctx, const float *lv1, int64_t n1, int64_t ld1, const float *lv2, int64_t n2, int64_t ld2,
                 int dim, int64_t query_offset, int64_t n1_global, int32_t *ranks, double *dstar, int32_t *ties) {
    int rc = rank_check(ctx, n1, ld1, n2, ld2, dim, query_offset, n1_global);
    if (rc != ASR_OK) return rc;
    if (n1 == 0) return ASR_OK;
    if (!lv1 || !lv2) return fail(ctx, ASR_ERR_INVALID, "rank: NULL embeddings");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    rc = ensure_norms(ctx, n1, n2);
    if (rc != ASR_OK) return rc;
    rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    // utils/train_dcca_pool.py:35-36 (py2 integer division)
    const int64_t k = n2 > n1_global ? n2 / n1_global : 1;
    const int64_t h = n1_global > n2 ? n1_global / n2 : 1;
    if ((query_offset + n1 - 1) / h * k >= n2)
        return fail(ctx, ASR_ERR_INVALID, "rank: query %lld has no correct candidate (n2=%lld)",
                    (long long)(query_offset + n1 - 1), (long long)n2);
    {
        ProfScope ps(ctx, "row_norms", 0, 2.0 * dim * (double)(n1 + n2), 4.0 * dim * (double)(n1 + n2));
        ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, lv1, n1, ld1, dim, ctx->norm1));
        ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, lv2, n2, ld2, dim, ctx->norm2));
    }
    {
        ProfScope ps(ctx, "rank", 0, 2.0 * dim * (double)n1 * (double)n2, 4.0 * dim * (double)(n1 + n2));
        const size_t need = asr::rank_workspace_bytes(n1, n2);           // shares the top-k scratch buffer
        if (need > ctx->topk_ws_bytes) {
            if (ctx->topk_ws) ASR_HIP(ctx, hipFree(ctx->topk_ws));
            ctx->topk_ws = nullptr; ctx->topk_ws_bytes = 0;
            ASR_HIP(ctx, hipMalloc(&ctx->topk_ws, need));
            ctx->topk_ws_bytes = need;
        }
        ASR_HIP(ctx, asr::launch_rank(ctx->stream, lv1, ctx->norm1, n1, ld1, lv2, ctx->norm2, n2, ld2, dim,
                                      query_offset, k, h, ranks, dstar, ties, ctx->topk_ws));
    }
    return mark_main(ctx);
}

int asr_rank(asr_ctx *ctx, const float *lv1, int64_t n1, int64_t ld1, const float *lv2, int64_t n2, int64_t ld2,
             int dim, int64_t query_offset, int64_t n1_global, int32_t *ranks, double *dstar, int32_t *ties) {
    int rc = rank_check(ctx, n1, ld1, n2, ld2, dim, query_offset, n1_global);
    if (rc != ASR_OK) return rc;
    if (n1 == 0) return ASR_OK;
    if (!lv1 || !lv2) return fail(ctx, ASR_ERR_INVALID, "rank: NULL embeddings");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // one growable device scratch for the host-buffer variant (five hipMalloc / hipFree pairs per call were half of
    // eval_retrieval's 0.33 ms at n = 2000)
    const size_t b1 = ((size_t)n1 * ld1 * sizeof(float) + 255) & ~(size_t)255, b2 = ((size_t)n2 * ld2 * sizeof(float) + 255) & ~(size_t)255;
    const size_t bd = ((size_t)n1 * sizeof(double) + 255) & ~(size_t)255, bi = ((size_t)n1 * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t need = b1 + b2 + bd + 2 * bi;
    if (need > ctx->rank_io_bytes) {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
        if (ctx->rank_io) ASR_HIP(ctx, hipFree(ctx->rank_io));
        ctx->rank_io = nullptr; ctx->rank_io_bytes = 0;
        ASR_HIP(ctx, hipMalloc(&ctx->rank_io, need));
        ctx->rank_io_bytes = need;
    }
    char *base = (char *)ctx->rank_io;
    float *d1 = (float *)base, *d2 = (float *)(base + b1);
    double *dd = (double *)(base + b1 + b2);
    int32_t *dr = (int32_t *)(base + b1 + b2 + bd), *dt = (int32_t *)(base + b1 + b2 + bd + bi);
#define RANK_HIP(call)                                                                                  \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            (void)hipStreamSynchronize(ctx->stream);                                                    \
            return fail(ctx, ASR_ERR_HIP, "asr_rank: %s failed: %s", #call, hipGetErrorString(e__));    \
        }                                                                                               \
    } while (0)
    RANK_HIP(hipMemcpyAsync(d1, lv1, (size_t)n1 * ld1 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    RANK_HIP(hipMemcpyAsync(d2, lv2, (size_t)n2 * ld2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = asr_rank_dev(ctx, d1, n1, ld1, d2, n2, ld2, dim, query_offset, n1_global, dr, dd, dt);
    if (rc != ASR_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if (ranks) RANK_HIP(hipMemcpyAsync(ranks, dr, (size_t)n1 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (dstar) RANK_HIP(hipMemcpyAsync(dstar, dd, (size_t)n1 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (ties) RANK_HIP(hipMemcpyAsync(ties, dt, (size_t)n1 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    RANK_HIP(hipStreamSynchronize(ctx->stream));
#undef RANK_HIP
    return ASR_OK;
}

static int ensure_topk_tickets(asr_ctx *ctx);

int asr_topk_dev(asr_ctx *ctx, const float *db, int64_t n_db, int64_t ld_db, const float *q, int64_t n_q, int64_t ld_q,
                 int dim, int k, int64_t idx_offset, int32_t *idx, double *dist) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n_db < 0 || n_q < 0 || dim < 1 || dim > 64 || ld_db < dim || ld_q < dim || k < 1 || k > 128)
        return fail(ctx, ASR_ERR_INVALID, "topk: bad sizes n_db=%lld n_q=%lld dim=%d k=%d (k <= 128)", (long long)n_db,
                    (long long)n_q, dim, k);
    if (n_q == 0) return ASR_OK;
    if (!q || !idx || !dist || (n_db > 0 && !db)) return fail(ctx, ASR_ERR_INVALID, "topk: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_norms(ctx, n_q, n_db > 0 ? n_db : 1);
    if (rc != ASR_OK) return rc;
    rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    // A large pool of packed 32-d rows is searched the way a resident data base is (asr_db_*): one pass derives the
    // float64 norms - which this call needs anyway - AND the unit-length copy the filter reads (0.02 ms at 250 k rows,
    // 0.1 ms at 2 M), and the seeded, one-compare-per-four-distances filter does the rest: 1024 queries x 250 k codes
    // 0.97 -> 0.55 ms, 64 x 2 M 0.72 -> 0.40 against the filter on raw rows.  A caller that keeps its pool creates an
    // asr_db and skips the pass.
    const bool as_db = dim == 32 && ld_db == 32 && ld_q == 32 && n_db >= 16384 && (reinterpret_cast<uintptr_t>(db) & 15) == 0 &&
                       !(getenv("ASR_TOPK_UNIT") && getenv("ASR_TOPK_UNIT")[0] == '0');
    float *unit = nullptr, *rn = nullptr;
    if (as_db) {
        const size_t n_pad = (size_t)((n_db + 3) & ~(int64_t)3), need_f = (size_t)n_db * 32 + n_pad;
        if (need_f > ctx->unit_ws_floats) {
            rc = sync_all(ctx);
            if (rc != ASR_OK) return rc;
            if (ctx->unit_ws) ASR_HIP(ctx, hipFree(ctx->unit_ws));
            ctx->unit_ws = nullptr; ctx->unit_ws_floats = 0;
            ASR_HIP(ctx, hipMalloc((void **)&ctx->unit_ws, need_f * sizeof(float)));
            ctx->unit_ws_floats = need_f;
        }
        unit = ctx->unit_ws;
        rn = ctx->unit_ws + (size_t)n_db * 32;
    }
    {
        ProfScope ps(ctx, "row_norms", 0, 2.0 * dim * (double)(n_q + n_db), (as_db ? 8.0 : 4.0) * dim * (double)(n_q + n_db));
        ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, q, n_q, ld_q, dim, ctx->norm1));
        if (as_db) ASR_HIP(ctx, asr::launch_db_prepare(ctx->stream, db, n_db, ctx->norm2, rn, unit));
        else ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, db, n_db, ld_db, dim, ctx->norm2));
    }
    {
        ProfScope ps(ctx, "topk", 0, 2.0 * dim * (double)n_q * (double)n_db, 4.0 * dim * (double)n_db * (double)n_q);
        const size_t need = asr::topk_workspace_bytes(n_db, n_q, k, as_db, false);
        if (need > ctx->topk_ws_bytes) {
            if (ctx->topk_ws) ASR_HIP(ctx, hipFree(ctx->topk_ws));
            ctx->topk_ws = nullptr; ctx->topk_ws_bytes = 0;
            ASR_HIP(ctx, hipMalloc(&ctx->topk_ws, need));
            ctx->topk_ws_bytes = need;
        }
        int rc2 = ensure_topk_tickets(ctx);                    // (the state block of the sort-free exact refine)
        if (rc2 != ASR_OK) return rc2;
        ASR_HIP(ctx, asr::launch_topk(ctx->stream, db, ctx->norm2, n_db, ld_db, q, ctx->norm1, n_q, ld_q, dim, k,
                                      idx_offset, idx, dist, ctx->topk_ws, unit, rn, nullptr, ctx->topk_tickets));
    }
    return mark_main(ctx);
}

// ---- resident code data base (audio_sheet_server.py:496-522: the server loads its code data base once and queries it
// per frame, :530-563) -------------------------------------------------------------------------------------------------
struct asr_db {
    asr_ctx *owner = nullptr;
    const float *codes = nullptr;       // caller-owned device rows (n, ld)
    int64_t n = 0, ld = 0;
    int dim = 0;
    double *norms = nullptr;            // float64 row norms
    float *rn = nullptr;                // fp32 reciprocal norms, zero padded to a multiple of 4
    float *unit = nullptr;              // unit-length copy (32-d packed rows only)
};

static int ensure_topk_tickets(asr_ctx *ctx) {
    if (ctx->topk_tickets) return ASR_OK;
    // [0, 1024): last-arriver tickets (ASR_TOPK_FOLD); [1024, 3072): survivor counts and scan flags of the sort-free refine
    ASR_HIP(ctx, hipMalloc((void **)&ctx->topk_tickets, 4096 * sizeof(unsigned)));
    ASR_HIP(ctx, hipMemsetAsync(ctx->topk_tickets, 0, 4096 * sizeof(unsigned), ctx->stream));
    return ASR_OK;
}

static int grow_topk_ws(asr_ctx *ctx, size_t need) {
    if (need > ctx->topk_ws_bytes) {
        int rc = sync_all(ctx);                      // an earlier launch may still read the old buffer
        if (rc != ASR_OK) return rc;
        if (ctx->topk_ws) ASR_HIP(ctx, hipFree(ctx->topk_ws));
        ctx->topk_ws = nullptr; ctx->topk_ws_bytes = 0;
        ASR_HIP(ctx, hipMalloc(&ctx->topk_ws, need));
        ctx->topk_ws_bytes = need;
    }
    return ASR_OK;
}

static int db_check(asr_ctx *ctx, const asr_db *db, const char *who) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!db || db->owner != ctx) return fail(ctx, ASR_ERR_INVALID, "%s: not a data base of this context", who);
    return ASR_OK;
}

int asr_db_refresh(asr_ctx *ctx, asr_db *db) {
    int rc = db_check(ctx, db, "db_refresh");
    if (rc != ASR_OK) return rc;
    if (db->n == 0) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    ProfScope ps(ctx, "db_prepare", 0, 2.0 * db->dim * (double)db->n, (db->unit ? 8.0 : 4.0) * db->dim * (double)db->n);
    if (db->unit)
        ASR_HIP(ctx, asr::launch_db_prepare(ctx->stream, db->codes, db->n, db->norms, db->rn, db->unit));
    else
        ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, db->codes, db->n, db->ld, db->dim, db->norms));
    return mark_main(ctx);
}

int asr_db_create(asr_ctx *ctx, const float *codes_dev, int64_t n, int64_t ld, int dim, asr_db **out) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!out) return fail(ctx, ASR_ERR_INVALID, "db_create: NULL output");
    *out = nullptr;
    if (n < 0 || dim < 1 || dim > 64 || ld < dim || (n > 0 && !codes_dev))
        return fail(ctx, ASR_ERR_INVALID, "db_create: bad sizes n=%lld ld=%lld dim=%d", (long long)n, (long long)ld, dim);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    std::unique_ptr<asr_db> db(new asr_db());
    db->owner = ctx; db->codes = codes_dev; db->n = n; db->ld = ld; db->dim = dim;
    const bool packed32 = dim == 32 && ld == 32 && (reinterpret_cast<uintptr_t>(codes_dev) & 15) == 0;
    const size_t n_pad = (size_t)((n + 3) & ~(int64_t)3);
    hipError_t e = hipMalloc((void **)&db->norms, std::max<size_t>(1, (size_t)n) * sizeof(double));
    if (e == hipSuccess && packed32) e = hipMalloc((void **)&db->rn, std::max<size_t>(4, n_pad) * sizeof(float));
    if (e == hipSuccess && packed32) e = hipMalloc((void **)&db->unit, std::max<size_t>(1, (size_t)n) * 32 * sizeof(float));
    if (e != hipSuccess) {
        if (db->norms) hipFree(db->norms);
        if (db->rn) hipFree(db->rn);
        if (db->unit) hipFree(db->unit);
        return fail(ctx, ASR_ERR_HIP, "db_create: %s", hipGetErrorString(e));
    }
    int rc = asr_db_refresh(ctx, db.get());
    if (rc != ASR_OK) {
        hipFree(db->norms);
        if (db->rn) hipFree(db->rn);
        if (db->unit) hipFree(db->unit);
        return rc;
    }
    *out = db.release();
    return ASR_OK;
}

int asr_db_destroy(asr_ctx *ctx, asr_db *db) {
    if (!db) return ASR_OK;
    int rc = db_check(ctx, db, "db_destroy");
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    rc = sync_all(ctx);
    if (db->norms) hipFree(db->norms);
    if (db->rn) hipFree(db->rn);
    if (db->unit) hipFree(db->unit);
    db->owner = nullptr;
    delete db;
    return rc;
}

int asr_db_size(asr_ctx *ctx, const asr_db *db, int64_t *n, int *dim) {
    int rc = db_check(ctx, db, "db_size");
    if (rc != ASR_OK) return rc;
    if (n) *n = db->n;
    if (dim) *dim = db->dim;
    return ASR_OK;
}

// shared front of the three query entry points: argument checks, the queries' norms
// norms_later: the caller's launcher computes the query norms itself (asr::launch_topk, norm_q_pending)
static int db_query_begin(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, const char *who,
                          bool norms_later = false) {
    int rc = db_check(ctx, db, who);
    if (rc != ASR_OK) return rc;
    if (n_q < 0 || ld_q < db->dim) return fail(ctx, ASR_ERR_INVALID, "%s: bad sizes n_q=%lld ld_q=%lld", who, (long long)n_q, (long long)ld_q);
    if (n_q == 0) return ASR_OK;
    if (!q) return fail(ctx, ASR_ERR_INVALID, "%s: NULL queries", who);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    rc = ensure_norms(ctx, n_q, 1);
    if (rc != ASR_OK) return rc;
    rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    if (norms_later) return ASR_OK;
    ProfScope ps(ctx, "row_norms", 0, 2.0 * db->dim * (double)n_q, 4.0 * db->dim * (double)n_q);
    ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, q, n_q, ld_q, db->dim, ctx->norm1));
    return ASR_OK;
}

// global indices are returned as int32: offset + row must stay below 2^31
static int db_offset_fits(asr_ctx *ctx, const asr_db *db, int64_t offset, const char *what) {
    if (offset < 0 || offset + db->n > (int64_t)INT32_MAX)
        return fail(ctx, ASR_ERR_INVALID, "%s: index offset %lld + %lld rows does not fit the int32 indices returned", what,
                    (long long)offset, (long long)db->n);
    return ASR_OK;
}

int asr_topk_db_dev(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, int k, int64_t idx_offset,
                    int32_t *idx, double *dist) {
    int rc = db_query_begin(ctx, db, q, n_q, ld_q, "topk_db", true);
    if (rc != ASR_OK || n_q == 0) return rc;
    if (k < 1 || k > 128 || !idx || !dist) return fail(ctx, ASR_ERR_INVALID, "topk_db: k=%d (1..128) / NULL output", k);
    if ((rc = db_offset_fits(ctx, db, idx_offset, "topk_db")) != ASR_OK) return rc;
    ProfScope ps(ctx, "topk", 0, 2.0 * db->dim * (double)n_q * (double)db->n, 4.0 * db->dim * (double)db->n);
    rc = grow_topk_ws(ctx, asr::topk_workspace_bytes(db->n, n_q, k, db->unit != nullptr, false));
    if (rc != ASR_OK) return rc;
    if ((rc = ensure_topk_tickets(ctx)) != ASR_OK) return rc;
    // (the query norms are formed inside the call: by the seeding kernel where that path runs)
    ASR_HIP(ctx, asr::launch_topk(ctx->stream, db->codes, db->norms, db->n, db->ld, q, ctx->norm1, n_q, ld_q, db->dim, k,
                                  idx_offset, idx, dist, ctx->topk_ws, db->unit, db->rn, ctx->norm1, ctx->topk_tickets));
    return mark_main(ctx);
}

static int db_rank_geometry(asr_ctx *ctx, const asr_db *db, int64_t n1, int64_t query_offset, int64_t n1_global,
                            int64_t *k, int64_t *h) {
    if (n1_global < 1 || query_offset < 0 || query_offset + n1 > n1_global)
        return fail(ctx, ASR_ERR_INVALID, "rank_db: queries [%lld, %lld) outside the %lld of the job", (long long)query_offset,
                    (long long)(query_offset + n1), (long long)n1_global);
    *k = db->n > n1_global ? db->n / n1_global : 1;            // utils/train_dcca_pool.py:35-36 (py2 integer division)
    *h = n1_global > db->n ? n1_global / db->n : 1;
    if (db->n < 1 || (query_offset + n1 - 1) / *h * *k >= db->n)
        return fail(ctx, ASR_ERR_INVALID, "rank_db: query %lld has no correct candidate (n2=%lld)",
                    (long long)(query_offset + n1 - 1), (long long)db->n);
    return ASR_OK;
}

int asr_rank_db_dev(asr_ctx *ctx, const asr_db *db, const float *lv1, int64_t n1, int64_t ld1, int64_t query_offset,
                    int64_t n1_global, int32_t *ranks, double *dstar, int32_t *ties) {
    int rc = db_query_begin(ctx, db, lv1, n1, ld1, "rank_db");
    if (rc != ASR_OK || n1 == 0) return rc;
    if (!ranks || !dstar || !ties) return fail(ctx, ASR_ERR_INVALID, "rank_db: NULL output");
    int64_t k, h;
    if ((rc = db_rank_geometry(ctx, db, n1, query_offset, n1_global, &k, &h)) != ASR_OK) return rc;
    ProfScope ps(ctx, "rank", 0, 2.0 * db->dim * (double)n1 * (double)db->n, 4.0 * db->dim * (double)(n1 + db->n));
    rc = grow_topk_ws(ctx, asr::rank_workspace_bytes(n1, db->n));
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, asr::launch_rank(ctx->stream, lv1, ctx->norm1, n1, ld1, db->codes, db->norms, db->n, db->ld, db->dim,
                                  query_offset, k, h, ranks, dstar, ties, ctx->topk_ws, db->rn));
    return mark_main(ctx);
}

int asr_topk_rank_db_dev(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, int k,
                         int64_t idx_offset, int32_t *idx, double *dist, int64_t query_offset, int64_t n1_global,
                         int32_t *ranks, double *dstar, int32_t *ties) {
    int rc = db_query_begin(ctx, db, q, n_q, ld_q, "topk_rank_db");
    if (rc != ASR_OK || n_q == 0) return rc;
    if (k < 1 || k > 128 || !idx || !dist || !ranks || !dstar || !ties)
        return fail(ctx, ASR_ERR_INVALID, "topk_rank_db: k=%d (1..128) / NULL output", k);
    if ((rc = db_offset_fits(ctx, db, idx_offset, "topk_rank_db")) != ASR_OK) return rc;
    int64_t kk, hh;
    if ((rc = db_rank_geometry(ctx, db, n_q, query_offset, n1_global, &kk, &hh)) != ASR_OK) return rc;
    if (db->unit && ld_q == 32 && asr::topk_rank_fusable(db->n, kk)) {
        // one walk over the pool's item tiles feeds the top-k candidate buffers and the rank counters
        ProfScope ps(ctx, "topk_rank", 0, 2.0 * 32 * (double)n_q * (double)db->n, 128.0 * (double)db->n);
        rc = grow_topk_ws(ctx, asr::topk_workspace_bytes(db->n, n_q, k, true, true));
        if (rc != ASR_OK) return rc;
        ASR_HIP(ctx, asr::launch_topk_rank_db(ctx->stream, db->codes, db->unit, db->norms, db->n, q, ctx->norm1, n_q, k,
                                              idx_offset, idx, dist, query_offset, kk, hh, ranks, dstar, ties,
                                              ctx->topk_ws));
        return mark_main(ctx);
    }
    {
        ProfScope ps(ctx, "topk", 0, 2.0 * db->dim * (double)n_q * (double)db->n, 4.0 * db->dim * (double)db->n);
        rc = grow_topk_ws(ctx, asr::topk_workspace_bytes(db->n, n_q, k, db->unit != nullptr, false));
        if (rc != ASR_OK) return rc;
        if ((rc = ensure_topk_tickets(ctx)) != ASR_OK) return rc;
        ASR_HIP(ctx, asr::launch_topk(ctx->stream, db->codes, db->norms, db->n, db->ld, q, ctx->norm1, n_q, ld_q, db->dim,
                                      k, idx_offset, idx, dist, ctx->topk_ws, db->unit, db->rn, nullptr, ctx->topk_tickets));
    }
    {
        ProfScope ps(ctx, "rank", 0, 2.0 * db->dim * (double)n_q * (double)db->n, 4.0 * db->dim * (double)(n_q + db->n));
        rc = grow_topk_ws(ctx, asr::rank_workspace_bytes(n_q, db->n));
        if (rc != ASR_OK) return rc;
        ASR_HIP(ctx, asr::launch_rank(ctx->stream, q, ctx->norm1, n_q, ld_q, db->codes, db->norms, db->n, db->ld, db->dim,
                                      query_offset, kk, hh, ranks, dstar, ties, ctx->topk_ws, db->rn));
    }
    return mark_main(ctx);
}

// ---- a data base that is one SHARD of a larger pool: what each rank of a query-sharded retrieval computes ------------
int asr_rank_dstar_db_dev(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, int64_t item_offset,
                          int64_t n2_global, int64_t query_offset, int64_t n1_global, double *dstar, int64_t *jstar) {
    int rc = db_query_begin(ctx, db, q, n_q, ld_q, "rank_dstar_db");
    if (rc != ASR_OK || n_q == 0) return rc;
    if (db->dim != 32 || db->ld != 32 || ld_q != 32) return fail(ctx, ASR_ERR_INVALID, "rank_dstar_db: 32-d packed rows only");
    if (!dstar || !jstar || n1_global < 1 || n2_global < db->n + item_offset || item_offset < 0 || query_offset < 0 ||
        query_offset + n_q > n1_global)
        return fail(ctx, ASR_ERR_INVALID, "rank_dstar_db: bad geometry");
    const int64_t kk = n2_global > n1_global ? n2_global / n1_global : 1, hh = n1_global > n2_global ? n1_global / n2_global : 1;
    const int64_t first = (query_offset / hh) * kk, last = ((query_offset + n_q - 1) / hh) * kk + kk;
    if (first < item_offset || std::min(last, n2_global) > item_offset + db->n)
        return fail(ctx, ASR_ERR_INVALID, "rank_dstar_db: the correct candidates [%lld, %lld) of these queries are not all in "
                    "this shard [%lld, %lld)", (long long)first, (long long)last, (long long)item_offset,
                    (long long)(item_offset + db->n));
    ProfScope ps(ctx, "rank_dstar", 0, 64.0 * (double)n_q * (double)kk, 128.0 * (double)n_q * (double)kk);
    ASR_HIP(ctx, asr::launch_rank_dstar(ctx->stream, q, ctx->norm1, n_q, db->codes, db->norms, db->n, item_offset, n2_global,
                                        query_offset, kk, hh, dstar, jstar));
    return mark_main(ctx);
}

int asr_topk_count_db_dev(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, int k,
                          int64_t item_offset, int32_t *idx, double *dist, const double *dstar, const int64_t *jstar,
                          int32_t *counts) {
    int rc = db_query_begin(ctx, db, q, n_q, ld_q, "topk_count_db");
    if (rc != ASR_OK || n_q == 0) return rc;
    if (k < 1 || k > 128 || !idx || !dist || !dstar || !jstar || !counts)
        return fail(ctx, ASR_ERR_INVALID, "topk_count_db: k=%d (1..128) / NULL argument", k);
    if (!db->unit || ld_q != 32 || db->n < 16384)
        return fail(ctx, ASR_ERR_INVALID, "topk_count_db: needs a data base of >= 16384 packed 32-d rows");
    if ((rc = db_offset_fits(ctx, db, item_offset, "topk_count_db")) != ASR_OK) return rc;
    ProfScope ps(ctx, "topk_count", 0, 2.0 * 32 * (double)n_q * (double)db->n, 128.0 * (double)db->n);
    rc = grow_topk_ws(ctx, asr::topk_workspace_bytes(db->n, n_q, k, true, true));
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, asr::launch_topk_count_db(ctx->stream, db->codes, db->unit, db->norms, db->n, q, ctx->norm1, n_q, k,
                                           item_offset, idx, dist, dstar, jstar, counts, ctx->topk_ws));
    return mark_main(ctx);
}

int asr_topk_merge_dev(asr_ctx *ctx, const int32_t *part_idx, const double *part_dist, int n_parts, int64_t n_q_total,
                       int64_t q_lo, int64_t n_q, int k, int32_t *idx, double *dist) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n_q == 0) return ASR_OK;
    if (!part_idx || !part_dist || !idx || !dist || n_parts < 1 || k < 1 || (int64_t)n_parts * k > 2048 || q_lo < 0 ||
        q_lo + n_q > n_q_total)
        return fail(ctx, ASR_ERR_INVALID, "topk_merge: %d lists of %d keys (product <= 2048), queries [%lld, %lld) of %lld",
                    n_parts, k, (long long)q_lo, (long long)(q_lo + n_q), (long long)n_q_total);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, asr::launch_topk_merge(ctx->stream, part_idx, part_dist, n_parts, n_q_total, q_lo, n_q, k, idx, dist));
    return mark_main(ctx);
}

int asr_rank_finish_dev(asr_ctx *ctx, const int32_t *counts, const double *dstar, int64_t n, int32_t *ranks,
                        double *dstar_out, int32_t *ties) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n == 0) return ASR_OK;
    if (!counts || !dstar) return fail(ctx, ASR_ERR_INVALID, "rank_finish: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, asr::launch_rank_finish(ctx->stream, counts, dstar, n, ranks, dstar_out, ties));
    return mark_main(ctx);
}

int asr_slice_windows_dev(asr_ctx *ctx, const float *src_dev, int64_t rows, int64_t T, int r0, int win_h, int win_w,
                          const int32_t *starts, int n, float *out_dev) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n < 0 || win_h < 1 || win_w < 1 || r0 < 0 || r0 + win_h > rows || win_w > T)
        return fail(ctx, ASR_ERR_INVALID, "slice_windows: window %dx%d at row %d does not fit %lld x %lld", win_h, win_w,
                    r0, (long long)rows, (long long)T);
    if (n == 0) return ASR_OK;
    if (!src_dev || !starts || !out_dev) return fail(ctx, ASR_ERR_INVALID, "slice_windows: NULL argument");
    for (int i = 0; i < n; ++i)
        if (starts[i] < 0 || starts[i] + win_w > T)
            return fail(ctx, ASR_ERR_INVALID, "slice_windows: start %d = %d outside [0, %lld]", i, starts[i],
                        (long long)(T - win_w));
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int32_t *d_starts = nullptr;
    ASR_HIP(ctx, hipMalloc((void **)&d_starts, (size_t)n * sizeof(int32_t)));
    hipError_t e = hipMemcpyAsync(d_starts, starts, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = asr::launch_slice_windows(ctx->stream, src_dev, T, r0, win_h, win_w, d_starts, n, out_dev);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_starts);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "slice_windows: %s", hipGetErrorString(e));
    return mark_main(ctx);
}

int asr_dtw_dev(asr_ctx *ctx, const float *a_dev, int64_t n_a, const float *b_dev, int64_t n_b, int dim, double *dists,
                int32_t *path_a, int32_t *path_b, int32_t *path_len, double *min_dist) {
    if (!ctx || !path_a || !path_b || !path_len) return ASR_ERR_INVALID;
    if (n_a < 1 || n_b < 1 || dim < 1 || dim > 64 || n_a > 100000 || n_b > 100000 || (n_a + 1) * (n_b + 1) > (1ll << 31))
        return fail(ctx, ASR_ERR_INVALID, "dtw: bad sizes n_a=%lld n_b=%lld dim=%d", (long long)n_a, (long long)n_b, dim);
    if (!a_dev || !b_dev) return fail(ctx, ASR_ERR_INVALID, "dtw: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = ensure_norms(ctx, n_a, n_b);
    if (rc != ASR_OK) return rc;
    rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    const size_t cells = (size_t)(n_a + 1) * (n_b + 1);
    double *D = nullptr, *dout = nullptr;
    int32_t *pbuf = nullptr;
    auto cleanup = [&]() { (void)hipFree(D); (void)hipFree(dout); (void)hipFree(pbuf); };
    hipError_t e = hipMalloc((void **)&D, (cells + 1) * sizeof(double));
    if (e == hipSuccess && dists) e = hipMalloc((void **)&dout, (size_t)n_a * n_b * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&pbuf, (size_t)(2 * (n_a + n_b) + 1) * sizeof(int32_t));
    int32_t *pi = pbuf, *pj = pbuf ? pbuf + (n_a + n_b) : nullptr, *plen = pbuf ? pbuf + 2 * (n_a + n_b) : nullptr;
    if (e == hipSuccess) {
        ProfScope ps(ctx, "dtw", 0, 2.0 * dim * (double)n_a * (double)n_b, 8.0 * (double)cells);
        e = asr::launch_row_norms(ctx->stream, a_dev, n_a, dim, dim, ctx->norm1);
        if (e == hipSuccess) e = asr::launch_row_norms(ctx->stream, b_dev, n_b, dim, dim, ctx->norm2);
        if (e == hipSuccess)
            e = asr::launch_dtw(ctx->stream, a_dev, ctx->norm1, n_a, dim, b_dev, ctx->norm2, n_b, dim, dim, D, dout, pi, pj,
                                plen, D + cells);
    }
    int32_t len = 0;
    double md = 0.0;
    if (e == hipSuccess) e = hipMemcpyAsync(&len, plen, sizeof len, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&md, D + cells, sizeof md, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && dists)
        e = hipMemcpyAsync(dists, dout, (size_t)n_a * n_b * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && len > 0) {
        std::vector<int32_t> ri(len), rj(len);
        e = hipMemcpy(ri.data(), pi, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(rj.data(), pj, (size_t)len * sizeof(int32_t), hipMemcpyDeviceToHost);
        for (int32_t k = 0; k < len && e == hipSuccess; ++k) {      // the kernel walks from the end to the origin
            path_a[k] = ri[len - 1 - k];
            path_b[k] = rj[len - 1 - k];
        }
    }
    cleanup();
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "dtw: %s", hipGetErrorString(e));
    *path_len = len;
    if (min_dist) *min_dist = md;
    return mark_main(ctx);
}

int asr_spectrogram_dev(asr_ctx *ctx, const float *samples_dev, int64_t n_samples, int frame_size, double hop,
                        const float *window, const int32_t *fb_start, const int32_t *fb_len, const float *fb_weights,
                        int n_filters, float mul, float add, int64_t n_frames, int transposed, float *out_dev) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n_samples < 0 || n_frames < 0 || frame_size < 64 || frame_size > 8192 || (frame_size & (frame_size - 1)) ||
        !(hop > 0.0) || n_filters < 1 || n_filters > 4096)
        return fail(ctx, ASR_ERR_INVALID, "spectrogram: bad sizes (frame_size must be a power of two in [64, 8192])");
    if (n_frames == 0) return ASR_OK;
    if (!samples_dev || !window || !fb_start || !fb_len || !fb_weights || !out_dev)
        return fail(ctx, ASR_ERR_INVALID, "spectrogram: NULL argument");
    std::vector<int32_t> off(n_filters);
    int64_t total_w = 0;
    int max_bin = 0;
    for (int f = 0; f < n_filters; ++f) {
        if (fb_start[f] < 0 || fb_len[f] < 0 || fb_start[f] + fb_len[f] > frame_size / 2)
            return fail(ctx, ASR_ERR_INVALID, "spectrogram: filter %d covers bins [%d, %d) outside [0, %d)", f, fb_start[f],
                        fb_start[f] + fb_len[f], frame_size / 2);
        off[f] = (int32_t)total_w;
        total_w += fb_len[f];
        max_bin = std::max(max_bin, fb_start[f] + fb_len[f]);
    }
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    char *buf = nullptr;
    const size_t b_win = (size_t)frame_size * 4, b_i = (size_t)n_filters * 4, b_w = (size_t)std::max<int64_t>(total_w, 1) * 4;
    ASR_HIP(ctx, hipMalloc((void **)&buf, b_win + 3 * b_i + b_w));
    float *d_win = (float *)buf;
    int32_t *d_start = (int32_t *)(buf + b_win), *d_len = d_start + n_filters, *d_off = d_len + n_filters;
    float *d_w = (float *)(buf + b_win + 3 * b_i);
    hipError_t e = hipMemcpyAsync(d_win, window, b_win, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_start, fb_start, b_i, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len, fb_len, b_i, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), b_i, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && total_w) e = hipMemcpyAsync(d_w, fb_weights, (size_t)total_w * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        ProfScope ps(ctx, "spectrogram", 0, 4.0 * frame_size * (double)max_bin * (double)n_frames,
                     4.0 * hop * (double)n_frames);
        e = asr::launch_spectrogram(ctx->stream, samples_dev, n_samples, d_win, frame_size, hop, max_bin, d_start, d_len,
                                    d_off, d_w, n_filters, mul, add, out_dev, n_frames, transposed);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "spectrogram: %s", hipGetErrorString(e));
    return mark_main(ctx);
}

int asr_gather_windows_dev(asr_ctx *ctx, const float *src_dev, int64_t src_floats, const double *desc, int n, int out_h,
                           int out_w, float *out_dev) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n < 0 || out_h < 1 || out_w < 1) return fail(ctx, ASR_ERR_INVALID, "gather_windows: bad sizes");
    if (n == 0) return ASR_OK;
    if (!src_dev || !desc || !out_dev) return fail(ctx, ASR_ERR_INVALID, "gather_windows: NULL argument");
    for (int i = 0; i < n; ++i) {       // every reachable source index must lie inside the pool buffer
        const double *d = desc + (size_t)i * 9;
        const double lo = d[0] + d[8], hi = d[0] + d[4] * d[1] + d[8] + d[7];
        if (!(d[1] >= 1 && d[4] >= 0 && d[7] >= 0 && lo >= 0 && hi < (double)src_floats && d[3] > 0 && d[6] > 0))
            return fail(ctx, ASR_ERR_INVALID, "gather_windows: descriptor %d addresses [%g, %g] outside the %lld-float pool",
                        i, lo, hi, (long long)src_floats);
    }
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    double *d_desc = nullptr;
    ASR_HIP(ctx, hipMalloc((void **)&d_desc, (size_t)n * 9 * sizeof(double)));
    hipError_t e = hipMemcpyAsync(d_desc, desc, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = asr::launch_gather_windows(ctx->stream, src_dev, d_desc, n, out_h, out_w, out_dev);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_desc);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "gather_windows: %s", hipGetErrorString(e));
    return mark_main(ctx);
}

int asr_piece_vote_dev(asr_ctx *ctx, const int32_t *idx_dev, int64_t n_idx, const int32_t *ids_dev, int64_t n_db,
                       int32_t n_pieces, int top_k, int32_t *pieces, int32_t *counts, int32_t *n_out) {
    if (!ctx || !pieces || !counts || !n_out) return ASR_ERR_INVALID;
    if (n_idx < 0 || n_db < 0 || n_pieces < 1 || top_k < 1 || top_k > 1024)
        return fail(ctx, ASR_ERR_INVALID, "piece_vote: bad sizes n_idx=%lld n_db=%lld n_pieces=%d top_k=%d",
                    (long long)n_idx, (long long)n_db, n_pieces, top_k);
    if (n_idx > 0 && (!idx_dev || !ids_dev)) return fail(ctx, ASR_ERR_INVALID, "piece_vote: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int32_t *ws = nullptr;
    ASR_HIP(ctx, hipMalloc((void **)&ws, ((size_t)n_pieces + 2 * (size_t)top_k) * sizeof(int32_t)));
    int32_t *d_piece = ws + n_pieces, *d_count = d_piece + top_k;
    hipError_t e;
    {
        ProfScope ps(ctx, "piece_vote", 0, 0.0, 8.0 * (double)n_idx);
        e = asr::launch_piece_vote(ctx->stream, idx_dev, n_idx, ids_dev, n_db, n_pieces, top_k, ws, d_piece, d_count);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(pieces, d_piece, (size_t)top_k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(counts, d_count, (size_t)top_k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(ws);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "piece_vote: %s", hipGetErrorString(e));
    int m = 0;
    while (m < top_k && pieces[m] >= 0) ++m;
    *n_out = m;
    return mark_main(ctx);
}

int asr_topk(asr_ctx *ctx, const float *db, int64_t n_db, int64_t ld_db, const float *q, int64_t n_q, int64_t ld_q,
             int dim, int k, int64_t idx_offset, int32_t *idx, double *dist) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n_db < 0 || n_q < 0 || ld_db < 1 || ld_q < 1 || k < 1) return fail(ctx, ASR_ERR_INVALID, "topk: bad sizes");
    if (n_q == 0) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    float *d_db = nullptr, *d_q = nullptr;
    int32_t *d_idx = nullptr;
    double *d_dist = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_db); (void)hipFree(d_q); (void)hipFree(d_idx); (void)hipFree(d_dist); };
#define TOPK_HIP(call)                                                                               \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            cleanup();                                                                               \
            return fail(ctx, ASR_ERR_HIP, "asr_topk: %s failed: %s", #call, hipGetErrorString(e__)); \
        }                                                                                            \
    } while (0)
    TOPK_HIP(hipMalloc((void **)&d_db, (size_t)std::max<int64_t>(n_db, 1) * ld_db * sizeof(float)));
    TOPK_HIP(hipMalloc((void **)&d_q, (size_t)n_q * ld_q * sizeof(float)));
    TOPK_HIP(hipMalloc((void **)&d_idx, (size_t)n_q * k * sizeof(int32_t)));
    TOPK_HIP(hipMalloc((void **)&d_dist, (size_t)n_q * k * sizeof(double)));
    if (n_db > 0) TOPK_HIP(hipMemcpyAsync(d_db, db, (size_t)n_db * ld_db * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    TOPK_HIP(hipMemcpyAsync(d_q, q, (size_t)n_q * ld_q * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = asr_topk_dev(ctx, d_db, n_db, ld_db, d_q, n_q, ld_q, dim, k, idx_offset, d_idx, d_dist);
    if (rc != ASR_OK) { cleanup(); return rc; }
    TOPK_HIP(hipMemcpyAsync(idx, d_idx, (size_t)n_q * k * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    TOPK_HIP(hipMemcpyAsync(dist, d_dist, (size_t)n_q * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TOPK_HIP(hipStreamSynchronize(ctx->stream));
#undef TOPK_HIP
    cleanup();
    return ASR_OK;
}

}  // extern "C"
