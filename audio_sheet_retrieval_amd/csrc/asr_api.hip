// C-ABI layer of libasr_hip.so (see include/asr_hip.h for the contract and the
// reference interfaces each entry point replaces).
#include "../../include/asr_hip.h"
#include "asr_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>     // types only: the library is resolved with dlopen in asr_comm_init
#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct LayerGeom {      // one conv block of a tower
    int cin, cout, k, pool;
    int H, W;           // input resolution
    int OH, OW;         // output resolution (after the pool, if any)
};

struct ProfRec {
    std::string name;
    std::string symbol;               // kernel symbol (rocprofv3 naming) the label maps to
    double flops = 0, bytes = 0;      // per launch (algorithmic)
    int64_t launches = 0;
    double total_ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Tower {
    LayerGeom g[9];
    asr::ConvPlan plan[9];          // valid for blocks 1..7 (index = block)
    float *w_dev[9] = {};           // packed weights
    float *bn_dev[9] = {};          // [3][COUTP]
    float *act[8] = {};             // outputs of blocks 0..7 for one chunk
    size_t act_floats[8] = {};      // per sample
    int in_h = 0, in_w = 0;         // network-resolution input
    bool tuned = false;             // launch plans chosen by timing (autotune)
    bool fuse1 = false;             // block 1 evaluated inside the block-2 kernel (its activation never hits HBM)
};

// device-resident training state (asr_train_begin)
struct TrainTower {
    float *x[9] = {};               // block inputs: x[0] prepared input, x[b] = output of block b-1
    float *z[9] = {};               // raw conv outputs (z[8]: 1x1 conv)
    float *stats[9] = {};           // batch [mu | inv_std]
    float *zsel[9] = {};            // pooled blocks: raw value of each pooling window's selected element (ASR_TRAIN_ZSEL)
    float *wdgrad[9] = {};          // data-gradient weight fragments (blocks 1..7)
    asr::ConvPlan fplan[9], dplan[9];
    asr::WgradPlan wplan[9];
    float *dz = nullptr;            // gradient wrt the raw conv output of the current block
    // weight gradients on a stream of their own (they are MFMA-bound, the BatchNorm backward of the next block that
    // the main stream continues with is HBM-bound): a second dz buffer, "dz of parity p written" / "wgrad done with dz of
    // parity p" events
    float *dz2 = nullptr;
    hipStream_t wstream = nullptr;
    hipEvent_t e_dz[2] = {nullptr, nullptr}, e_wg[2] = {nullptr, nullptr};
    float *dA = nullptr, *dB = nullptr;   // gradients wrt block outputs (rotating)
    float *H = nullptr, *dH = nullptr, *lv = nullptr;
    double *partial = nullptr;      // reduction partials (BN stats/bwd, tail, conv1 wgrad)
    float *wpartial = nullptr;      // wgrad per-block partials
    size_t wpartial_floats = 0;
    double *sums = nullptr;
};

struct TrainState {
    int B = 0;                      // batch size the buffers were sized for
    int64_t ptotal = 0;
    std::vector<int64_t> poff;      // offsets of the 97 arrays in the flat buffers
    float *pmaster = nullptr, *pgrad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    unsigned char *mask = nullptr;
    int adam_t = 0;
    TrainTower tw[2];
    void *cca_ws = nullptr;
    float *loss_dev = nullptr;      // [0] ranking loss, [1..32] corr
    double *l2_dev = nullptr;
    float *lvv[2] = {nullptr, nullptr};   // deterministic embeddings for asr_valid_loss
    hipEvent_t cca_done = nullptr;
    asr::RepackDesc *repack_dev = nullptr;   // table of repack_all_kernel: every layout derived from the master
    int n_repack = 0;
    bool master_dirty = false;      // device master newer than the host mirror
    // data-parallel training (asr_comm_*): tower outputs, train-mode embeddings and dL/dH of the FULL batch
    float *Hg[2] = {nullptr, nullptr}, *dHg[2] = {nullptr, nullptr}, *lvg[2] = {nullptr, nullptr};
    float *Hpad[2] = {nullptr, nullptr};    // all-gather target when the shards differ in size: [world][largest shard][32]
    int world = 1;                  // ranks the buffers were sized for
    int64_t global_batch = 0;       // asr_train_set_global_batch: rows of the whole batch (0: batch * world, equal shards)
};

// Collective transport of one context: RCCL (resolved at run time) or host callbacks supplied by the caller.
struct Comm {
    int rank = 0, world = 1;
    bool force = false;             // ASR_COMM_FORCE=1: route world-1 collectives through the transport (tests)
    int64_t n_allreduce = 0, b_allreduce = 0, n_allgather = 0, b_allgather = 0;      // asr_comm_stats
    asr_allreduce_fn ar = nullptr;
    asr_allgather_fn ag = nullptr;
    void *user = nullptr;
    void *dl = nullptr;
    ncclComm_t nccl = nullptr;
    ncclResult_t (*pAllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pAllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*pCommDestroy)(ncclComm_t) = nullptr;
    const char *(*pGetErrorString)(ncclResult_t) = nullptr;
};

// Worker threads that move a caller's (pageable) array into a page-locked staging slot: one thread copies at
// ~10 GB/s, less than the towers consume (47.5 KB/pair x 285 k pairs/s = 13.5 GB/s with uint8 sheets, three times
// that with float sheets).  The calling thread takes pieces too; run() returns when the whole range is in place.
struct CopyPool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const char *src = nullptr;
    char *dst = nullptr;
    size_t bytes = 0, piece = 1 << 20;
    std::atomic<size_t> next{0};
    uint64_t gen = 0;
    int busy = 0;
    bool stop = false;

    explicit CopyPool(int n_workers) {
        for (int i = 0; i < n_workers; ++i) workers.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
    void pieces() {
        for (;;) {
            const size_t off = next.fetch_add(piece);
            if (off >= bytes) return;
            memcpy(dst + off, src + off, std::min(piece, bytes - off));
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            pieces();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--busy == 0) cv_done.notify_all();
            }
        }
    }
    void run(void *d, const void *s, size_t n) {
        if (workers.empty() || n < 4 * piece) { memcpy(d, s, n); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = (char *)d; src = (const char *)s; bytes = n;
            next.store(0);
            busy = (int)workers.size();
            ++gen;
        }
        cv_work.notify_all();
        pieces();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return busy == 0; });
    }
};

}  // namespace

struct asr_ctx {
    asr_config cfg{};
    int num_cus = 256;
    hipStream_t stream = nullptr;             // main stream: ranking, CCA fit, copies
    hipStream_t vstream[2] = {nullptr, nullptr};   // one per tower: the training step overlaps the two towers
    hipStream_t estream[2] = {nullptr, nullptr};   // embedding: the main stream (default) or the tower streams
    hipStream_t tstream[2] = {nullptr, nullptr};   // training step: tower streams, or the main stream when data parallel
    bool in_train = false;                         // which set the profiler's events go on
    bool wino_stale = false;                       // training moved the weights: Winograd-domain copies need a refresh
    hipEvent_t vdone[2] = {nullptr, nullptr};      // last embed of each tower
    bool vpending[2] = {false, false};
    hipEvent_t main_done = nullptr;                // last consumer (rank / cca_fit) on the main stream
    bool main_pending = false;
    bool single_stream = false;
    std::unique_ptr<TrainState> train;
    std::unique_ptr<Comm> comm;
    int tune_checked = 0, tune_bad = 0;       // ASR_TUNE_VERIFY=1: candidates compared with the first one / mismatches
    float tune_max_diff = 0.0f;
    asr::Exchange exch{};                     // what the kernel launchers see of `comm`
    int chunk = 256;
    bool params_set = false;
    std::vector<std::vector<float>> params;   // host mirror, reference order
    std::vector<std::vector<int64_t>> pshape;
    Tower tw[2];
    float *cca_dev = nullptr;                 // U[1024] V[1024] mean1[32] mean2[32]
    void *in_stage[2] = {nullptr, nullptr};   // chunk input staging per tower (host-buffer API)
    size_t in_stage_bytes = 0;
    float *out_stage[2] = {nullptr, nullptr}; // chunk x 32
    double *norm1 = nullptr, *norm2 = nullptr;
    int64_t norm_cap1 = 0, norm_cap2 = 0;
    void *cca_ws = nullptr;                   // CCA-fit partial sums
    size_t cca_ws_bytes = 0;
    void *topk_ws = nullptr;                  // top-k filter stage: fp32 reciprocal norms + candidate lists
    size_t topk_ws_bytes = 0;
    float *unit_ws = nullptr;                 // asr_topk_dev on a large pool: unit-length copy + reciprocal norms of the
    size_t unit_ws_floats = 0;                // pool, rebuilt per call (what an asr_db keeps)
    void *rank_io = nullptr;                  // asr_rank (host buffers): embeddings in, ranks / d* / ties out
    size_t rank_io_bytes = 0;
    // asr_eval_batches: double-buffered host-to-host pipeline (inputs, embeddings, ranking outputs; copy streams)
    struct Pipe {
        hipStream_t h2d = nullptr, d2h = nullptr;
        void *in1[2] = {nullptr, nullptr};
        float *in2[2] = {nullptr, nullptr}, *lv1[2] = {nullptr, nullptr}, *lv2[2] = {nullptr, nullptr};
        int32_t *ranks[2] = {nullptr, nullptr}, *ties[2] = {nullptr, nullptr};
        double *dstar[2] = {nullptr, nullptr};
        hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, out[2] = {nullptr, nullptr};
        size_t b1 = 0, b2 = 0;
        int64_t n = 0;
    } pipe;
    // host-buffer embedding (asr_embed_view1/2/both): the caller's array is cut into granules that travel through a ring
    // of page-locked staging slots and device input buffers - staging copy (CopyPool), H2D on a copy stream and the
    // towers of successive granules overlap; all embeddings return in one D2H at the end
    struct HostPipe {
        static constexpr int NSLOT = 3;
        hipStream_t h2d = nullptr;
        void *pin[NSLOT] = {nullptr, nullptr, nullptr}, *dev[NSLOT] = {nullptr, nullptr, nullptr};
        size_t slot_bytes = 0;
        hipEvent_t copied[NSLOT] = {nullptr, nullptr, nullptr};     // H2D into dev[s] finished (pin[s] is free again)
        hipEvent_t consumed[NSLOT] = {nullptr, nullptr, nullptr};   // the tower has read dev[s]
        bool used[NSLOT] = {false, false, false};
        float *out_dev = nullptr;
        size_t out_floats = 0;
        std::unique_ptr<CopyPool> pool;
        int granule = 0, granule_first = 0;     // samples per granule: granule_first, doubling up to granule
        bool staged = false;                    // ASR_HOST_STAGE as it stood when the pipe was set up (latched)
    } hpipe;
    int last_n[2] = {0, 0};                   // samples of the last chunk per tower (debug)
    bool profiling = false;
    std::string prof_filter;                  // non-empty: only launches of this kernel symbol are bracketed by events
    std::vector<std::unique_ptr<ProfRec>> prof;
    std::string err;
};

namespace {

int fail(asr_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define ASR_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, ASR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

void build_geometry(Tower &t, int nf, int H, int W) {
    const int ch[9][2] = {{1, nf}, {nf, nf}, {nf, 2 * nf}, {2 * nf, 2 * nf}, {2 * nf, 4 * nf},
                          {4 * nf, 4 * nf}, {4 * nf, 4 * nf}, {4 * nf, 4 * nf}, {4 * nf, 32}};
    t.in_h = H; t.in_w = W;
    int h = H, w = W;
    for (int b = 0; b < 9; ++b) {
        LayerGeom &g = t.g[b];
        g.cin = ch[b][0]; g.cout = ch[b][1];
        g.k = b < 8 ? 3 : 1;
        g.pool = (b == 1 || b == 3 || b == 5 || b == 7) ? 1 : 0;
        g.H = h; g.W = w;
        g.OH = g.pool ? h / 2 : h;
        g.OW = g.pool ? w / 2 : w;
        h = g.OH; w = g.OW;
    }
}

ProfRec *prof_rec(asr_ctx *ctx, const std::string &name, double flops, double bytes) {
    for (auto &r : ctx->prof)
        if (r->name == name) { r->flops = flops; r->bytes = bytes; return r.get(); }
    ctx->prof.emplace_back(new ProfRec());
    ProfRec *r = ctx->prof.back().get();
    r->name = name; r->flops = flops; r->bytes = bytes;
    return r;
}

void prof_fold(ProfRec *r) {
    for (auto &p : r->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            r->total_ms += ms;
            r->launches += 1;
        }
        hipEventDestroy(p.first);
        hipEventDestroy(p.second);
    }
    r->pending.clear();
}

// RAII bracket around one kernel launch
struct ProfScope {
    asr_ctx *ctx; ProfRec *rec = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t st = nullptr;
    ProfScope(asr_ctx *c, const char *name, int view, double flops, double bytes, const char *symbol = "",
              hipStream_t on = nullptr)
        : ctx(c) {
        if (!c->profiling) return;
        if (!c->prof_filter.empty() && c->prof_filter != symbol) return;
        // events go on the stream the kernel runs on
        st = on ? on : !view ? c->stream : c->in_train ? c->tstream[view - 1] : c->estream[view - 1];
        rec = prof_rec(c, std::string(name) + (view ? (view == 1 ? "_v1" : "_v2") : ""), flops, bytes);
        rec->symbol = symbol;
        // ASR_LAUNCH_LOG=<file>: label, algorithmic FLOP / bytes and kernel symbol of every profiled launch, in launch
        // order - what tools/summarize_pmc.py joins rocprofv3's per-dispatch counters with
        static const char *log_path = getenv("ASR_LAUNCH_LOG");
        if (log_path)
            if (FILE *fp = fopen(log_path, "a")) {
                fprintf(fp, "%s\t%.0f\t%.0f\t%s\n", rec->name.c_str(), flops, bytes, symbol);
                fclose(fp);
            }
        if (rec->pending.size() >= 2048) prof_fold(rec);
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
    }
    ~ProfScope() {
        if (!rec) return;
        hipEventRecord(e1, st);
        rec->pending.emplace_back(e0, e1);
    }
};

// (re)derive the per-block launch plans of one tower from its geometry
int plan_tower(asr_ctx *ctx, Tower &tw, int view) {
    if (tw.g[8].H < 1 || tw.g[8].W < 1)
        return fail(ctx, ASR_ERR_INVALID, "view %d input %dx%d too small for four 2x2 pools", view, tw.in_h, tw.in_w);
    // ASR_FUSE1=1/3 starts from a fused plan; otherwise the autotuner decides (autotune_tower): the fused block 2
    // pays block 1's VALU work inside its staging phase (VALU does not hide under the fp32 MFMAs) but the largest
    // activation of the network (3 MB/pair written and read back) never reaches HBM.
    tw.fuse1 = getenv("ASR_FUSE1") != nullptr && (getenv("ASR_FUSE1")[0] == '1' || getenv("ASR_FUSE1")[0] == '3') &&
               getenv("ASR_NO_FUSE1") == nullptr &&
               asr::plan_conv(tw.g[1].cin, tw.g[1].cout, tw.g[1].pool, tw.g[1].H, tw.g[1].W, &tw.plan[1], 0, 1);
    for (int b = tw.fuse1 ? 2 : 1; b < 8; ++b) {
        const LayerGeom &g = tw.g[b];
        if (!asr::plan_conv_v2(g.cin, g.cout, g.pool, g.H, g.W, &tw.plan[b]) &&
            !asr::plan_conv(g.cin, g.cout, g.pool, g.H, g.W, &tw.plan[b]))
            return fail(ctx, ASR_ERR_INVALID, "no conv kernel for %d->%d pool=%d at %dx%d", g.cin, g.cout, g.pool,
                        g.H, g.W);
    }
    for (int b = 0; b < 8; ++b) tw.act_floats[b] = (size_t)tw.g[b].OH * tw.g[b].OW * tw.g[b].cout;
    return ASR_OK;
}

int check_cfg(const asr_config *cfg) {
    if (!cfg) return fail(nullptr, ASR_ERR_INVALID, "asr_create: cfg is NULL");
    if (cfg->struct_size != (int32_t)sizeof(asr_config))
        return fail(nullptr, ASR_ERR_INVALID, "asr_create: struct_size %d != %d (ABI mismatch)", cfg->struct_size,
                    (int)sizeof(asr_config));
    if (cfg->num_filters != 12 && cfg->num_filters != 24)
        return fail(nullptr, ASR_ERR_INVALID, "asr_create: num_filters must be 12 or 24, got %d", cfg->num_filters);
    if (cfg->dim_latent != 32) return fail(nullptr, ASR_ERR_INVALID, "asr_create: dim_latent must be 32");
    if (cfg->h1 < 16 || cfg->w1 < 16 || cfg->h2 < 16 || cfg->w2 < 16 || cfg->h1 > 4096 || cfg->w1 > 4096 ||
        cfg->h2 > 4096 || cfg->w2 > 4096)
        return fail(nullptr, ASR_ERR_INVALID, "asr_create: input sizes out of range");
    return ASR_OK;
}

void free_train(asr_ctx *ctx) {
    if (!ctx->train) return;
    TrainState &T = *ctx->train;
    for (auto &t : T.tw) {
        for (int b = 0; b < 9; ++b) {
            if (t.x[b]) hipFree(t.x[b]);
            if (t.z[b]) hipFree(t.z[b]);
            if (t.stats[b]) hipFree(t.stats[b]);
            if (t.zsel[b]) hipFree(t.zsel[b]);
            if (t.wdgrad[b]) hipFree(t.wdgrad[b]);
        }
        float *fp[] = {t.dz, t.dz2, t.dA, t.dB, t.H, t.dH, t.lv, t.wpartial};
        for (float *q : fp) if (q) hipFree(q);
        if (t.wstream) { (void)hipStreamSynchronize(t.wstream); (void)hipStreamDestroy(t.wstream); }
        for (int k = 0; k < 2; ++k) {
            if (t.e_dz[k]) hipEventDestroy(t.e_dz[k]);
            if (t.e_wg[k]) hipEventDestroy(t.e_wg[k]);
        }
        if (t.partial) hipFree(t.partial);
        if (t.sums && &t == &T.tw[0]) hipFree(t.sums);         // (tower 2's is the second half of tower 1's)
    }
    float *fp[] = {T.pmaster, T.pgrad, T.adam_m, T.adam_v, T.loss_dev, T.lvv[0], T.lvv[1],
                   T.Hg[0], T.Hg[1], T.dHg[0], T.dHg[1], T.lvg[0], T.lvg[1], T.Hpad[0], T.Hpad[1]};
    for (float *q : fp) if (q) hipFree(q);
    if (T.mask) hipFree(T.mask);
    if (T.repack_dev) hipFree(T.repack_dev);
    if (T.cca_ws) hipFree(T.cca_ws);
    if (T.l2_dev) hipFree(T.l2_dev);
    if (T.cca_done) hipEventDestroy(T.cca_done);
    ctx->train.reset();
}

void free_comm(asr_ctx *ctx) {
    if (!ctx->comm) return;
    Comm &c = *ctx->comm;
    if (c.nccl && c.pCommDestroy) (void)c.pCommDestroy(c.nccl);
    if (c.dl) dlclose(c.dl);
    ctx->comm.reset();
    ctx->exch = asr::Exchange{};
}

void free_pipe(asr_ctx *ctx) {
    auto &P = ctx->pipe;
    for (int s = 0; s < 2; ++s) {
        void *bufs[] = {P.in1[s], P.in2[s], P.lv1[s], P.lv2[s], P.ranks[s], P.ties[s], P.dstar[s]};
        for (void *b : bufs) if (b) (void)hipFree(b);
        hipEvent_t evs[] = {P.ready[s], P.done[s], P.out[s]};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
    }
    if (P.h2d) (void)hipStreamDestroy(P.h2d);
    if (P.d2h) (void)hipStreamDestroy(P.d2h);
    P = asr_ctx::Pipe{};
}

void free_hpipe(asr_ctx *ctx) {
    auto &H = ctx->hpipe;
    for (int s = 0; s < asr_ctx::HostPipe::NSLOT; ++s) {
        if (H.pin[s]) (void)hipHostFree(H.pin[s]);
        if (H.dev[s]) (void)hipFree(H.dev[s]);
        if (H.copied[s]) (void)hipEventDestroy(H.copied[s]);
        if (H.consumed[s]) (void)hipEventDestroy(H.consumed[s]);
        H.pin[s] = H.dev[s] = nullptr; H.copied[s] = H.consumed[s] = nullptr; H.used[s] = false;
    }
    H.slot_bytes = 0;
    if (H.out_dev) (void)hipFree(H.out_dev);
    H.out_dev = nullptr; H.out_floats = 0;
    if (H.h2d) (void)hipStreamDestroy(H.h2d);
    H.h2d = nullptr;
    H.pool.reset();
}

void free_ctx_buffers(asr_ctx *ctx) {
    free_train(ctx);
    free_comm(ctx);
    free_pipe(ctx);
    free_hpipe(ctx);
    for (auto &t : ctx->tw) {
        for (int b = 0; b < 9; ++b) { if (t.w_dev[b]) hipFree(t.w_dev[b]); if (t.bn_dev[b]) hipFree(t.bn_dev[b]); }
        for (int b = 0; b < 8; ++b) if (t.act[b]) hipFree(t.act[b]);
    }
    if (ctx->cca_dev) hipFree(ctx->cca_dev);
    for (int v = 0; v < 2; ++v) {
        if (ctx->in_stage[v]) hipFree(ctx->in_stage[v]);
        if (ctx->out_stage[v]) hipFree(ctx->out_stage[v]);
        if (ctx->vdone[v]) hipEventDestroy(ctx->vdone[v]);
        if (ctx->vstream[v]) hipStreamDestroy(ctx->vstream[v]);
    }
    if (ctx->main_done) hipEventDestroy(ctx->main_done);
    if (ctx->norm1) hipFree(ctx->norm1);
    if (ctx->norm2) hipFree(ctx->norm2);
    if (ctx->cca_ws) hipFree(ctx->cca_ws);
    if (ctx->topk_ws) hipFree(ctx->topk_ws);
    if (ctx->unit_ws) hipFree(ctx->unit_ws);
    if (ctx->rank_io) hipFree(ctx->rank_io);
    for (auto &r : ctx->prof) prof_fold(r.get());
    if (ctx->stream) hipStreamDestroy(ctx->stream);
}

// stats / stats_rows: train-mode forward only - a RAW Winograd plan also writes the BatchNorm partial sums of its outputs
// (*stats_rows > 0 on return); every other plan leaves *stats_rows at 0 and the caller runs the separate pass
hipError_t launch_conv_any(asr_ctx *ctx, hipStream_t st, const asr::ConvPlan &p, const float *in, const float *w,
                           const float *bn, float *out, int n, const asr::Fuse1Args *f1 = nullptr,
                           double *stats = nullptr, int *stats_rows = nullptr) {
    if (stats_rows) *stats_rows = 0;
    if (p.variant >= 4000)
        return asr::launch_conv_wino4(st, p, in, w + asr::conv_wpack_floats(p.cin, p.cout) + asr::wino_wpack_floats(p.cin, p.cout),
                                      bn, out, n, ctx->num_cus, stats, stats_rows);
    if (p.variant >= 3000)
        return asr::launch_conv_wino(st, p, in, w + asr::conv_wpack_floats(p.cin, p.cout), bn, out, n, ctx->num_cus,
                                     stats, stats_rows, p.fuse1 ? f1 : nullptr);
    if (p.variant >= 2000) return asr::launch_conv_v3(st, p, in, w, bn, out, n, ctx->num_cus, p.fuse1 ? f1 : nullptr);
    return p.variant >= 1000 ? asr::launch_conv_v2(st, p, in, w, bn, out, n, ctx->num_cus)
                             : asr::launch_conv(st, p, in, w, bn, out, n, ctx->num_cus, f1);
}

// "Measure, don't guess": for every MFMA conv block, time the cheapest few tilings of both schedules (by the
// planner's model) on the real buffers at the context's chunk size and keep the fastest.  ~0.5 s once per context.
// 31-bit tag of this build (asr_version() carries the source hash) for the lines of an ASR_TUNE_CACHE file
int tune_cache_tag() {
    static int tag = -1;
    if (tag < 0) {
        unsigned h = 2166136261u;
        for (const char *p = asr_version(); *p; ++p) h = (h ^ (unsigned char)*p) * 16777619u;
        tag = (int)(h & 0x7fffffffu);
    }
    return tag;
}

int autotune_tower(asr_ctx *ctx, int view) {
    Tower &t = ctx->tw[view - 1];
    hipStream_t st = ctx->estream[view - 1];
    const int n = ctx->chunk;
    hipEvent_t e0, e1;
    ASR_HIP(ctx, hipEventCreate(&e0));
    ASR_HIP(ctx, hipEventCreate(&e1));
    const bool dbg = getenv("ASR_DEBUG") != nullptr;
    for (int b = 1; b < 8; ++b) {
        const LayerGeom &g = t.g[b];
        std::vector<asr::ConvPlan> cands;
        cands.push_back(t.plan[b]);                                   // the model's choice stays a candidate
        // block 2 may absorb block 1 (ConvPlan.fuse1): ASR_FUSE1=auto - the tuner decides by time, counting block 1's
        // own kernel against the unfused candidates; "1" / "3" always fused; unset or "0" never
        const char *fenv = getenv("ASR_FUSE1");
        const bool forced = (b == 1 && t.fuse1);
        // default (unset): not tried - on the 160x200 tower fusion wins by ~2 % only, and which of the two nearly
        // equal schedules a context ends up with would vary from run to run; "auto" lets the tuner decide
        const bool try_fused = (b == 1) && fenv && fenv[0] != '0' && fenv[0] != 'w' && getenv("ASR_NO_FUSE1") == nullptr;
        // block 1 evaluated by producer waves inside the Winograd block 2 (conv3x3_wino, PW > 0): part of the default
        // candidate set wherever the build exists (C_in = 12: the `cont` model); ASR_FUSE1=0 keeps it out, =w forces it
        const bool try_wfused = (b == 1) && !(fenv && fenv[0] == '0') && getenv("ASR_NO_FUSE1") == nullptr &&
                                !(view == 1 && ctx->cfg.resize_view1) && !forced;
        const bool only_wfused = try_wfused && fenv && fenv[0] == 'w';
        if (!forced) {
            asr::conv_candidates_v1(g.cin, g.cout, g.pool, g.H, g.W, 0, 5, &cands, 0);
            asr::conv_candidates_v2(g.cin, g.cout, g.pool, g.H, g.W, 5, &cands);
            asr::conv_candidates_v3(g.cin, g.cout, g.pool, g.H, g.W, 6, &cands, 0);
            asr::conv_candidates_wino(g.cin, g.cout, g.pool, g.H, g.W,
                                      getenv("ASR_WINO_CANDS") ? atoi(getenv("ASR_WINO_CANDS")) : 4, &cands);
            asr::conv_candidates_wino4(g.cin, g.cout, g.pool, g.H, g.W, &cands);
        } else {
            asr::conv_candidates_v1(g.cin, g.cout, g.pool, g.H, g.W, 0, 5, &cands, 1);
        }
        if (try_fused) asr::conv_candidates_v3(g.cin, g.cout, g.pool, g.H, g.W, 6, &cands, 1);
        if (try_wfused) {
            asr::conv_candidates_wino_fused(g.cin, g.cout, g.pool, g.H, g.W, 3, &cands);
            if (only_wfused) {
                std::vector<asr::ConvPlan> only;
                for (auto &c : cands)
                    if (c.variant >= 3000 && c.variant < 3500 && c.fuse1) only.push_back(c);
                if (!only.empty()) cands.swap(only);
            }
        }
        if (b == 1 && fenv && fenv[0] == '3') {      // tests: only the v3 fused schedule
            std::vector<asr::ConvPlan> only;
            for (auto &c : cands)
                if (c.variant >= 2000 && c.fuse1) only.push_back(c);
            if (!only.empty()) cands.swap(only);
        }
        // ASR_TUNE_ONLY=direct|wino|winog|wino4 (tests, experiments): keep one family of schedules where the block has it
        if (const char *only_env = getenv("ASR_TUNE_ONLY")) {
            const int lo = !strcmp(only_env, "wino4") ? 4000 : !strcmp(only_env, "winog") ? 3500 : !strcmp(only_env, "wino") ? 3000 : 0;
            const int hi = !strcmp(only_env, "wino4") ? 5000 : !strcmp(only_env, "winog") ? 4000 : !strcmp(only_env, "wino") ? 3500 : 3000;
            std::vector<asr::ConvPlan> only;
            for (auto &c : cands)
                if (c.variant >= lo && c.variant < hi) only.push_back(c);
            if (!only.empty()) cands.swap(only);
        }
        // a fused block 2 reads the raw input: time it on the (0.5-filled) block-1 buffer taken as a prepared image
        asr::Fuse1Args f1{t.act[0], t.w_dev[0], t.bn_dev[0], ASR_IN_F32_PREPARED, 0, g.H, g.W};
        const asr::Fuse1Args *pf1 = &f1;
        double conv1_ms = 0.0;
        bool any_fused = false;
        for (auto &c : cands) any_fused = any_fused || c.fuse1;
        if (b == 1 && any_fused && !forced) {
            // what the unfused candidates pay on top: block 1's own kernel (input: block 2's output buffer as an image)
            const LayerGeom &g0 = t.g[0];
            ASR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)t.act[1], 0x3f000000, (size_t)g0.H * g0.W * n, st));
            (void)asr::launch_conv1(st, t.act[1], ASR_IN_F32_PREPARED, 0, t.w_dev[0], t.bn_dev[0], t.act[0], n, g0.H, g0.W,
                                    g0.H, g0.W, g0.cout);
            ASR_HIP(ctx, hipEventRecord(e0, st));
            for (int r = 0; r < 2; ++r)
                (void)asr::launch_conv1(st, t.act[1], ASR_IN_F32_PREPARED, 0, t.w_dev[0], t.bn_dev[0], t.act[0], n, g0.H,
                                        g0.W, g0.H, g0.W, g0.cout);
            ASR_HIP(ctx, hipEventRecord(e1, st));
            ASR_HIP(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            ASR_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            conv1_ms = ms / 2;
            if (dbg) fprintf(stderr, "[asr] tune v%d conv1 alone: %.4f ms\n", view, conv1_ms);
        }
        // defined input values (0.5f): timing must not depend on stale NaN / denormal bit patterns.
        // ASR_TUNE_VERIFY=1: a deterministic pattern instead, and every (unfused) candidate's output is compared
        // with the first one's - all schedules evaluate the same fp32 FMA chains in the same order.
        const bool verify = getenv("ASR_TUNE_VERIFY") != nullptr;
        const size_t out_floats = t.act_floats[b] * (size_t)n;
        float *vref = nullptr;
        uint32_t *vbits = nullptr;
        bool have_ref = false;
        if (verify) {
            ASR_HIP(ctx, asr::launch_fill_pattern(st, t.act[b - 1], (int64_t)t.act_floats[b - 1] * n));
            ASR_HIP(ctx, hipMalloc((void **)&vref, out_floats * sizeof(float) + 16));
            vbits = reinterpret_cast<uint32_t *>(vref + out_floats);
        } else
        ASR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)t.act[b - 1], 0x3f000000, t.act_floats[b - 1] * n, st));
        double best_ms = 1e30;
        int best = -1;
        // ASR_TUNE_CACHE=<file>: choices of an earlier run on the same geometry are re-used (and new ones appended),
        // so that a server restart or a profiled run does not repeat the timing launches
        const char *cache = getenv("ASR_TUNE_CACHE");
        const int nf = ctx->cfg.num_filters;
        if (cache) {
            // lines carry the build's tag (variant numbers are table indices: a cache written by another build of the
            // library would name other kernels); lines of other builds or formats are skipped
            if (FILE *fp = fopen(cache, "r")) {
                char line[256];
                while (fgets(line, sizeof line, fp)) {
                    int k[10], tag = 0;
                    if (sscanf(line, "v2 %d %d %d %d %d %d %d %d %d %d %d", &tag, &k[0], &k[1], &k[2], &k[3], &k[4], &k[5],
                               &k[6], &k[7], &k[8], &k[9]) != 11 || tag != tune_cache_tag())
                        continue;
                    if (k[0] != nf || k[1] != view || k[2] != b || k[3] != g.H || k[4] != g.W || k[5] != n) continue;
                    for (size_t c = 0; c < cands.size(); ++c)
                        if (cands[c].variant == k[6] && cands[c].TH == k[7] && cands[c].TW == k[8] && cands[c].NI == k[9])
                            best = (int)c;
                }
                fclose(fp);
            }
        }
        if (best >= 0) {
            t.plan[b] = cands[best];
            if (b == 1) t.fuse1 = cands[best].fuse1 != 0;
            if (dbg) fprintf(stderr, "[asr] tuned v%d conv%d from cache%s\n", view, b + 1, t.fuse1 && b == 1 ? " (fused)" : "");
            continue;
        }
        best = 0;
        for (size_t c = 0; c < cands.size(); ++c) {
            hipError_t e = launch_conv_any(ctx, st, cands[c], t.act[b - 1], t.w_dev[b], t.bn_dev[b], t.act[b], n, pf1);
            if (e != hipSuccess) { (void)hipGetLastError(); continue; }          // e.g. LDS request refused
            if (verify && !cands[c].fuse1) {
                if (!have_ref) {
                    ASR_HIP(ctx, hipMemcpyAsync(vref, t.act[b], out_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
                    have_ref = true;
                } else {
                    uint32_t bits = 0;
                    ASR_HIP(ctx, asr::launch_max_abs_diff(st, t.act[b], vref, (int64_t)out_floats, vbits));
                    ASR_HIP(ctx, hipMemcpyAsync(&bits, vbits, sizeof bits, hipMemcpyDeviceToHost, st));
                    ASR_HIP(ctx, hipStreamSynchronize(st));
                    float diff;
                    memcpy(&diff, &bits, sizeof diff);
                    ctx->tune_checked += 1;
                    // the Winograd schedule sums in a different order: same fp32, 1e-4 instead of 1e-5
                    if (!(diff <= (cands[c].variant >= 3000 ? 1e-4f : 1e-5f))) {
                        ctx->tune_bad += 1;
                        fprintf(stderr, "[asr] TUNE VERIFY MISMATCH view %d conv%d variant %d tile %dx%d x%d: max |diff| %g\n",
                                view, b + 1, cands[c].variant, cands[c].TH, cands[c].TW, cands[c].NI, (double)diff);
                    }
                    if (diff > ctx->tune_max_diff || diff != diff) ctx->tune_max_diff = diff;
                }
                ASR_HIP(ctx, hipMemsetAsync(t.act[b], 0xff, out_floats * sizeof(float), st));   // next candidate starts from NaNs
                (void)launch_conv_any(ctx, st, cands[c], t.act[b - 1], t.w_dev[b], t.bn_dev[b], t.act[b], n, pf1);
            }
            ASR_HIP(ctx, hipEventRecord(e0, st));
            for (int r = 0; r < 2; ++r)
                (void)launch_conv_any(ctx, st, cands[c], t.act[b - 1], t.w_dev[b], t.bn_dev[b], t.act[b], n, pf1);
            ASR_HIP(ctx, hipEventRecord(e1, st));
            ASR_HIP(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            ASR_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (dbg)
                fprintf(stderr, "[asr] tune v%d conv%d%s %s#%d tile %dx%d x%d lds %d bpc %d: %.4f ms\n", view, b + 1,
                        cands[c].fuse1 ? "+1" : "",
                        cands[c].variant >= 4000 ? "wino4" : cands[c].variant >= 3000 ? "wino" : cands[c].variant >= 2000 ? "v3" : cands[c].variant >= 1000 ? "v2" : "v1",
                        cands[c].variant % 1000, cands[c].TH, cands[c].TW,
                        cands[c].NI, cands[c].lds_bytes, cands[c].blocks_per_cu, ms / 2);
            // the row-major tile order of the global-A Winograd kernel re-reads up to 1.6x its input from HBM: it has to
            // beat the two-row strips by more than 2 % to be chosen
            const bool rowmajor_winog = cands[c].variant >= 3500 && cands[c].variant < 4000 && cands[c].TH == 1;
            const double cost = (ms / 2) * (rowmajor_winog ? 1.02 : 1.0) + (cands[c].fuse1 ? 0.0 : conv1_ms);
            if (cost < best_ms) { best_ms = cost; best = (int)c; }
        }
        if (vref) (void)hipFree(vref);
        t.plan[b] = cands[best];
        if (b == 1) t.fuse1 = cands[best].fuse1 != 0;
        if (cache) {
            if (FILE *fp = fopen(cache, "a")) {
                fprintf(fp, "v2 %d %d %d %d %d %d %d %d %d %d %d\n", tune_cache_tag(), nf, view, b, g.H, g.W, n,
                        cands[best].variant, cands[best].TH, cands[best].TW, cands[best].NI);
                fclose(fp);
            }
        }
        if (dbg)
            fprintf(stderr, "[asr] tuned v%d conv%d -> %s#%d tile %dx%d x%d (%.4f ms for %d samples)\n", view, b + 1,
                    cands[best].variant >= 4000 ? "wino4" : cands[best].variant >= 3000 ? "wino" : cands[best].variant >= 2000 ? "v3" : cands[best].variant >= 1000 ? "v2" : "v1",
                    cands[best].variant % 1000, cands[best].TH,
                    cands[best].TW, cands[best].NI, best_ms, n);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ASR_OK;
}

// activation / staging buffers are allocated on the first embed call, so that a
// context used only for ranking or CCA fitting stays small
int ensure_workspace(asr_ctx *ctx, int view) {
    Tower &t = ctx->tw[view - 1];
    for (int b = 0; b < 8; ++b)
        if (!t.act[b]) ASR_HIP(ctx, hipMalloc((void **)&t.act[b], t.act_floats[b] * ctx->chunk * sizeof(float)));
    if (!t.tuned) {
        t.tuned = true;
        const char *e = getenv("ASR_AUTOTUNE");
        if (!(e && e[0] == '0')) return autotune_tower(ctx, view);
    }
    return ASR_OK;
}

int ensure_staging(asr_ctx *ctx, int view) {
    const int v = view - 1;
    if (!ctx->in_stage[v]) ASR_HIP(ctx, hipMalloc(&ctx->in_stage[v], ctx->in_stage_bytes));
    if (!ctx->out_stage[v])
        ASR_HIP(ctx, hipMalloc((void **)&ctx->out_stage[v], (size_t)ctx->chunk * 32 * sizeof(float)));
    return ASR_OK;
}

// the main stream consumes what the tower streams produced
int join_views(asr_ctx *ctx) {
    for (int v = 0; v < 2; ++v)
        if (ctx->vpending[v]) {
            ASR_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->vdone[v], 0));
            ctx->vpending[v] = false;
        }
    return ASR_OK;
}

int mark_main(asr_ctx *ctx) {
    ASR_HIP(ctx, hipEventRecord(ctx->main_done, ctx->stream));
    ctx->main_pending = true;
    return ASR_OK;
}

int sync_all(asr_ctx *ctx) {
    for (int v = 0; v < 2; ++v) ASR_HIP(ctx, hipStreamSynchronize(ctx->vstream[v]));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->vpending[0] = ctx->vpending[1] = false;
    ctx->main_pending = false;
    return ASR_OK;
}

int refresh_wino_weights(asr_ctx *ctx);

// one tower, one chunk already on the device
int run_tower(asr_ctx *ctx, int view, const void *x_dev, int in_mode, int n, float *features_dev, float *latent_dev) {
    Tower &t = ctx->tw[view - 1];
    const asr_config &c = ctx->cfg;
    hipStream_t st = ctx->estream[view - 1];
    const int rsz = (view == 1) ? c.resize_view1 : 0;
    const int hraw = (view == 1) ? c.h1 : c.h2, wraw = (view == 1) ? c.w1 : c.w2;
    if (!t.fuse1) {
        const LayerGeom &g = t.g[0];
        ProfScope ps(ctx, "conv1", view, 2.0 * n * g.H * g.W * 9.0 * g.cout,
                     (double)n * g.H * g.W * (4.0 + 4.0 * g.cout), asr::conv1_symbol(g.cout, in_mode, rsz, n, g.H, g.W, hraw, wraw));
        ASR_HIP(ctx, asr::launch_conv1(st, x_dev, in_mode, rsz, t.w_dev[0], t.bn_dev[0], t.act[0], n, hraw,
                                       wraw, g.H, g.W, g.cout));
    }
    for (int b = 1; b < 8; ++b) {
        const LayerGeom &g = t.g[b];
        const bool fused = (b == 1 && t.fuse1);
        char name[32];
        snprintf(name, sizeof name, fused ? "conv1+%d" : "conv%d", b + 1);
        ProfScope ps(ctx, name, view,
                     2.0 * n * g.H * g.W * 9.0 * g.cin * g.cout + (fused ? 2.0 * n * g.H * g.W * 9.0 * g.cin : 0.0),
                     4.0 * n * ((double)g.H * g.W * (fused ? 1 : g.cin) + (double)g.OH * g.OW * g.cout),
                     fused ? asr::conv_wino_symbol(t.plan[b], in_mode) : t.plan[b].symbol);
        asr::Fuse1Args f1{x_dev, t.w_dev[0], t.bn_dev[0], in_mode, rsz, hraw, wraw};
        ASR_HIP(ctx, launch_conv_any(ctx, st, t.plan[b], t.act[b - 1], t.w_dev[b], t.bn_dev[b], t.act[b], n,
                                     fused ? &f1 : nullptr));
    }
    {
        const LayerGeom &g = t.g[8];
        ProfScope ps(ctx, "tail", view, 2.0 * n * (g.H * g.W * (double)g.cin * 32 + 32.0 * 32),
                     4.0 * n * ((double)g.H * g.W * g.cin + 64));
        const float *mean = ctx->cca_dev + 2048 + (view == 1 ? 0 : 32);
        const float *proj = ctx->cca_dev + (view == 1 ? 0 : 1024);
        ASR_HIP(ctx, asr::launch_tail(st, t.act[7], n, g.H, g.W, g.cin, t.w_dev[8], t.bn_dev[8], mean, proj,
                                      features_dev, latent_dev));
    }
    ctx->last_n[view - 1] = n;
    return ASR_OK;
}

size_t input_bytes_per_sample(const asr_ctx *ctx, int view, int in_mode) {
    const asr_config &c = ctx->cfg;
    if (view == 2) return (size_t)c.h2 * c.w2 * 4;
    if (in_mode == ASR_IN_F32_PREPARED) return (size_t)ctx->tw[0].in_h * ctx->tw[0].in_w * 4;
    return (size_t)c.h1 * c.w1 * (in_mode == ASR_IN_U8_RAW ? 1 : 4);
}

int embed_common(asr_ctx *ctx, int view, const void *x, int in_mode, int64_t n, int out_kind, float *out,
                 bool on_device) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->params_set) return fail(ctx, ASR_ERR_STATE, "embed: asr_set_params has not been called");
    if (n < 0 || (n > 0 && (!x || !out))) return fail(ctx, ASR_ERR_INVALID, "embed: NULL buffer or negative n");
    if (in_mode < 0 || in_mode > 2 || (view == 2 && in_mode != ASR_IN_F32_PREPARED))
        return fail(ctx, ASR_ERR_INVALID, "embed: bad in_mode %d for view %d", in_mode, view);
    if (out_kind != ASR_OUT_LATENT && out_kind != ASR_OUT_FEATURES)
        return fail(ctx, ASR_ERR_INVALID, "embed: bad out_kind %d", out_kind);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (n > 0) {
        int rcw = ensure_workspace(ctx, view);
        if (rcw != ASR_OK) return rcw;
    }
    if (ctx->wino_stale) {
        int rcr = refresh_wino_weights(ctx);
        if (rcr != ASR_OK) return rcr;
    }
    const size_t bps = input_bytes_per_sample(ctx, view, in_mode);
    hipStream_t st = ctx->estream[view - 1];
    if (n > 0 && !on_device) {
        int rcs = ensure_staging(ctx, view);
        if (rcs != ASR_OK) return rcs;
    }
    if (n > 0 && on_device && ctx->main_pending)      // the previous consumer may still read out_dev
        ASR_HIP(ctx, hipStreamWaitEvent(st, ctx->main_done, 0));
    for (int64_t s0 = 0; s0 < n; s0 += ctx->chunk) {
        const int nc = (int)std::min<int64_t>(ctx->chunk, n - s0);
        const void *xin;
        float *o;
        if (on_device) {
            xin = (const char *)x + (size_t)s0 * bps;
            o = out + (size_t)s0 * 32;
        } else {
            ASR_HIP(ctx, hipMemcpyAsync(ctx->in_stage[view - 1], (const char *)x + (size_t)s0 * bps,
                                        (size_t)nc * bps, hipMemcpyHostToDevice, st));
            xin = ctx->in_stage[view - 1];
            o = ctx->out_stage[view - 1];
        }
        int rc = run_tower(ctx, view, xin, in_mode, nc, out_kind == ASR_OUT_FEATURES ? o : nullptr,
                           out_kind == ASR_OUT_LATENT ? o : nullptr);
        if (rc != ASR_OK) return rc;
        if (!on_device) {
            ASR_HIP(ctx, hipMemcpyAsync(out + (size_t)s0 * 32, o, (size_t)nc * 32 * sizeof(float),
                                        hipMemcpyDeviceToHost, st));
            ASR_HIP(ctx, hipStreamSynchronize(st));
        }
    }
    if (n > 0 && on_device) {
        ASR_HIP(ctx, hipEventRecord(ctx->vdone[view - 1], st));
        ctx->vpending[view - 1] = true;
    }
    return ASR_OK;
}

// One host-buffer request: n samples of one view at x -> n x 32 floats at out.
struct HostJob {
    int view; const void *x; int in_mode; int64_t n; int out_kind; float *out;
};

bool host_pointer_is_pinned(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {      // plain malloc'ed memory: "invalid value"
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Host-buffer embedding of any length (what RetrievalWrapper.compute_view_1/2, run_eval.py:107-108 and
// refine_cca.py:95-97 ask for, chunk by chunk, through batch_compute1/2).  Rows are independent in deterministic
// mode, so the caller's chunking is not observable; here the array is cut into granules (125, 250, 500, 500 ...
// samples) and the H2D of granule k+1 on the copy stream overlaps the tower of granule k (with ASR_HOST_STAGE=1 also
// the staging copy of granule k+2, pageable -> page-locked; skipped when the caller's memory is page-locked already).
// Every embedding lands in one device buffer and returns in a single D2H; one host synchronisation per call.
int embed_host(asr_ctx *ctx, const HostJob *jobs, int njobs) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->params_set) return fail(ctx, ASR_ERR_STATE, "embed: asr_set_params has not been called");
    int64_t total = 0;
    size_t max_bps = 0;
    for (int j = 0; j < njobs; ++j) {
        const HostJob &J = jobs[j];
        if (J.n < 0 || (J.n > 0 && (!J.x || !J.out))) return fail(ctx, ASR_ERR_INVALID, "embed: NULL buffer or negative n");
        if (J.in_mode < 0 || J.in_mode > 2 || (J.view == 2 && J.in_mode != ASR_IN_F32_PREPARED))
            return fail(ctx, ASR_ERR_INVALID, "embed: bad in_mode %d for view %d", J.in_mode, J.view);
        if (J.out_kind != ASR_OUT_LATENT && J.out_kind != ASR_OUT_FEATURES)
            return fail(ctx, ASR_ERR_INVALID, "embed: bad out_kind %d", J.out_kind);
        total += J.n;
        if (J.n > 0) max_bps = std::max(max_bps, input_bytes_per_sample(ctx, J.view, J.in_mode));
    }
    if (total == 0) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].n > 0) {
            int rcw = ensure_workspace(ctx, jobs[j].view);
            if (rcw != ASR_OK) return rcw;
        }
    if (ctx->wino_stale) {
        int rcr = refresh_wino_weights(ctx);
        if (rcr != ASR_OK) return rcr;
    }
    auto &H = ctx->hpipe;
    constexpr int NS = asr_ctx::HostPipe::NSLOT;
    // ASR_HOST_STAGE=1: copy pageable caller memory into the page-locked slots first (CopyPool) instead of handing it
    // to hipMemcpyAsync directly.  Measured on the MI355X box (2000 pairs through RetrievalWrapper, uint8 / float32
    // sheets): direct 9.2 / 9.6 ms, staged with 4 + 1 copy threads 10.6 / 13.4 ms - the runtime's own pageable path
    // (pin in place) is faster than any host-side copy here, so direct is the default.
    // The switch is read ONCE, when the pipe is set up, and latched: the page-locked slots and the copy threads exist or
    // not for the life of the context (a value flipped to 1 later met slots without page-locked memory: a copy to NULL).
    // `granule` marks a completed set-up and is written last: a failed stream / event creation leaves the pipe unset
    // and the next call tries again instead of running on null handles.
    if (!H.granule) {
        H.staged = getenv("ASR_HOST_STAGE") && getenv("ASR_HOST_STAGE")[0] == '1';
        const char *g = getenv("ASR_HOST_GRANULE"), *g0 = getenv("ASR_HOST_GRANULE_FIRST");
        const int granule = std::max(1, std::min(ctx->chunk, g ? atoi(g) : 500));
        H.granule_first = std::max(1, std::min(granule, g0 ? atoi(g0) : 125));
        const char *t = getenv("ASR_COPY_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        const int nt = !H.staged ? 0 : t ? atoi(t) : std::max(0, std::min(4, hw / 2 - 1));
        H.pool.reset(new CopyPool(std::max(0, std::min(nt, 32))));
        if (!H.h2d) ASR_HIP(ctx, hipStreamCreateWithFlags(&H.h2d, hipStreamNonBlocking));
        for (int s = 0; s < NS; ++s) {
            if (!H.copied[s]) ASR_HIP(ctx, hipEventCreateWithFlags(&H.copied[s], hipEventDisableTiming));
            if (!H.consumed[s]) ASR_HIP(ctx, hipEventCreateWithFlags(&H.consumed[s], hipEventDisableTiming));
        }
        H.granule = granule;
    }
    const bool staged = H.staged;
    const int G = H.granule;
    // a job's largest granule: G samples, or what 16 MiB hold when samples are small (spectrograms: 1000)
    auto granule_of = [&](size_t bps) { return std::min(ctx->chunk, std::max(G, (int)((16u << 20) / bps))); };
    size_t need = 0;
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].n > 0) {
            const size_t bps = input_bytes_per_sample(ctx, jobs[j].view, jobs[j].in_mode);
            need = std::max(need, (size_t)std::min<int64_t>(granule_of(bps), jobs[j].n) * bps);
        }
    if (H.slot_bytes < need) {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
        ASR_HIP(ctx, hipStreamSynchronize(H.h2d));
        const size_t sz = std::max(need, (size_t)G * std::min<size_t>(max_bps, 1 << 16));      // (room for the usual sizes at once)
        for (int s = 0; s < NS; ++s) {
            if (H.pin[s]) { ASR_HIP(ctx, hipHostFree(H.pin[s])); H.pin[s] = nullptr; }
            if (H.dev[s]) { ASR_HIP(ctx, hipFree(H.dev[s])); H.dev[s] = nullptr; }
            H.used[s] = false;
        }
        H.slot_bytes = 0;
        for (int s = 0; s < NS; ++s) {
            if (staged) ASR_HIP(ctx, hipHostMalloc(&H.pin[s], sz, hipHostMallocDefault));      // (page-locked slots only when used)
            ASR_HIP(ctx, hipMalloc(&H.dev[s], sz));
        }
        H.slot_bytes = sz;
    }
    if (H.out_floats < (size_t)total * 32) {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
        if (H.out_dev) { ASR_HIP(ctx, hipFree(H.out_dev)); H.out_dev = nullptr; H.out_floats = 0; }
        const size_t fl = std::max((size_t)total * 32, (size_t)ctx->chunk * 32);
        ASR_HIP(ctx, hipMalloc((void **)&H.out_dev, fl * sizeof(float)));
        H.out_floats = fl;
    }
    // results of an earlier "_dev" call may still be read by the main stream
    for (int v = 0; v < 2; ++v)
        if (ctx->main_pending) ASR_HIP(ctx, hipStreamWaitEvent(ctx->estream[v], ctx->main_done, 0));
    int64_t out_row = 0;
    int slot = 0;
    for (int j = 0; j < njobs; ++j) {
        const HostJob &J = jobs[j];
        if (J.n == 0) continue;
        const size_t bps = input_bytes_per_sample(ctx, J.view, J.in_mode);
        const bool direct = !staged || host_pointer_is_pinned(J.x);
        hipStream_t st = ctx->estream[J.view - 1];
        // the first granule's copy is exposed (nothing to overlap it with): start small, double up to G
        // (at least ~4 MiB: a 125-sample granule of spectrograms is 1.9 MB and nine tiny kernels)
        const int Gj = granule_of(bps);
        int g_now = std::min(Gj, std::max(H.granule_first, (int)((4u << 20) / bps)));
        for (int64_t s0 = 0; s0 < J.n;) {
            const int nc = (int)std::min<int64_t>(g_now, J.n - s0);
            g_now = std::min(Gj, g_now * 2);
            const int s = slot;
            slot = (slot + 1) % NS;
            const char *src = (const char *)J.x + (size_t)s0 * bps;
            const size_t bytes = (size_t)nc * bps;
            if (H.used[s]) ASR_HIP(ctx, hipStreamWaitEvent(H.h2d, H.consumed[s], 0));     // dev[s] has been read
            if (!direct) {
                if (H.used[s]) ASR_HIP(ctx, hipEventSynchronize(H.copied[s]));            // pin[s] is free
                H.pool->run(H.pin[s], src, bytes);
                src = (const char *)H.pin[s];
            }
            ASR_HIP(ctx, hipMemcpyAsync(H.dev[s], src, bytes, hipMemcpyHostToDevice, H.h2d));
            ASR_HIP(ctx, hipEventRecord(H.copied[s], H.h2d));
            H.used[s] = true;
            ASR_HIP(ctx, hipStreamWaitEvent(st, H.copied[s], 0));
            float *o = H.out_dev + (size_t)(out_row + s0) * 32;
            int rc = run_tower(ctx, J.view, H.dev[s], J.in_mode, nc, J.out_kind == ASR_OUT_FEATURES ? o : nullptr,
                               J.out_kind == ASR_OUT_LATENT ? o : nullptr);
            if (rc != ASR_OK) { (void)sync_all(ctx); (void)hipStreamSynchronize(H.h2d); return rc; }
            ASR_HIP(ctx, hipEventRecord(H.consumed[s], st));
            s0 += nc;
        }
        out_row += J.n;
    }
    // copy-outs last: into pageable memory they block the host until the job's tower is through
    out_row = 0;
    for (int j = 0; j < njobs; ++j) {
        const HostJob &J = jobs[j];
        if (J.n == 0) continue;
        ASR_HIP(ctx, hipMemcpyAsync(J.out, H.out_dev + (size_t)out_row * 32, (size_t)J.n * 32 * sizeof(float),
                                    hipMemcpyDeviceToHost, ctx->estream[J.view - 1]));
        out_row += J.n;
    }
    ASR_HIP(ctx, hipStreamSynchronize(H.h2d));
    return sync_all(ctx);
}

int ensure_norms(asr_ctx *ctx, int64_t n1, int64_t n2) {
    if (n1 > ctx->norm_cap1) {
        if (ctx->norm1) hipFree(ctx->norm1);
        ctx->norm1 = nullptr; ctx->norm_cap1 = 0;
        ASR_HIP(ctx, hipMalloc((void **)&ctx->norm1, (size_t)n1 * sizeof(double)));
        ctx->norm_cap1 = n1;
    }
    if (n2 > ctx->norm_cap2) {
        if (ctx->norm2) hipFree(ctx->norm2);
        ctx->norm2 = nullptr; ctx->norm_cap2 = 0;
        ASR_HIP(ctx, hipMalloc((void **)&ctx->norm2, (size_t)n2 * sizeof(double)));
        ctx->norm_cap2 = n2;
    }
    return ASR_OK;
}

int rank_check(asr_ctx *ctx, int64_t n1, int64_t ld1, int64_t n2, int64_t ld2, int dim, int64_t query_offset,
               int64_t n1_global) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n1 < 0 || n2 < 0 || dim < 1 || dim > 64 || ld1 < dim || ld2 < dim || query_offset < 0 ||
        n1_global < query_offset + n1)
        return fail(ctx, ASR_ERR_INVALID, "rank: bad sizes n1=%lld n2=%lld dim=%d ld=(%lld,%lld) off=%lld n1g=%lld",
                    (long long)n1, (long long)n2, dim, (long long)ld1, (long long)ld2, (long long)query_offset,
                    (long long)n1_global);
    if (n1 > 0 && n2 == 0) return fail(ctx, ASR_ERR_INVALID, "rank: empty candidate list");
    return ASR_OK;
}

}  // namespace

static int train_download_master(asr_ctx *ctx);
static int train_upload_master(asr_ctx *ctx);

extern "C" {

const char *asr_last_error(const asr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int asr_create(const asr_config *cfg, asr_ctx **out) {
    if (!out) return fail(nullptr, ASR_ERR_INVALID, "asr_create: out is NULL");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc != ASR_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, ASR_ERR_HIP, "asr_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, ASR_ERR_INVALID, "asr_create: device %d out of range (%d devices)", cfg->device, ndev);
    std::unique_ptr<asr_ctx> ctx(new asr_ctx());
    ctx->cfg = *cfg;
    asr_ctx *c = ctx.get();
#define CREATE_HIP(call)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) {                                                                           \
            free_ctx_buffers(c);                                                                           \
            return fail(nullptr, ASR_ERR_HIP, "asr_create: %s failed: %s", #call, hipGetErrorString(e__)); \
        }                                                                                                  \
    } while (0)
    CREATE_HIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, cfg->device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // Both towers on one stream by default: the persistent conv kernels fill the chip on their own, every kernel's
    // measured duration is its own (bench.py's roofline, rocprofv3), and letting the towers overlap on two streams
    // (ASR_TWO_STREAMS=1) buys under 3 % of throughput
    {
        const char *two = getenv("ASR_TWO_STREAMS");
        c->single_stream = !(two && two[0] == '1') || getenv("ASR_SINGLE_STREAM") != nullptr;
    }
    for (int v = 0; v < 2; ++v) {
        CREATE_HIP(hipStreamCreateWithFlags(&c->vstream[v], hipStreamNonBlocking));
        c->estream[v] = c->single_stream ? c->stream : c->vstream[v];
        CREATE_HIP(hipEventCreateWithFlags(&c->vdone[v], hipEventDisableTiming));
    }
    CREATE_HIP(hipEventCreateWithFlags(&c->main_done, hipEventDisableTiming));
    c->chunk = cfg->max_chunk > 0 ? cfg->max_chunk : 1000;

    const int nf = cfg->num_filters;
    const int H1 = cfg->resize_view1 ? cfg->h1 / 2 : cfg->h1, W1 = cfg->resize_view1 ? cfg->w1 / 2 : cfg->w1;
    build_geometry(c->tw[0], nf, H1, W1);
    build_geometry(c->tw[1], nf, cfg->h2, cfg->w2);

    // parameter table in the reference's order
    for (int t = 0; t < 2; ++t)
        for (int b = 0; b < 9; ++b) {
            const LayerGeom &g = c->tw[t].g[b];
            c->pshape.push_back({g.cout, g.cin, g.k, g.k});
            for (int q = 0; q < 4; ++q) c->pshape.push_back({g.cout});
        }
    c->pshape.push_back({32, 32}); c->pshape.push_back({32, 32});
    c->pshape.push_back({32}); c->pshape.push_back({32});
    c->pshape.push_back({32, 32}); c->pshape.push_back({32, 32}); c->pshape.push_back({32, 32});
    for (auto &s : c->pshape) {
        int64_t n = 1;
        for (auto d : s) n *= d;
        c->params.emplace_back((size_t)n, 0.0f);
    }

    for (int t = 0; t < 2; ++t) {
        Tower &tw = c->tw[t];
        for (int b = 0; b < 9; ++b) {
            const LayerGeom &g = tw.g[b];
            size_t wfl;
            if (b == 0) wfl = (size_t)g.cout * 9;
            else if (b < 8) wfl = asr::conv_wpack_floats(g.cin, g.cout) + asr::wino_wpack_floats(g.cin, g.cout) +
                                  asr::wino4_wpack_floats(g.cin, g.cout);
            else wfl = (size_t)32 * g.cin;
            CREATE_HIP(hipMalloc((void **)&tw.w_dev[b], wfl * sizeof(float)));
            const int coutp = (g.cout + 15) / 16 * 16;
            CREATE_HIP(hipMalloc((void **)&tw.bn_dev[b], (size_t)3 * coutp * sizeof(float)));
        }
        int rcp = plan_tower(nullptr, tw, t + 1);
        if (rcp != ASR_OK) { free_ctx_buffers(c); return rcp; }
    }
    CREATE_HIP(hipMalloc((void **)&c->cca_dev, (size_t)(2048 + 64) * sizeof(float)));
    CREATE_HIP(hipMemsetAsync(c->cca_dev, 0, (size_t)(2048 + 64) * sizeof(float), c->stream));
    c->in_stage_bytes = (size_t)c->chunk * std::max((size_t)cfg->h1 * cfg->w1, (size_t)cfg->h2 * cfg->w2) * 4;
    CREATE_HIP(hipStreamSynchronize(c->stream));
#undef CREATE_HIP
    *out = ctx.release();
    return ASR_OK;
}

void asr_destroy(asr_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->cfg.device);
    for (int v = 0; v < 2; ++v)
        if (ctx->vstream[v]) hipStreamSynchronize(ctx->vstream[v]);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    free_ctx_buffers(ctx);
    delete ctx;
}

int asr_sync(asr_ctx *ctx) {
    if (!ctx) return ASR_ERR_INVALID;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    return sync_all(ctx);
}

int asr_set_input_size(asr_ctx *ctx, int view, int h, int w) {
    if (!ctx || view < 1 || view > 2) return ASR_ERR_INVALID;
    if (h < 16 || w < 16 || h > 4096 || w > 4096)
        return fail(ctx, ASR_ERR_INVALID, "set_input_size: %dx%d out of range", h, w);
    asr_config &c = ctx->cfg;
    if (view == 1 && c.h1 == h && c.w1 == w) return ASR_OK;
    if (view == 2 && c.h2 == h && c.w2 == w) return ASR_OK;
    if (ctx->train)       // the training buffers and plans were sized for the current geometry
        return fail(ctx, ASR_ERR_STATE, "set_input_size: a training state is active (sized for %dx%d / %dx%d); call "
                                        "asr_train_end first", c.h1, c.w1, c.h2, c.w2);
    ASR_HIP(ctx, hipSetDevice(c.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    Tower saved = ctx->tw[view - 1];
    Tower &tw = ctx->tw[view - 1];
    const int nh = (view == 1 && c.resize_view1) ? h / 2 : h, nw = (view == 1 && c.resize_view1) ? w / 2 : w;
    build_geometry(tw, c.num_filters, nh, nw);
    int rc = plan_tower(ctx, tw, view);
    if (rc != ASR_OK) { ctx->tw[view - 1] = saved; return rc; }
    for (int b = 0; b < 8; ++b) {       // activation buffers are re-allocated lazily at the new size
        if (tw.act[b]) ASR_HIP(ctx, hipFree(tw.act[b]));
        tw.act[b] = nullptr;
    }
    tw.tuned = false;
    if (view == 1) { c.h1 = h; c.w1 = w; } else { c.h2 = h; c.w2 = w; }
    for (int v = 0; v < 2; ++v)
        if (ctx->in_stage[v]) { ASR_HIP(ctx, hipFree(ctx->in_stage[v])); ctx->in_stage[v] = nullptr; }
    ctx->in_stage_bytes = (size_t)ctx->chunk * std::max((size_t)c.h1 * c.w1, (size_t)c.h2 * c.w2) * 4;
    ctx->last_n[view - 1] = 0;
    return ASR_OK;
}

int asr_param_count(const asr_ctx *ctx) { return ctx ? (int)ctx->params.size() : -1; }

int asr_param_size(const asr_ctx *ctx, int index, int64_t *n_elements) {
    if (!ctx || !n_elements || index < 0 || index >= (int)ctx->params.size()) return ASR_ERR_INVALID;
    *n_elements = (int64_t)ctx->params[index].size();
    return ASR_OK;
}

static int upload_network(asr_ctx *ctx) {
    for (int t = 0; t < 2; ++t) {
        Tower &tw = ctx->tw[t];
        for (int b = 0; b < 9; ++b) {
            const LayerGeom &g = tw.g[b];
            const int base = 45 * t + 5 * b;
            const std::vector<float> &W = ctx->params[base];
            const float *beta = ctx->params[base + 1].data(), *gamma = ctx->params[base + 2].data();
            const float *mean = ctx->params[base + 3].data(), *istd = ctx->params[base + 4].data();
            std::vector<float> wdev;
            if (b == 0) {              // [co][9] correlation-form taps: W[co][0][2-a][2-b]
                wdev.resize((size_t)g.cout * 9);
                for (int co = 0; co < g.cout; ++co)
                    for (int a = 0; a < 3; ++a)
                        for (int bb = 0; bb < 3; ++bb)
                            wdev[(size_t)co * 9 + a * 3 + bb] = W[((size_t)co * 1 + 0) * 9 + (2 - a) * 3 + (2 - bb)];
            } else if (b < 8) {        // [tap][ci][co] correlation form -> MFMA fragment order
                std::vector<float> wc((size_t)9 * g.cin * g.cout);
                for (int co = 0; co < g.cout; ++co)
                    for (int ci = 0; ci < g.cin; ++ci)
                        for (int a = 0; a < 3; ++a)
                            for (int bb = 0; bb < 3; ++bb)
                                wc[((size_t)(a * 3 + bb) * g.cin + ci) * g.cout + co] =
                                    W[((size_t)co * g.cin + ci) * 9 + (2 - a) * 3 + (2 - bb)];
                wdev.resize(asr::conv_wpack_floats(g.cin, g.cout));
                asr::pack_conv_weights(wc.data(), g.cin, g.cout, wdev.data());
            } else {                   // 1x1: [o][c]
                wdev.assign(W.begin(), W.end());
            }
            ASR_HIP(ctx, hipMemcpyAsync(tw.w_dev[b], wdev.data(), wdev.size() * sizeof(float), hipMemcpyHostToDevice,
                                        ctx->stream));
            if (b >= 1 && b < 8) {     // Winograd-domain weights behind the direct-form fragments, transformed on the device
                float *raw = nullptr;
                ASR_HIP(ctx, hipMalloc((void **)&raw, W.size() * sizeof(float)));
                hipError_t e1 = hipMemcpyAsync(raw, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
                if (e1 == hipSuccess)
                    e1 = asr::launch_wino_pack(ctx->stream, raw, g.cin, g.cout, tw.w_dev[b] + wdev.size());
                if (e1 == hipSuccess)
                    e1 = asr::launch_wino4_pack(ctx->stream, raw, g.cin, g.cout,
                                                tw.w_dev[b] + wdev.size() + asr::wino_wpack_floats(g.cin, g.cout));
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
                (void)hipFree(raw);
                ASR_HIP(ctx, e1);
            }
            const int coutp = (g.cout + 15) / 16 * 16;
            std::vector<float> bn((size_t)3 * coutp, 0.0f);
            for (int co = 0; co < g.cout; ++co) {
                bn[co] = mean[co];
                bn[coutp + co] = gamma[co] * istd[co];   // fp32 product, as (gamma * inv_std) in the reference
                bn[2 * coutp + co] = beta[co];
            }
            ASR_HIP(ctx, hipMemcpyAsync(tw.bn_dev[b], bn.data(), bn.size() * sizeof(float), hipMemcpyHostToDevice,
                                        ctx->stream));
            ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));   // staging vectors die at scope end
        }
    }
    std::vector<float> cca(2048 + 64);
    memcpy(cca.data(), ctx->params[90].data(), 1024 * sizeof(float));
    memcpy(cca.data() + 1024, ctx->params[91].data(), 1024 * sizeof(float));
    memcpy(cca.data() + 2048, ctx->params[92].data(), 32 * sizeof(float));
    memcpy(cca.data() + 2080, ctx->params[93].data(), 32 * sizeof(float));
    ASR_HIP(ctx, hipMemcpyAsync(ctx->cca_dev, cca.data(), cca.size() * sizeof(float), hipMemcpyHostToDevice,
                                ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_OK;
}

int asr_set_params(asr_ctx *ctx, const float *const *arrays, const int64_t *sizes, int n_arrays) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!arrays || !sizes || n_arrays != (int)ctx->params.size())
        return fail(ctx, ASR_ERR_INVALID, "set_params: expected %d arrays, got %d", (int)ctx->params.size(), n_arrays);
    for (int i = 0; i < n_arrays; ++i)
        if (!arrays[i] || sizes[i] != (int64_t)ctx->params[i].size())
            return fail(ctx, ASR_ERR_INVALID, "set_params: array %d has %lld elements, expected %lld", i,
                        (long long)sizes[i], (long long)ctx->params[i].size());
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);      // towers in flight still read the old weights
        if (rcs != ASR_OK) return rcs;
    }
    for (int i = 0; i < n_arrays; ++i) memcpy(ctx->params[i].data(), arrays[i], (size_t)sizes[i] * sizeof(float));
    int rc = upload_network(ctx);
    if (rc == ASR_OK) ctx->params_set = true;
    if (rc == ASR_OK && ctx->train) rc = train_upload_master(ctx);
    return rc;
}

int asr_get_params(asr_ctx *ctx, float *const *arrays, const int64_t *sizes, int n_arrays) {
    if (!ctx) return ASR_ERR_INVALID;
    if (ctx->train && ctx->train->master_dirty) {
        int rcd = train_download_master(ctx);
        if (rcd != ASR_OK) return rcd;
    }
    if (!arrays || !sizes || n_arrays != (int)ctx->params.size())
        return fail(ctx, ASR_ERR_INVALID, "get_params: expected %d arrays, got %d", (int)ctx->params.size(), n_arrays);
    for (int i = 0; i < n_arrays; ++i) {
        if (!arrays[i] || sizes[i] != (int64_t)ctx->params[i].size())
            return fail(ctx, ASR_ERR_INVALID, "get_params: array %d has %lld elements, expected %lld", i,
                        (long long)sizes[i], (long long)ctx->params[i].size());
        memcpy(arrays[i], ctx->params[i].data(), (size_t)sizes[i] * sizeof(float));
    }
    return ASR_OK;
}

int asr_set_cca(asr_ctx *ctx, const float *U, const float *V, const float *mean1, const float *mean2) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!U || !V || !mean1 || !mean2) return fail(ctx, ASR_ERR_INVALID, "set_cca: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    // training moved the device master: the host mirror train_upload_master() rebuilds it from must be current
    if (ctx->train && ctx->train->master_dirty) {
        int rcd = train_download_master(ctx);
        if (rcd != ASR_OK) return rcd;
    }
    memcpy(ctx->params[90].data(), U, 1024 * sizeof(float));
    memcpy(ctx->params[91].data(), V, 1024 * sizeof(float));
    memcpy(ctx->params[92].data(), mean1, 32 * sizeof(float));
    memcpy(ctx->params[93].data(), mean2, 32 * sizeof(float));
    std::vector<float> cca(2048 + 64);
    memcpy(cca.data(), U, 1024 * sizeof(float));
    memcpy(cca.data() + 1024, V, 1024 * sizeof(float));
    memcpy(cca.data() + 2048, mean1, 32 * sizeof(float));
    memcpy(cca.data() + 2080, mean2, 32 * sizeof(float));
    ASR_HIP(ctx, hipMemcpyAsync(ctx->cca_dev, cca.data(), cca.size() * sizeof(float), hipMemcpyHostToDevice,
                                ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->train) return train_upload_master(ctx);
    return ASR_OK;
}

// ASR_HOST_PIPE=0: the round-2 form (one synchronous copy-in / tower / copy-out per chunk), kept for A/B timing
static bool host_pipe_off() {
    static const bool off = getenv("ASR_HOST_PIPE") && getenv("ASR_HOST_PIPE")[0] == '0';
    return off;
}
int asr_embed_view1(asr_ctx *ctx, const void *x, int in_mode, int64_t n, int out_kind, float *out) {
    if (host_pipe_off()) return embed_common(ctx, 1, x, in_mode, n, out_kind, out, false);
    const HostJob job{1, x, in_mode, n, out_kind, out};
    return embed_host(ctx, &job, 1);
}
int asr_embed_view2(asr_ctx *ctx, const float *z, int64_t n, int out_kind, float *out) {
    if (host_pipe_off()) return embed_common(ctx, 2, z, ASR_IN_F32_PREPARED, n, out_kind, out, false);
    const HostJob job{2, z, ASR_IN_F32_PREPARED, n, out_kind, out};
    return embed_host(ctx, &job, 1);
}
int asr_embed_both(asr_ctx *ctx, const void *x, int in_mode, const float *z, int64_t n, int out_kind, float *out1,
                   float *out2) {
    if (host_pipe_off()) {
        int rc = embed_common(ctx, 1, x, in_mode, n, out_kind, out1, false);
        if (rc != ASR_OK) return rc;
        return embed_common(ctx, 2, z, ASR_IN_F32_PREPARED, n, out_kind, out2, false);
    }
    // one pass of the pipeline over both views: no host synchronisation between the towers
    const HostJob jobs[2] = {{1, x, in_mode, n, out_kind, out1}, {2, z, ASR_IN_F32_PREPARED, n, out_kind, out2}};
    return embed_host(ctx, jobs, 2);
}
int asr_embed_view1_dev(asr_ctx *ctx, const void *x_dev, int in_mode, int64_t n, int out_kind, float *out_dev) {
    return embed_common(ctx, 1, x_dev, in_mode, n, out_kind, out_dev, true);
}
int asr_embed_view2_dev(asr_ctx *ctx, const float *z_dev, int64_t n, int out_kind, float *out_dev) {
    return embed_common(ctx, 2, z_dev, ASR_IN_F32_PREPARED, n, out_kind, out_dev, true);
}

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
        ASR_HIP(ctx, asr::launch_topk(ctx->stream, db, ctx->norm2, n_db, ld_db, q, ctx->norm1, n_q, ld_q, dim, k,
                                      idx_offset, idx, dist, ctx->topk_ws, unit, rn));
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
static int db_query_begin(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, const char *who) {
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
    ProfScope ps(ctx, "row_norms", 0, 2.0 * db->dim * (double)n_q, 4.0 * db->dim * (double)n_q);
    ASR_HIP(ctx, asr::launch_row_norms(ctx->stream, q, n_q, ld_q, db->dim, ctx->norm1));
    return ASR_OK;
}

int asr_topk_db_dev(asr_ctx *ctx, const asr_db *db, const float *q, int64_t n_q, int64_t ld_q, int k, int64_t idx_offset,
                    int32_t *idx, double *dist) {
    int rc = db_query_begin(ctx, db, q, n_q, ld_q, "topk_db");
    if (rc != ASR_OK || n_q == 0) return rc;
    if (k < 1 || k > 128 || !idx || !dist) return fail(ctx, ASR_ERR_INVALID, "topk_db: k=%d (1..128) / NULL output", k);
    ProfScope ps(ctx, "topk", 0, 2.0 * db->dim * (double)n_q * (double)db->n, 4.0 * db->dim * (double)db->n);
    rc = grow_topk_ws(ctx, asr::topk_workspace_bytes(db->n, n_q, k, db->unit != nullptr, false));
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, asr::launch_topk(ctx->stream, db->codes, db->norms, db->n, db->ld, q, ctx->norm1, n_q, ld_q, db->dim, k,
                                  idx_offset, idx, dist, ctx->topk_ws, db->unit, db->rn));
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
    if (k < 1 || k > 128 || !idx || !dist) return fail(ctx, ASR_ERR_INVALID, "topk_rank_db: k=%d (1..128) / NULL output", k);
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
        ASR_HIP(ctx, asr::launch_topk(ctx->stream, db->codes, db->norms, db->n, db->ld, q, ctx->norm1, n_q, ld_q, db->dim,
                                      k, idx_offset, idx, dist, ctx->topk_ws, db->unit, db->rn));
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

int asr_cca_fit_dev(asr_ctx *ctx, const float *H1_dev, const float *H2_dev, int64_t n, float *U_dev, float *V_dev,
                    float *means_dev, double *coeffs_dev) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n < 2) return fail(ctx, ASR_ERR_INVALID, "cca_fit: needs at least 2 samples, got %lld", (long long)n);
    if (!H1_dev || !H2_dev || !U_dev || !V_dev || !means_dev || !coeffs_dev)
        return fail(ctx, ASR_ERR_INVALID, "cca_fit: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t need = asr::cca_workspace_bytes(n);
    if (need > ctx->cca_ws_bytes) {
        if (ctx->cca_ws) ASR_HIP(ctx, hipFree(ctx->cca_ws));
        ctx->cca_ws = nullptr; ctx->cca_ws_bytes = 0;
        ASR_HIP(ctx, hipMalloc(&ctx->cca_ws, need));
        ctx->cca_ws_bytes = need;
    }
    {
        int rcj = join_views(ctx);
        if (rcj != ASR_OK) return rcj;
    }
    {
        ProfScope ps(ctx, "cca_fit", 0, 6.0 * 32 * 32 * (double)n, 2.0 * 256.0 * (double)n);
        ASR_HIP(ctx, asr::launch_cca_fit(ctx->stream, H1_dev, H2_dev, n, ctx->cfg.r1, ctx->cfg.r2, ctx->cca_ws,
                                         U_dev, V_dev, means_dev, coeffs_dev));
    }
    return mark_main(ctx);
}

int asr_cca_fit(asr_ctx *ctx, const float *H1, const float *H2, int64_t n, float *U, float *V, float *mean1,
                float *mean2, double *coeffs) {
    if (!ctx) return ASR_ERR_INVALID;
    if (n < 2) return fail(ctx, ASR_ERR_INVALID, "cca_fit: needs at least 2 samples, got %lld", (long long)n);
    if (!H1 || !H2 || !U || !V || !mean1 || !mean2) return fail(ctx, ASR_ERR_INVALID, "cca_fit: NULL argument");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    float *d = nullptr;       // H1 | H2 | U | V | means ; then coeffs (double)
    double *dc = nullptr;
    const size_t hf = (size_t)n * 32;
    auto cleanup = [&]() { (void)hipFree(d); (void)hipFree(dc); };
#define CCA_HIP(call)                                                                                  \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            cleanup();                                                                                 \
            return fail(ctx, ASR_ERR_HIP, "asr_cca_fit: %s failed: %s", #call, hipGetErrorString(e__)); \
        }                                                                                              \
    } while (0)
    CCA_HIP(hipMalloc((void **)&d, (2 * hf + 2048 + 64) * sizeof(float)));
    CCA_HIP(hipMalloc((void **)&dc, 32 * sizeof(double)));
    CCA_HIP(hipMemcpyAsync(d, H1, hf * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    CCA_HIP(hipMemcpyAsync(d + hf, H2, hf * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = asr_cca_fit_dev(ctx, d, d + hf, n, d + 2 * hf, d + 2 * hf + 1024, d + 2 * hf + 2048, dc);
    if (rc != ASR_OK) { cleanup(); return rc; }
    CCA_HIP(hipMemcpyAsync(U, d + 2 * hf, 1024 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CCA_HIP(hipMemcpyAsync(V, d + 2 * hf + 1024, 1024 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CCA_HIP(hipMemcpyAsync(mean1, d + 2 * hf + 2048, 32 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    CCA_HIP(hipMemcpyAsync(mean2, d + 2 * hf + 2080, 32 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (coeffs) CCA_HIP(hipMemcpyAsync(coeffs, dc, 32 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CCA_HIP(hipStreamSynchronize(ctx->stream));
#undef CCA_HIP
    cleanup();
    return ASR_OK;
}

/* ---- host-buffer pipeline ---------------------------------------------------------------------------------- */
int asr_host_alloc(asr_ctx *ctx, size_t bytes, void **hptr) {
    if (!ctx || !hptr) return ASR_ERR_INVALID;
    *hptr = nullptr;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    hipError_t e = hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_NOMEM, "host_alloc(%zu): %s", bytes, hipGetErrorString(e));
    return ASR_OK;
}
int asr_host_free(asr_ctx *ctx, void *hptr) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!hptr) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ASR_HIP(ctx, hipHostFree(hptr));
    return ASR_OK;
}

int asr_eval_batches(asr_ctx *ctx, const void *const *x, int in_mode, const float *const *z, int n_batches, int64_t n,
                     int32_t *const *ranks, double *const *dstar, int32_t *const *ties, float *const *lv1,
                     float *const *lv2) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->params_set) return fail(ctx, ASR_ERR_STATE, "eval_batches: asr_set_params has not been called");
    if (n_batches < 0 || n < 1 || n > (1 << 24) || in_mode < 0 || in_mode > 2)
        return fail(ctx, ASR_ERR_INVALID, "eval_batches: bad sizes (n_batches %d, n %lld, in_mode %d)", n_batches,
                    (long long)n, in_mode);
    if (n_batches == 0) return ASR_OK;
    if (!x || !z || !ranks) return fail(ctx, ASR_ERR_INVALID, "eval_batches: NULL argument");
    for (int k = 0; k < n_batches; ++k)
        if (!x[k] || !z[k] || !ranks[k]) return fail(ctx, ASR_ERR_INVALID, "eval_batches: batch %d has a NULL buffer", k);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    auto &P = ctx->pipe;
    const size_t b1 = (size_t)n * input_bytes_per_sample(ctx, 1, in_mode), b2 = (size_t)n * input_bytes_per_sample(ctx, 2, 0);
    if (P.n != n || P.b1 < b1 || P.b2 < b2) {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
        free_pipe(ctx);
        ASR_HIP(ctx, hipStreamCreateWithFlags(&P.h2d, hipStreamNonBlocking));
        ASR_HIP(ctx, hipStreamCreateWithFlags(&P.d2h, hipStreamNonBlocking));
        for (int s = 0; s < 2; ++s) {
            ASR_HIP(ctx, hipMalloc(&P.in1[s], b1));
            ASR_HIP(ctx, hipMalloc((void **)&P.in2[s], b2));
            ASR_HIP(ctx, hipMalloc((void **)&P.lv1[s], (size_t)n * 32 * sizeof(float)));
            ASR_HIP(ctx, hipMalloc((void **)&P.lv2[s], (size_t)n * 32 * sizeof(float)));
            ASR_HIP(ctx, hipMalloc((void **)&P.ranks[s], (size_t)n * sizeof(int32_t)));
            ASR_HIP(ctx, hipMalloc((void **)&P.ties[s], (size_t)n * sizeof(int32_t)));
            ASR_HIP(ctx, hipMalloc((void **)&P.dstar[s], (size_t)n * sizeof(double)));
            ASR_HIP(ctx, hipEventCreateWithFlags(&P.ready[s], hipEventDisableTiming));
            ASR_HIP(ctx, hipEventCreateWithFlags(&P.done[s], hipEventDisableTiming));
            ASR_HIP(ctx, hipEventCreateWithFlags(&P.out[s], hipEventDisableTiming));
        }
        P.n = n; P.b1 = b1; P.b2 = b2;
    }
    // the pipeline starts from an idle context: whatever the compute streams still hold is older than batch 0
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    auto upload = [&](int k) -> int {
        const int s = k & 1;
        if (k >= 2) ASR_HIP(ctx, hipStreamWaitEvent(P.h2d, P.done[s], 0));          // batch k-2 has read in[s]
        ASR_HIP(ctx, hipMemcpyAsync(P.in1[s], x[k], b1, hipMemcpyHostToDevice, P.h2d));
        ASR_HIP(ctx, hipMemcpyAsync(P.in2[s], z[k], b2, hipMemcpyHostToDevice, P.h2d));
        ASR_HIP(ctx, hipEventRecord(P.ready[s], P.h2d));
        return ASR_OK;
    };
    // every exit after the first enqueued copy drains the copy streams: asynchronous copies into or out of the caller's
    // host buffers must not be in flight when the caller gets control back (and frees them)
    auto run = [&]() -> int {
        int rc = upload(0);
        if (rc != ASR_OK) return rc;
        for (int k = 0; k < n_batches; ++k) {
            const int s = k & 1;
            if (k + 1 < n_batches && (rc = upload(k + 1)) != ASR_OK) return rc;      // overlaps the compute of batch k
            for (int v = 0; v < 2; ++v) {
                ASR_HIP(ctx, hipStreamWaitEvent(ctx->estream[v], P.ready[s], 0));
                if (k >= 2) ASR_HIP(ctx, hipStreamWaitEvent(ctx->estream[v], P.out[s], 0));   // outputs of k-2 are on the host
            }
            if ((rc = embed_common(ctx, 1, P.in1[s], in_mode, n, ASR_OUT_LATENT, P.lv1[s], true)) != ASR_OK) return rc;
            if ((rc = embed_common(ctx, 2, P.in2[s], ASR_IN_F32_PREPARED, n, ASR_OUT_LATENT, P.lv2[s], true)) != ASR_OK) return rc;
            if (k >= 2) ASR_HIP(ctx, hipStreamWaitEvent(ctx->stream, P.out[s], 0));
            if ((rc = asr_rank_dev(ctx, P.lv1[s], n, 32, P.lv2[s], n, 32, 32, 0, n, P.ranks[s], P.dstar[s], P.ties[s])) != ASR_OK)
                return rc;
            ASR_HIP(ctx, hipEventRecord(P.done[s], ctx->stream));
            ASR_HIP(ctx, hipStreamWaitEvent(P.d2h, P.done[s], 0));
            ASR_HIP(ctx, hipMemcpyAsync(ranks[k], P.ranks[s], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, P.d2h));
            if (dstar && dstar[k])
                ASR_HIP(ctx, hipMemcpyAsync(dstar[k], P.dstar[s], (size_t)n * sizeof(double), hipMemcpyDeviceToHost, P.d2h));
            if (ties && ties[k])
                ASR_HIP(ctx, hipMemcpyAsync(ties[k], P.ties[s], (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, P.d2h));
            if (lv1 && lv1[k])
                ASR_HIP(ctx, hipMemcpyAsync(lv1[k], P.lv1[s], (size_t)n * 32 * sizeof(float), hipMemcpyDeviceToHost, P.d2h));
            if (lv2 && lv2[k])
                ASR_HIP(ctx, hipMemcpyAsync(lv2[k], P.lv2[s], (size_t)n * 32 * sizeof(float), hipMemcpyDeviceToHost, P.d2h));
            ASR_HIP(ctx, hipEventRecord(P.out[s], P.d2h));
        }
        return ASR_OK;
    };
    const int rc = run();
    const std::string first_error = rc != ASR_OK ? ctx->err : std::string();
    const hipError_t e1 = hipStreamSynchronize(P.h2d), e2 = hipStreamSynchronize(P.d2h);
    const int rcs = sync_all(ctx);
    if (rc != ASR_OK) { ctx->err = first_error; return rc; }
    if (e1 != hipSuccess || e2 != hipSuccess)
        return fail(ctx, ASR_ERR_HIP, "eval_batches: copy stream failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return rcs;
}

int asr_dev_alloc(asr_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return ASR_ERR_INVALID;
    *dptr = nullptr;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_NOMEM, "dev_alloc(%zu): %s", bytes, hipGetErrorString(e));
    return ASR_OK;
}
int asr_dev_free(asr_ctx *ctx, void *dptr) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!dptr) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    ASR_HIP(ctx, hipFree(dptr));
    return ASR_OK;
}
int asr_dev_upload(asr_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!ctx || (bytes && (!dst_dev || !src_host))) return ASR_ERR_INVALID;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    ASR_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_OK;
}
int asr_dev_download(asr_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx || (bytes && (!dst_host || !src_dev))) return ASR_ERR_INVALID;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    ASR_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_OK;
}

int asr_profile_enable(asr_ctx *ctx, int on) {
    if (!ctx) return ASR_ERR_INVALID;
    ctx->profiling = on != 0;
    return ASR_OK;
}
int asr_profile_filter(asr_ctx *ctx, const char *symbol) {
    if (!ctx) return ASR_ERR_INVALID;
    ctx->prof_filter = symbol ? symbol : "";
    return ASR_OK;
}
int asr_profile_reset(asr_ctx *ctx) {
    if (!ctx) return ASR_ERR_INVALID;
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    for (auto &r : ctx->prof) prof_fold(r.get());
    ctx->prof.clear();
    return ASR_OK;
}
int asr_profile_count(asr_ctx *ctx) { return ctx ? (int)ctx->prof.size() : -1; }
int asr_profile_get(asr_ctx *ctx, int index, char *name, int name_cap, int64_t *launches, double *total_ms,
                    double *flops, double *bytes) {
    if (!ctx || index < 0 || index >= (int)ctx->prof.size()) return ASR_ERR_INVALID;
    ProfRec *r = ctx->prof[index].get();
    prof_fold(r);
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", r->name.c_str());
    if (launches) *launches = r->launches;
    if (total_ms) *total_ms = r->total_ms;
    if (flops) *flops = r->flops;
    if (bytes) *bytes = r->bytes;
    return ASR_OK;
}

int asr_profile_symbol(asr_ctx *ctx, int index, char *symbol, int symbol_cap) {
    if (!ctx || index < 0 || index >= (int)ctx->prof.size() || !symbol || symbol_cap < 1) return ASR_ERR_INVALID;
    snprintf(symbol, (size_t)symbol_cap, "%s", ctx->prof[index]->symbol.c_str());
    return ASR_OK;
}

int asr_debug_activation(asr_ctx *ctx, int view, int block, int64_t n, float *out, int *h, int *w, int *c) {
    if (!ctx || view < 1 || view > 2 || block < 0 || block > 7) return ASR_ERR_INVALID;
    const Tower &t = ctx->tw[view - 1];
    const LayerGeom &g = t.g[block];
    if (h) *h = g.OH;
    if (w) *w = g.OW;
    if (c) *c = g.cout;
    if (!out) return ASR_OK;
    if (block == 0 && t.fuse1)
        return fail(ctx, ASR_ERR_STATE, "debug_activation: block 1 is fused into block 2 and never materialised; "
                                        "create the context with ASR_NO_FUSE1=1 to inspect it");
    if (n < 0 || n > ctx->last_n[view - 1] || !t.act[block])
        return fail(ctx, ASR_ERR_INVALID, "debug_activation: n=%lld but the last chunk held %d samples", (long long)n,
                    ctx->last_n[view - 1]);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    {
        int rcs = sync_all(ctx);
        if (rcs != ASR_OK) return rcs;
    }
    ASR_HIP(ctx, hipMemcpyAsync(out, t.act[block], (size_t)n * t.act_floats[block] * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_OK;
}

}  // extern "C"


// ===========================================================================
// training step (utils/train_dcca_pool.py:85-167 compiled `train` / `valid`)
// ===========================================================================
namespace {

inline float *pm(TrainState &T, int idx) { return T.pmaster + T.poff[idx]; }
inline float *pg(TrainState &T, int idx) { return T.pgrad + T.poff[idx]; }

int train_repack(asr_ctx *ctx) {
    // device master -> the layouts the kernels read (deterministic path included), on the main stream: ONE launch
    // driven by a table built once per training state (build_repack_table)
    TrainState &T = *ctx->train;
    ASR_HIP(ctx, asr::launch_repack_all(ctx->stream, T.repack_dev, T.n_repack));
    ctx->wino_stale = true;       // Winograd copies the step's own plans do not use are refreshed on demand (embedding)
    return ASR_OK;
}

// The forward and data-gradient convolutions of the training step are timed like the deterministic path's: every RAW
// Winograd schedule of a block (both tile orders of the global-A form, the LDS form's tilings at three budgets, the RAW
// F(4x4) build of the 48-channel blocks) on the
// step's own buffers at the step's batch size, the model's pick included; ~0.3 s once per asr_train_begin.
// ASR_AUTOTUNE=0 keeps the model's picks.  All candidates are the same kernels with other tile parameters: same results
// up to the float32 summation order of the Winograd transforms.
int tune_train_plans(asr_ctx *ctx, int B) {
    static const bool on = !(getenv("ASR_AUTOTUNE") && getenv("ASR_AUTOTUNE")[0] == '0') &&
                           !(getenv("ASR_TRAIN_TUNE") && getenv("ASR_TRAIN_TUNE")[0] == '0');
    if (!on) return ASR_OK;
    TrainState &T = *ctx->train;
    const bool dbg = getenv("ASR_DEBUG") != nullptr;
    hipStream_t st = ctx->stream;
    // ASR_TUNE_CACHE=<file>: the picks of an earlier asr_train_begin of this build on the same geometry and batch size
    // are re-used and new ones appended ("t1": forward / data-gradient schedules, "t2": weight-gradient tilings), so
    // that every rank of a data-parallel job and every restart run the same schedules (same float32 summation order)
    const char *cache = getenv("ASR_TUNE_CACHE");
    const int nf = ctx->cfg.num_filters, tag = tune_cache_tag();
    struct CacheLine { int k[12]; };
    std::vector<CacheLine> t1, t2;
    if (cache)
        if (FILE *fp = fopen(cache, "r")) {
            char line[256];
            while (fgets(line, sizeof line, fp)) {
                CacheLine c{};
                int ltag = 0;
                if (sscanf(line, "t1 %d %d %d %d %d %d %d %d %d %d %d %d", &ltag, &c.k[0], &c.k[1], &c.k[2], &c.k[3], &c.k[4],
                           &c.k[5], &c.k[6], &c.k[7], &c.k[8], &c.k[9], &c.k[10]) == 12 && ltag == tag)
                    t1.push_back(c);
                else if (sscanf(line, "t2 %d %d %d %d %d %d %d %d %d %d %d %d", &ltag, &c.k[0], &c.k[1], &c.k[2], &c.k[3],
                                &c.k[4], &c.k[5], &c.k[6], &c.k[7], &c.k[8], &c.k[9], &c.k[10]) == 12 && ltag == tag)
                    t2.push_back(c);
            }
            fclose(fp);
        }
    auto cache_append = [&](const char *kind, const int (&k)[11]) {
        if (!cache) return;
        if (FILE *fp = fopen(cache, "a")) {
            fprintf(fp, "%s %d %d %d %d %d %d %d %d %d %d %d %d\n", kind, tag, k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7],
                    k[8], k[9], k[10]);
            fclose(fp);
        }
    };
    hipEvent_t e0, e1;
    ASR_HIP(ctx, hipEventCreate(&e0));
    ASR_HIP(ctx, hipEventCreate(&e1));
    int rc = ASR_OK;
    for (int t = 0; t < 2 && rc == ASR_OK; ++t) {
        Tower &tw = ctx->tw[t];
        TrainTower &tt = T.tw[t];
        for (int b = 1; b < 8 && rc == ASR_OK; ++b) {
            const LayerGeom &g = tw.g[b];
            for (int dir = 0; dir < 2 && rc == ASR_OK; ++dir) {          // 0: forward x[b] -> z[b]; 1: data gradient dz -> dB
                asr::ConvPlan &plan = dir ? tt.dplan[b] : tt.fplan[b];
                if (plan.variant < 3000) continue;                       // direct schedule: nothing to choose from
                const int cin = dir ? g.cout : g.cin, cout = dir ? g.cin : g.cout;
                std::vector<asr::ConvPlan> cands;
                cands.push_back(plan);
                asr::conv_candidates_wino_raw(cin, cout, g.H, g.W, 2, &cands);
                asr::conv_candidates_wino4_raw(cin, cout, g.H, g.W, &cands, dir);
                const float *in = dir ? tt.dz : tt.x[b];
                const float *w = dir ? tt.wdgrad[b] : tw.w_dev[b];
                float *out = dir ? tt.dB : tt.z[b];
                {
                    int hit = -1;
                    for (auto &c : t1)
                        if (c.k[0] == nf && c.k[1] == t + 1 && c.k[2] == b && c.k[3] == g.H && c.k[4] == g.W && c.k[5] == B &&
                            c.k[6] == dir)
                            for (size_t q = 0; q < cands.size(); ++q)
                                if (cands[q].variant == c.k[7] && cands[q].TH == c.k[8] && cands[q].TW == c.k[9] &&
                                    cands[q].NI == c.k[10])
                                    hit = (int)q;
                    if (hit >= 0) {
                        plan = cands[hit];
                        if (dbg) fprintf(stderr, "[asr] train tune v%d conv%d %s from cache\n", t + 1, b + 1, dir ? "dgrad" : "fwd");
                        continue;
                    }
                }
                // data parallel with a shared tune cache: rank 0 timed every schedule before the others got here
                // (distributed.tune_in_rank_order).  A miss on another rank means the replicas would run different
                // float32 summation orders - and two ranks appending to one file: stop instead
                if (cache && ctx->comm && ctx->comm->world > 1 && ctx->comm->rank != 0) {
                    rc = fail(ctx, ASR_ERR_STATE, "train tuner: rank %d found no schedule for view %d conv%d (%s) in the job's "
                              "tune cache %s - rank 0 times, the other ranks read", ctx->comm->rank, t + 1, b + 1,
                              dir ? "dgrad" : "fwd", cache);
                    break;
                }
                // defined input values (0.5f): timing must not depend on stale bit patterns
                if (hipMemsetD32Async((hipDeviceptr_t)in, 0x3f000000, (size_t)B * g.H * g.W * cin, st) != hipSuccess) {
                    rc = fail(ctx, ASR_ERR_HIP, "tune_train_plans: memset");
                    break;
                }
                int best = 0;
                float best_ms = 1e30f;
                for (size_t c = 0; c < cands.size(); ++c) {
                    if (c > 0 && cands[c].variant == cands[0].variant && cands[c].TH == cands[0].TH &&
                        cands[c].TW == cands[0].TW && cands[c].NI == cands[0].NI)
                        continue;
                    hipError_t e = launch_conv_any(ctx, st, cands[c], in, w, nullptr, out, B);      // warm-up
                    if (e == hipSuccess) e = hipEventRecord(e0, st);
                    for (int r = 0; r < 2 && e == hipSuccess; ++r) e = launch_conv_any(ctx, st, cands[c], in, w, nullptr, out, B);
                    if (e == hipSuccess) e = hipEventRecord(e1, st);
                    if (e == hipSuccess) e = hipEventSynchronize(e1);
                    float ms = 0.f;
                    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
                    if (e != hipSuccess) { (void)hipGetLastError(); continue; }       // a candidate that cannot launch
                    ms *= (c == 0) ? 0.99f : 1.0f;                                    // ties go to the model's pick
                    if (dbg)
                        fprintf(stderr, "[asr] train tune v%d conv%d %s %s#%d tile %dx%d x%d: %.4f ms\n", t + 1, b + 1,
                                dir ? "dgrad" : "fwd", cands[c].variant >= 4000 ? "wino4" : cands[c].variant >= 3500 ? "winog" : "wino",
                                cands[c].variant,
                                cands[c].TH, cands[c].TW, cands[c].NI, ms / 2);
                    if (ms < best_ms) { best_ms = ms; best = (int)c; }
                }
                if (best_ms < 1e30f) {
                    plan = cands[best];
                    const int k[11] = {nf, t + 1, b, g.H, g.W, B, dir, plan.variant, plan.TH, plan.TW, plan.NI};
                    cache_append("t1", k);
                }
            }
            // the weight gradient: the planner's tiling against the next-cheapest tile shapes of its model
            if (rc == ASR_OK) {
                std::vector<asr::WgradPlan> wc;
                asr::wgrad_candidates(g.cin, g.cout, g.H, g.W, ctx->num_cus, 6, &wc);
                int best = -1;
                float best_ms = 1e30f;
                for (auto &c : t2)
                    if (c.k[0] == nf && c.k[1] == t + 1 && c.k[2] == b && c.k[3] == g.H && c.k[4] == g.W && c.k[5] == B)
                        for (size_t q = 0; q < wc.size(); ++q)
                            if (wc[q].variant == c.k[6] && wc[q].TH == c.k[7] && wc[q].TW == c.k[8] &&
                                wc[q].lds_bytes == c.k[9] && wc[q].grid_cap == c.k[10] &&
                                asr::wgrad_partial_floats(wc[q]) <= tt.wpartial_floats)
                                best = (int)q;
                if (best >= 0) {
                    tt.wplan[b] = wc[best];
                    if (dbg) fprintf(stderr, "[asr] train tune v%d conv%d wgrad from cache\n", t + 1, b + 1);
                    continue;
                }
                if (wc.size() > 1 && cache && ctx->comm && ctx->comm->world > 1 && ctx->comm->rank != 0) {
                    rc = fail(ctx, ASR_ERR_STATE, "train tuner: rank %d found no weight-gradient schedule for view %d conv%d in "
                              "the job's tune cache %s - rank 0 times, the other ranks read", ctx->comm->rank, t + 1, b + 1, cache);
                    break;
                }
                // defined operands for every candidate (x[b] was filled above only when the forward plan is Winograd)
                if (wc.size() > 1 &&
                    (hipMemsetD32Async((hipDeviceptr_t)tt.x[b], 0x3f000000, (size_t)B * g.H * g.W * g.cin, st) != hipSuccess ||
                     hipMemsetD32Async((hipDeviceptr_t)tt.dz, 0x3f000000, (size_t)B * g.H * g.W * g.cout, st) != hipSuccess)) {
                    rc = fail(ctx, ASR_ERR_HIP, "tune_train_plans: memset");
                    break;
                }
                for (size_t c = 0; c < wc.size() && wc.size() > 1; ++c) {
                    if (asr::wgrad_partial_floats(wc[c]) > tt.wpartial_floats) continue;
                    hipError_t e = asr::launch_wgrad(st, wc[c], tt.x[b], tt.dz, B, tt.wpartial, pg(T, 45 * t + 5 * b));
                    if (e == hipSuccess) e = hipEventRecord(e0, st);
                    for (int r = 0; r < 2 && e == hipSuccess; ++r)
                        e = asr::launch_wgrad(st, wc[c], tt.x[b], tt.dz, B, tt.wpartial, pg(T, 45 * t + 5 * b));
                    if (e == hipSuccess) e = hipEventRecord(e1, st);
                    if (e == hipSuccess) e = hipEventSynchronize(e1);
                    float ms = 0.f;
                    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
                    if (e != hipSuccess) { (void)hipGetLastError(); continue; }
                    ms *= (c == 0) ? 0.99f : 1.0f;
                    if (dbg)
                        fprintf(stderr, "[asr] train tune v%d conv%d wgrad variant %d tile %dx%d lds %d, %d workgroups: %.4f ms\n",
                                t + 1, b + 1, wc[c].variant, wc[c].TH, wc[c].TW, wc[c].lds_bytes, wc[c].grid_cap, ms / 2);
                    if (ms < best_ms) { best_ms = ms; best = (int)c; }
                }
                if (best >= 0) {
                    tt.wplan[b] = wc[best];
                    const int k[11] = {nf, t + 1, b, g.H, g.W, B, wc[best].variant, wc[best].TH, wc[best].TW,
                                       wc[best].lds_bytes, wc[best].grid_cap};
                    cache_append("t2", k);
                }
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int build_repack_table(asr_ctx *ctx, bool all4 = false) {
    TrainState &T = *ctx->train;
    std::vector<asr::RepackDesc> descs;
    for (int t = 0; t < 2; ++t) {
        Tower &tw = ctx->tw[t];
        for (int b = 0; b < 9; ++b) {
            const LayerGeom &g = tw.g[b];
            const int base = 45 * t + 5 * b;
            asr::RepackDesc d{};
            d.W = pm(T, base); d.beta = pm(T, base + 1); d.gamma = pm(T, base + 2);
            d.mean = pm(T, base + 3); d.istd = pm(T, base + 4);
            d.cin = g.cin; d.cout = g.cout;
            d.bnp = tw.bn_dev[b];
            if (b == 0) {
                d.kind = 0;
                d.wfwd = tw.w_dev[0];
            } else if (b < 8) {
                d.kind = 1;
                d.wfwd = tw.w_dev[b];
                d.wdgrad = T.tw[t].wdgrad[b];
                // Winograd-domain copies only where the training step's own plans use them
                if (T.tw[t].fplan[b].variant >= 3000) d.wino_fwd = tw.w_dev[b] + asr::conv_wpack_floats(g.cin, g.cout);
                if (T.tw[t].dplan[b].variant >= 3000)
                    d.wino_dgrad = T.tw[t].wdgrad[b] + asr::conv_wpack_floats(g.cout, g.cin);
                // F(4x4) copies: while the tuner has not run (`all4`) wherever a RAW F(4x4) build exists, afterwards only
                // where a plan uses one
                {
                    std::vector<asr::ConvPlan> c4f, c4d;
                    asr::conv_candidates_wino4_raw(g.cin, g.cout, g.H, g.W, &c4f, 0);
                    asr::conv_candidates_wino4_raw(g.cout, g.cin, g.H, g.W, &c4d, 1);
                    if (!c4f.empty() && (all4 || T.tw[t].fplan[b].variant >= 4000))
                        d.wino4_fwd = tw.w_dev[b] + asr::conv_wpack_floats(g.cin, g.cout) + asr::wino_wpack_floats(g.cin, g.cout);
                    if (!c4d.empty() && (all4 || T.tw[t].dplan[b].variant >= 4000))
                        d.wino4_dgrad = T.tw[t].wdgrad[b] + asr::conv_wpack_floats(g.cout, g.cin) +
                                        asr::wino_wpack_floats(g.cout, g.cin);
                }
            } else {
                d.kind = 2;                               // 1x1 conv: [o][c] as stored
                d.wfwd = tw.w_dev[8];
                d.cin = g.cin; d.cout = 32;
            }
            descs.push_back(d);
        }
    }
    asr::RepackDesc c{};                                  // CCALayer block U V mean1 mean2 (contiguous in the master)
    c.kind = 2; c.W = pm(T, 90); c.wfwd = ctx->cca_dev; c.cin = 1; c.cout = 2048 + 64;
    descs.push_back(c);
    if (T.repack_dev) ASR_HIP(ctx, hipFree(T.repack_dev));
    T.repack_dev = nullptr;
    ASR_HIP(ctx, hipMalloc((void **)&T.repack_dev, descs.size() * sizeof(asr::RepackDesc)));
    ASR_HIP(ctx, hipMemcpy(T.repack_dev, descs.data(), descs.size() * sizeof(asr::RepackDesc), hipMemcpyHostToDevice));
    T.n_repack = (int)descs.size();
    return ASR_OK;
}

// Winograd-domain weights of all conv blocks, rebuilt after training steps moved the parameters: from the device
// master while a training state exists, otherwise from the host mirror (asr_train_end downloads the master first).
// The flag is only ever cleared by a completed refresh.
int refresh_wino_weights(asr_ctx *ctx) {
    if (!ctx->wino_stale) return ASR_OK;
    float *raw = nullptr;
    for (int t = 0; t < 2; ++t)
        for (int b = 1; b < 8; ++b) {
            const LayerGeom &g = ctx->tw[t].g[b];
            const float *src;
            if (ctx->train) {
                src = pm(*ctx->train, 45 * t + 5 * b);
            } else {
                const std::vector<float> &W = ctx->params[45 * t + 5 * b];
                if (!raw) ASR_HIP(ctx, hipMalloc((void **)&raw, (size_t)96 * 96 * 9 * sizeof(float)));
                hipError_t e = hipMemcpyAsync(raw, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) { (void)hipFree(raw); ASR_HIP(ctx, e); }
                src = raw;
            }
            hipError_t e = asr::launch_wino_pack(ctx->stream, src, g.cin, g.cout,
                                                 ctx->tw[t].w_dev[b] + asr::conv_wpack_floats(g.cin, g.cout));
            if (e == hipSuccess)
                e = asr::launch_wino4_pack(ctx->stream, src, g.cin, g.cout,
                                           ctx->tw[t].w_dev[b] + asr::conv_wpack_floats(g.cin, g.cout) +
                                               asr::wino_wpack_floats(g.cin, g.cout));
            if (e == hipSuccess && !ctx->train) e = hipStreamSynchronize(ctx->stream);   // `raw` is re-used per block
            if (e != hipSuccess) { (void)hipFree(raw); ASR_HIP(ctx, e); }
        }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(raw);
    ASR_HIP(ctx, e);
    ctx->wino_stale = false;
    return ASR_OK;
}

}  // namespace

static int train_upload_master(asr_ctx *ctx) {
    TrainState &T = *ctx->train;
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    std::vector<float> flat((size_t)T.ptotal);
    for (size_t i = 0; i < ctx->params.size(); ++i)
        memcpy(flat.data() + T.poff[i], ctx->params[i].data(), ctx->params[i].size() * sizeof(float));
    ASR_HIP(ctx, hipMemcpyAsync(T.pmaster, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T.master_dirty = false;
    rc = train_repack(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_OK;
}

static int train_download_master(asr_ctx *ctx) {
    TrainState &T = *ctx->train;
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    std::vector<float> flat((size_t)T.ptotal);
    ASR_HIP(ctx, hipMemcpyAsync(flat.data(), T.pmaster, flat.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->params.size(); ++i)
        memcpy(ctx->params[i].data(), flat.data() + T.poff[i], ctx->params[i].size() * sizeof(float));
    T.master_dirty = false;
    return ASR_OK;
}

namespace {

// ---- collectives ------------------------------------------------------------------------------------
// RCCL calls are enqueued on the stream; a host callback is host-synchronous (the stream is drained first and the
// callback returns with the result in place).
int comm_allreduce(asr_ctx *ctx, hipStream_t st, void *buf, int64_t count, int dtype) {
    Comm *c = ctx->comm.get();
    if (!c || (c->world <= 1 && !c->force) || count <= 0) return ASR_OK;
    if (dtype != ASR_DTYPE_F32 && dtype != ASR_DTYPE_F64 && dtype != ASR_DTYPE_I32)
        return fail(ctx, ASR_ERR_INVALID, "comm: all-reduce dtype %d", dtype);
    c->n_allreduce += 1;
    c->b_allreduce += count * (dtype == ASR_DTYPE_F64 ? 8 : 4);
    if (c->ar) {
        ASR_HIP(ctx, hipStreamSynchronize(st));
        if (c->ar(c->user, buf, count, dtype) != 0) return fail(ctx, ASR_ERR_STATE, "comm: all-reduce callback failed");
        return ASR_OK;
    }
    const ncclResult_t r = c->pAllReduce(buf, buf, (size_t)count,
                                         dtype == ASR_DTYPE_F64 ? ncclFloat64 : dtype == ASR_DTYPE_I32 ? ncclInt32 : ncclFloat32,
                                         ncclSum, c->nccl, st);
    if (r != ncclSuccess) return fail(ctx, ASR_ERR_HIP, "comm: ncclAllReduce: %s", c->pGetErrorString(r));
    return ASR_OK;
}

int comm_allgather(asr_ctx *ctx, hipStream_t st, const void *send, void *recv, int64_t bytes_per_rank) {
    Comm *c = ctx->comm.get();
    if (!c || (c->world <= 1 && !c->force)) {
        if (send != recv) ASR_HIP(ctx, hipMemcpyAsync(recv, send, (size_t)bytes_per_rank, hipMemcpyDeviceToDevice, st));
        return ASR_OK;
    }
    c->n_allgather += 1;
    c->b_allgather += bytes_per_rank;
    if (c->ag) {
        ASR_HIP(ctx, hipStreamSynchronize(st));
        if (c->ag(c->user, send, recv, bytes_per_rank) != 0)
            return fail(ctx, ASR_ERR_STATE, "comm: all-gather callback failed");
        return ASR_OK;
    }
    const ncclResult_t r = c->pAllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, c->nccl, st);
    if (r != ncclSuccess) return fail(ctx, ASR_ERR_HIP, "comm: ncclAllGather: %s", c->pGetErrorString(r));
    return ASR_OK;
}

int exch_allreduce_f64(void *self, hipStream_t s, double *buf, int64_t count) {
    return comm_allreduce(static_cast<asr_ctx *>(self), s, buf, count, ASR_DTYPE_F64);
}

int comm_world(const asr_ctx *ctx) { return ctx->comm ? ctx->comm->world : 1; }
int comm_rank(const asr_ctx *ctx) { return ctx->comm ? ctx->comm->rank : 0; }
bool comm_active(const asr_ctx *ctx) { return ctx->comm && (ctx->comm->world > 1 || ctx->comm->force); }
// data-parallel training keeps both towers on the main stream: one communicator, one issue order on every rank
// (ASR_TRAIN_ONE_STREAM=1: also without a communicator - per-stage timings that no concurrent kernel stretches)
hipStream_t train_stream(asr_ctx *ctx, int t) {
    static const bool one = getenv("ASR_TRAIN_ONE_STREAM") && getenv("ASR_TRAIN_ONE_STREAM")[0] == '1';
    return (one || comm_active(ctx)) ? ctx->stream : ctx->vstream[t];
}
const asr::Exchange *train_exch(asr_ctx *ctx) { return comm_active(ctx) ? &ctx->exch : nullptr; }

void install_comm(asr_ctx *ctx, std::unique_ptr<Comm> c) {
    ctx->exch.allreduce_f64 = exch_allreduce_f64;
    ctx->exch.self = ctx;
    ctx->exch.world = c->world;
    const char *f = getenv("ASR_COMM_FORCE");
    c->force = f && f[0] == '1';
    ctx->comm = std::move(c);
}

int train_alloc(asr_ctx *ctx, int B) {
    free_train(ctx);
    ctx->train.reset(new TrainState());
    TrainState &T = *ctx->train;
    T.B = B;
    T.poff.resize(ctx->params.size() + 1);
    T.poff[0] = 0;
    for (size_t i = 0; i < ctx->params.size(); ++i) T.poff[i + 1] = T.poff[i] + (int64_t)ctx->params[i].size();
    T.ptotal = T.poff.back();
    const size_t pb = (size_t)T.ptotal * sizeof(float);
    ASR_HIP(ctx, hipMalloc((void **)&T.pmaster, pb));
    ASR_HIP(ctx, hipMalloc((void **)&T.pgrad, pb));
    ASR_HIP(ctx, hipMalloc((void **)&T.adam_m, pb));
    ASR_HIP(ctx, hipMalloc((void **)&T.adam_v, pb));
    ASR_HIP(ctx, hipMalloc((void **)&T.mask, (size_t)T.ptotal));
    ASR_HIP(ctx, hipMemsetAsync(T.pgrad, 0, pb, ctx->stream));
    ASR_HIP(ctx, hipMemsetAsync(T.adam_m, 0, pb, ctx->stream));
    ASR_HIP(ctx, hipMemsetAsync(T.adam_v, 0, pb, ctx->stream));
    std::vector<unsigned char> mask((size_t)T.ptotal, 0);
    for (int i = 0; i < 90; ++i)
        if (i % 5 <= 2) std::fill(mask.begin() + T.poff[i], mask.begin() + T.poff[i + 1], (unsigned char)1);
    ASR_HIP(ctx, hipMemcpyAsync(T.mask, mask.data(), mask.size(), hipMemcpyHostToDevice, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T.adam_t = 0;
    T.world = comm_world(ctx);
    ASR_HIP(ctx, hipMalloc(&T.cca_ws, asr::cca_train_ws_bytes(B * T.world)));
    // zero: the "eigenvectors of the previous step are valid" flag of the warm-started Jacobi lives in there
    ASR_HIP(ctx, hipMemsetAsync(T.cca_ws, 0, asr::cca_train_ws_bytes(B * T.world), ctx->stream));
    if (comm_active(ctx))
        for (int t = 0; t < 2; ++t) {
            const size_t gb = (size_t)B * T.world * 32 * sizeof(float);
            ASR_HIP(ctx, hipMalloc((void **)&T.Hg[t], gb));
            ASR_HIP(ctx, hipMalloc((void **)&T.dHg[t], gb));
            ASR_HIP(ctx, hipMalloc((void **)&T.lvg[t], gb));
            ASR_HIP(ctx, hipMalloc((void **)&T.Hpad[t], gb));
        }
    ASR_HIP(ctx, hipMalloc((void **)&T.loss_dev, 64 * sizeof(float)));
    ASR_HIP(ctx, hipMalloc((void **)&T.l2_dev, sizeof(double)));
    ASR_HIP(ctx, hipEventCreateWithFlags(&T.cca_done, hipEventDisableTiming));
    for (int v = 0; v < 2; ++v) ASR_HIP(ctx, hipMalloc((void **)&T.lvv[v], (size_t)B * 32 * sizeof(float)));

    for (int t = 0; t < 2; ++t) {
        Tower &tw = ctx->tw[t];
        TrainTower &tt = T.tw[t];
        size_t max_z = 0, max_x = 0, max_wp = 0;
        size_t max_partial = 0;
        for (int b = 0; b < 9; ++b) {
            const LayerGeom &g = tw.g[b];
            const size_t xin = (size_t)B * g.H * g.W * g.cin;
            const size_t zo = (size_t)B * g.H * g.W * g.cout;
            ASR_HIP(ctx, hipMalloc((void **)&tt.x[b], xin * sizeof(float)));
            ASR_HIP(ctx, hipMalloc((void **)&tt.z[b], zo * sizeof(float)));
            ASR_HIP(ctx, hipMalloc((void **)&tt.stats[b], (size_t)2 * g.cout * sizeof(float)));
            // pooled blocks: the raw value of every pooling window's selected element, written by the forward apply pass
            // for the reduce pass of the BatchNorm backward (ASR_TRAIN_ZSEL=0: that pass re-reads the four window elements)
            static const bool use_zsel = !(getenv("ASR_TRAIN_ZSEL") && getenv("ASR_TRAIN_ZSEL")[0] == '0');
            if (b < 8 && g.pool && use_zsel)
                ASR_HIP(ctx, hipMalloc((void **)&tt.zsel[b], (size_t)B * (g.H / 2) * (g.W / 2) * g.cout * sizeof(float)));
            if (b < 8) max_z = std::max(max_z, zo);
            if (b >= 1) max_x = std::max(max_x, xin);
            const int64_t rows = (int64_t)B * g.H * g.W;
            max_partial = std::max(max_partial, (size_t)asr::bn_stats_blocks(rows) * 2 * g.cout);
            max_partial = std::max(max_partial, (size_t)asr::bn_bwd_blocks(rows) * 2 * g.cout);
            if (b >= 1 && b < 8) {
                if ((!asr::plan_conv_wino_raw(g.cin, g.cout, g.H, g.W, &tt.fplan[b]) &&
                     !asr::plan_conv_v3_raw(g.cin, g.cout, g.H, g.W, &tt.fplan[b]) &&
                     !asr::plan_conv(g.cin, g.cout, 0, g.H, g.W, &tt.fplan[b], 1)) ||
                    (!asr::plan_conv_wino_raw(g.cout, g.cin, g.H, g.W, &tt.dplan[b]) &&
                     !asr::plan_conv_v3_raw(g.cout, g.cin, g.H, g.W, &tt.dplan[b]) &&
                     !asr::plan_conv(g.cout, g.cin, 0, g.H, g.W, &tt.dplan[b], 1)) ||
                    !asr::plan_wgrad(g.cin, g.cout, g.H, g.W, ctx->num_cus, &tt.wplan[b]))
                    return fail(ctx, ASR_ERR_INVALID, "train: no kernel variant for block %d (%d->%d)", b + 1, g.cin, g.cout);
                // ASR_TRAIN_WINO4=2: the RAW F(4x4) build wherever one exists (otherwise only the training tuner picks it, and
                // by default only for data gradients); 3: forward convolutions only, 4: data gradients only
                static const int force4 = getenv("ASR_TRAIN_WINO4") ? atoi(getenv("ASR_TRAIN_WINO4")) : 0;
                if (force4 >= 2) {
                    std::vector<asr::ConvPlan> c4;
                    asr::conv_candidates_wino4_raw(g.cin, g.cout, g.H, g.W, &c4, 0);
                    if (!c4.empty() && force4 != 4) tt.fplan[b] = c4[0];
                    c4.clear();
                    asr::conv_candidates_wino4_raw(g.cout, g.cin, g.H, g.W, &c4, 1);
                    if (!c4.empty() && force4 != 3) tt.dplan[b] = c4[0];
                }
                max_wp = std::max(max_wp, asr::wgrad_partial_floats(tt.wplan[b]));
                // data-gradient weights: direct-form fragments, then the Winograd-domain copy (same layout as w_dev)
                ASR_HIP(ctx, hipMalloc((void **)&tt.wdgrad[b], (asr::conv_wpack_floats(g.cout, g.cin) +
                                                                asr::wino_wpack_floats(g.cout, g.cin) +
                                                                asr::wino4_wpack_floats(g.cout, g.cin)) * sizeof(float)));
            }
        }
        const LayerGeom &g8 = tw.g[8];
        max_partial = std::max(max_partial, (size_t)asr::tail_dw_blocks((int64_t)B * g8.H * g8.W) * 32 * g8.cin +
                                                (size_t)std::max(256, B) * 64);      // + one row of 64 per sample (tail_bwd)
        max_partial = std::max(max_partial, (size_t)asr::conv1_wgrad_blocks() * tw.g[0].cout * 9);
        // statistics tables written by the convolutions themselves: one row per wave (Winograd) / workgroup (block 1)
        max_partial = std::max(max_partial, (size_t)std::max(std::max(asr::conv_wino_stats_rows_max(ctx->num_cus),
                                                                      asr::conv_wino4_stats_rows_max(ctx->num_cus)), 4096) * 2 *
                                                (size_t)tw.g[7].cout);
        ASR_HIP(ctx, hipMalloc((void **)&tt.dz, max_z * sizeof(float)));
        {
            static const bool wside = !(getenv("ASR_TRAIN_WGRAD_STREAM") && getenv("ASR_TRAIN_WGRAD_STREAM")[0] == '0');
            const bool one = getenv("ASR_TRAIN_ONE_STREAM") && getenv("ASR_TRAIN_ONE_STREAM")[0] == '1';
            if (wside && !one && !comm_active(ctx)) {
                ASR_HIP(ctx, hipMalloc((void **)&tt.dz2, max_z * sizeof(float)));
                ASR_HIP(ctx, hipStreamCreateWithFlags(&tt.wstream, hipStreamNonBlocking));
                for (int k = 0; k < 2; ++k) {
                    ASR_HIP(ctx, hipEventCreateWithFlags(&tt.e_dz[k], hipEventDisableTiming));
                    ASR_HIP(ctx, hipEventCreateWithFlags(&tt.e_wg[k], hipEventDisableTiming));
                }
            }
        }
        ASR_HIP(ctx, hipMalloc((void **)&tt.dA, max_x * sizeof(float)));
        ASR_HIP(ctx, hipMalloc((void **)&tt.dB, max_x * sizeof(float)));
        ASR_HIP(ctx, hipMalloc((void **)&tt.H, (size_t)B * 32 * sizeof(float)));
        ASR_HIP(ctx, hipMalloc((void **)&tt.dH, (size_t)B * 32 * sizeof(float)));
        ASR_HIP(ctx, hipMalloc((void **)&tt.lv, (size_t)B * 32 * sizeof(float)));
        ASR_HIP(ctx, hipMalloc((void **)&tt.partial, (max_partial + asr::colsum_stage_extra(max_partial)) * sizeof(double)));
        max_wp = std::max<size_t>(max_wp * 2, 1);             // room for the tuner's picks (more workgroups per CU)
        ASR_HIP(ctx, hipMalloc((void **)&tt.wpartial, max_wp * sizeof(float)));
        tt.wpartial_floats = max_wp;
        if (t == 0) {               // one allocation for both towers (train_pair_allreduce): [tower 1: 512 | tower 2: 512]
            ASR_HIP(ctx, hipMalloc((void **)&tt.sums, 1024 * sizeof(double)));
            ASR_HIP(ctx, hipMemsetAsync(tt.sums, 0, 1024 * sizeof(double), ctx->stream));
        } else {
            tt.sums = T.tw[0].sums + 512;
        }
    }
    {
        // the tuner times the real kernels on the real weights: master and every derived layout first.  (A plan the tuner
        // replaces stays inside its family - Winograd for Winograd - so the repack table does not change afterwards.)
        int rct = build_repack_table(ctx, true);
        if (rct != ASR_OK) return rct;
        rct = train_upload_master(ctx);
        if (rct != ASR_OK) return rct;
        rct = tune_train_plans(ctx, B);
        if (rct != ASR_OK) return rct;
        rct = build_repack_table(ctx);                        // only the layouts the final plans read
        if (rct != ASR_OK) return rct;
        // the weight-gradient candidates wrote into the gradient buffer
        ASR_HIP(ctx, hipMemsetAsync(T.pgrad, 0, (size_t)T.ptotal * sizeof(float), ctx->stream));
        ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ASR_OK;
}

// Block 1 of the training step without its raw tensor: statistics pass + apply pass in the forward direction, z
// recomputed from the image by its two backward readers (needs the fused statistics and the fused block-1 apply)
static bool train_recompute1() {
    static const bool on = !(getenv("ASR_TRAIN_RECOMPUTE1") && getenv("ASR_TRAIN_RECOMPUTE1")[0] == '0') &&
                           !(getenv("ASR_TRAIN_FUSE_STATS") && getenv("ASR_TRAIN_FUSE_STATS")[0] == '0') &&
                           !(getenv("ASR_TRAIN_FUSE_BN1") && getenv("ASR_TRAIN_FUSE_BN1")[0] == '0');
    return on;
}

// One block of one tower's train-mode forward pass.  phase 0: all of it.  Data parallel, the two towers' exchanges paired
// (see asr::Exchange::phase): phase 1 = convolution + the local BatchNorm sums, phase 2 = statistics from the summed
// sums + the apply pass.
int train_forward_block(asr_ctx *ctx, int t, int B, int b, int phase) {
    TrainState &T = *ctx->train;
    Tower &tw = ctx->tw[t];
    TrainTower &tt = T.tw[t];
    hipStream_t st = train_stream(ctx, t);
    const asr::Exchange *ex = train_exch(ctx);
    const int view = t + 1;
    ctx->exch.phase = phase;
    {
        const LayerGeom &g = tw.g[b];
        const int base = 45 * t + 5 * b;
        const int64_t rows = (int64_t)B * g.H * g.W;
        char name[32];
        snprintf(name, sizeof name, "train_fwd_conv%d", b + 1);
        // BatchNorm statistics: the RAW Winograd kernels and the block-1 kernel gather the per-channel sums of z in
        // their epilogues (a partial table of `srows` rows); other plans leave srows = 0 and z is re-read once
        static const bool fuse_stats = !(getenv("ASR_TRAIN_FUSE_STATS") && getenv("ASR_TRAIN_FUSE_STATS")[0] == '0');
        int srows = 0;
        if (phase != 2) {
            // algorithmic bytes: input read, raw output written - block 1 in the recompute form stores nothing (its
            // statistics pass only reads the image)
            ProfScope ps(ctx, name, view, 2.0 * rows * g.k * g.k * g.cin * g.cout,
                         4.0 * rows * (g.cin + ((b == 0 && train_recompute1()) ? 0 : g.cout)),
                         b >= 1 && b < 8 ? tt.fplan[b].symbol : "");
            if (b == 0)
                ASR_HIP(ctx, asr::launch_conv1_raw(st, tt.x[0], tw.w_dev[0], tt.z[0], B, g.H, g.W, g.cout,
                                                   fuse_stats ? tt.partial : nullptr, &srows, train_recompute1() ? 1 : 0));
            else if (b < 8)
                ASR_HIP(ctx, launch_conv_any(ctx, st, tt.fplan[b], tt.x[b], tw.w_dev[b], nullptr, tt.z[b], B, nullptr,
                                             fuse_stats ? tt.partial : nullptr, &srows));
            else ASR_HIP(ctx, asr::launch_conv1x1_raw(st, tt.x[8], pm(T, base), tt.z[8], rows, g.cin));
            if (!fuse_stats) srows = 0;
        }
        char bname[32];
        snprintf(bname, sizeof bname, "train_fwd_bn%d", b + 1);
        // bytes: z read (once more when the statistics were not gathered by the convolution), pooled output written
        // (block 1, recompute form: the apply pass runs the stencil on the image again - it reads the image, not z)
        ProfScope ps2(ctx, bname, view, 6.0 * rows * g.cout + ((b == 0 && train_recompute1()) ? 2.0 * rows * 9.0 * g.cout : 0.0),
                      (b == 0 && train_recompute1())
                          ? 4.0 * rows * (g.cin + g.cout)
                          : 4.0 * rows * g.cout * ((srows ? 1.0 : 2.0) + (b == 8 ? 0.0 : g.pool ? (tt.zsel[b] ? 0.5 : 0.25) : 1.0)));
        if (srows > 0 || phase == 2)          // (phase 2: only the finish from the all-reduced sums runs)
            ASR_HIP(ctx, asr::launch_bn_stats_final(st, tt.partial, std::max(srows, 1), rows, g.cout, tt.stats[b], pm(T, base + 3),
                                                    pm(T, base + 4), 1e-4f, 0.1f, ex, tt.sums));
        else
            ASR_HIP(ctx, asr::launch_bn_stats(st, tt.z[b], rows, g.cout, tt.partial, tt.stats[b], pm(T, base + 3),
                                              pm(T, base + 4), 1e-4f, 0.1f, ex, tt.sums));
        if (phase == 1) { ctx->exch.phase = 0; return ASR_OK; }
        if (b == 0 && train_recompute1())
            ASR_HIP(ctx, asr::launch_conv1_raw(st, tt.x[0], tw.w_dev[0], tt.x[1], B, g.H, g.W, g.cout, nullptr, nullptr, 2,
                                               tt.stats[0], pm(T, base + 2), pm(T, base + 1)));
        else if (b < 8)
            ASR_HIP(ctx, asr::launch_bn_apply(st, tt.z[b], tt.stats[b], pm(T, base + 2), pm(T, base + 1), tt.x[b + 1],
                                              B, g.H, g.W, g.cout, g.pool, 1, tt.zsel[b]));
        else
            ASR_HIP(ctx, asr::launch_bn_gpool(st, tt.z[8], tt.stats[8], pm(T, base + 2), pm(T, base + 1), tt.H, B,
                                              g.H * g.W));
    }
    ctx->exch.phase = 0;
    return ASR_OK;
}

// Sum the two towers' BatchNorm sums over the ranks in ONE all-reduce: tower 2's `sums` follow tower 1's at a distance of
// 512 doubles in one allocation (train_alloc), so the pair is the contiguous range [0, 512 + count) - the unused middle
// is zeros.  18 + 18 + 1 all-reduces per update become 9 + 9 + 1 (each costs ~15 us even among ONE rank).
int train_pair_allreduce(asr_ctx *ctx, int count) {
    return comm_allreduce(ctx, ctx->stream, ctx->train->tw[0].sums, 512 + count, ASR_DTYPE_F64);
}

int train_forward_towers(asr_ctx *ctx, int B) {
    int rc;
    if (comm_active(ctx)) {               // both towers share the main stream here: block by block, exchanges paired
        for (int b = 0; b < 9; ++b) {
            for (int t = 0; t < 2; ++t)
                if ((rc = train_forward_block(ctx, t, B, b, 1)) != ASR_OK) return rc;
            if ((rc = train_pair_allreduce(ctx, 2 * ctx->tw[0].g[b].cout)) != ASR_OK) return rc;
            for (int t = 0; t < 2; ++t)
                if ((rc = train_forward_block(ctx, t, B, b, 2)) != ASR_OK) return rc;
        }
    } else {
        for (int t = 0; t < 2; ++t)
            for (int b = 0; b < 9; ++b)
                if ((rc = train_forward_block(ctx, t, B, b, 0)) != ASR_OK) return rc;
    }
    for (int t = 0; t < 2; ++t) {
        ASR_HIP(ctx, hipEventRecord(ctx->vdone[t], train_stream(ctx, t)));
        ctx->vpending[t] = true;
    }
    return ASR_OK;
}

struct BwdState {
    float *dA, *dB;                 // gradients wrt block outputs (rotating)
    bool wg_pending[2];
};

// the end of a tower's backward pass (block 9, global pooling); phases as in train_forward_block
int train_backward_tail(asr_ctx *ctx, int t, int B, int64_t row_lo, int phase, BwdState &S) {
    TrainState &T = *ctx->train;
    Tower &tw = ctx->tw[t];
    TrainTower &tt = T.tw[t];
    hipStream_t st = train_stream(ctx, t);
    const asr::Exchange *ex = train_exch(ctx);
    const int view = t + 1;
    if (phase != 2) {
        ASR_HIP(ctx, hipStreamWaitEvent(st, T.cca_done, 0));
        S.dA = tt.dA; S.dB = tt.dB;
        S.wg_pending[0] = S.wg_pending[1] = false;
    }
    // data parallel: this rank's rows of the full-batch dL/dH
    const float *dH = ex ? T.dHg[t] + (size_t)row_lo * 32 : tt.dH;
    ctx->exch.phase = phase;
    {
        const LayerGeom &g = tw.g[8];
        ProfScope ps(ctx, "train_bwd_tail", view, 6.0 * B * g.H * g.W * g.cin * 32.0, 0.0);
        ASR_HIP(ctx, asr::launch_tail_bwd(st, dH, tt.z[8], tt.x[8], pm(T, 45 * t + 40), tt.stats[8],
                                          pm(T, 45 * t + 42), B, g.H * g.W, g.cin, tt.sums, tt.partial,
                                          pg(T, 45 * t + 41), pg(T, 45 * t + 42), pg(T, 45 * t + 40), S.dA, ex));
    }
    ctx->exch.phase = 0;
    return ASR_OK;
}

// one block (b = 7..0) of a tower's backward pass: BatchNorm backward, weight gradient, data gradient
int train_backward_block(asr_ctx *ctx, int t, int B, int b, int phase, BwdState &S) {
    TrainState &T = *ctx->train;
    Tower &tw = ctx->tw[t];
    TrainTower &tt = T.tw[t];
    hipStream_t st = train_stream(ctx, t);
    const asr::Exchange *ex = train_exch(ctx);
    const int view = t + 1;
    float *&dA = S.dA, *&dB = S.dB;
    bool (&wg_pending)[2] = S.wg_pending;
    // Weight gradients run on the tower's side stream when it has one: wgrad(b) needs dz(b) and x(b) only, and while it
    // multiplies (MFMA-bound) the main stream goes on with the data gradient and the BatchNorm backward of block b - 1
    // (HBM-bound).  dz alternates between two buffers; bn_bwd(b - 2) waits for wgrad(b) before it overwrites dz(b)'s.
    hipStream_t ws = tt.wstream ? tt.wstream : st;
    float *dzb[2] = {tt.dz, tt.dz2 ? tt.dz2 : tt.dz};
    static const bool fuse1 = !(getenv("ASR_TRAIN_FUSE_BN1") && getenv("ASR_TRAIN_FUSE_BN1")[0] == '0');
    ctx->exch.phase = phase;
    {
        const LayerGeom &g = tw.g[b];
        const int base = 45 * t + 5 * b;
        const double rows = (double)B * g.H * g.W;
        const int cur = b & 1;
        float *dz = dzb[cur];
        if (phase != 2 && tt.wstream && wg_pending[cur]) {
            ASR_HIP(ctx, hipStreamWaitEvent(st, tt.e_wg[cur], 0));
            wg_pending[cur] = false;
        }
        {
            char bname[32];
            snprintf(bname, sizeof bname, "train_bwd_bn%d", b + 1);
            // bytes: z and the pooled gradient read by both passes, dz written (pooled blocks with zsel: the reduce pass
            // reads one selected value per window instead of z).  Block 1: only the reduce pass runs
            // here (the apply pass lives in the weight-gradient kernel) - it reads z and the gradient once, or, in the
            // recompute form, the image and the gradient (z is recomputed, never read)
            const double bn_bytes = (b == 0 && train_recompute1()) ? 4.0 * rows * (g.cin + g.cout)
                                    : (b == 0 && fuse1)            ? 4.0 * rows * g.cout * 2.0
                                    : (g.pool && tt.zsel[b])       ? 4.0 * rows * g.cout * 2.75   // (zsel + dA) + (z + dA + dz)
                                                                   : 4.0 * rows * g.cout * (3.0 + (g.pool ? 0.5 : 2.0));
            ProfScope ps(ctx, bname, view, 12.0 * rows * g.cout + ((b == 0 && train_recompute1()) ? 2.0 * rows * 9.0 * g.cout : 0.0),
                         bn_bytes);
            // block 1: the apply pass is fused into the weight-gradient kernel, dz's only reader there
            if (b == 0 && train_recompute1())
                ASR_HIP(ctx, asr::launch_bn_bwd_conv1(st, tt.x[0], tw.w_dev[0], dA, tt.stats[0], pm(T, base + 2), pm(T, base + 1),
                                                      tt.partial, tt.sums, pg(T, base + 1), pg(T, base + 2), B, g.H, g.W,
                                                      g.cout, ex));
            else
            ASR_HIP(ctx, asr::launch_bn_bwd(st, tt.z[b], (b == 0 && fuse1) ? nullptr : dz, dA, tt.stats[b], pm(T, base + 2), pm(T, base + 1),
                                            tt.partial, tt.sums, pg(T, base + 1), pg(T, base + 2), B, g.H, g.W, g.cout,
                                            g.pool, 1, ex, tt.zsel[b]));
        }
        ctx->exch.phase = 0;
        if (phase == 1) return ASR_OK;
        if (tt.wstream) {
            ASR_HIP(ctx, hipEventRecord(tt.e_dz[cur], st));
            ASR_HIP(ctx, hipStreamWaitEvent(ws, tt.e_dz[cur], 0));
        }
        char name[32];
        snprintf(name, sizeof name, "train_wgrad_conv%d", b + 1);
        {
            ProfScope ps(ctx, name, view, 2.0 * rows * 9.0 * g.cin * g.cout, 4.0 * rows * (g.cin + g.cout), "", ws);
            if (b == 0) {
                if (fuse1)
                    ASR_HIP(ctx, asr::launch_conv1_wgrad(ws, tt.x[0], nullptr, B, g.H, g.W, g.cout, tt.partial, pg(T, base),
                                                         train_recompute1() ? nullptr : tt.z[0], dA, tt.stats[0],
                                                         pm(T, base + 2), pm(T, base + 1), tt.sums, ex ? ex->n_global : 0,
                                                         train_recompute1() ? tw.w_dev[0] : nullptr));
                else
                    ASR_HIP(ctx, asr::launch_conv1_wgrad(ws, tt.x[0], dz, B, g.H, g.W, g.cout, tt.partial, pg(T, base)));
            } else {
                ASR_HIP(ctx, asr::launch_wgrad(ws, tt.wplan[b], tt.x[b], dz, B, tt.wpartial, pg(T, base)));
            }
        }
        if (tt.wstream) {
            ASR_HIP(ctx, hipEventRecord(tt.e_wg[cur], ws));
            wg_pending[cur] = true;
        }
        if (b >= 1) {
            snprintf(name, sizeof name, "train_dgrad_conv%d", b + 1);
            ProfScope ps(ctx, name, view, 2.0 * rows * 9.0 * g.cin * g.cout, 4.0 * rows * (g.cin + g.cout),
                         tt.dplan[b].symbol);
            ASR_HIP(ctx, launch_conv_any(ctx, st, tt.dplan[b], dz, tt.wdgrad[b], nullptr, dB, B));
            std::swap(dA, dB);
        }
    }
    return ASR_OK;
}

int train_backward_towers(asr_ctx *ctx, int B, int64_t row_lo) {
    int rc;
    BwdState S[2];
    if (comm_active(ctx)) {               // exchanges of the two towers paired, block by block (train_forward_towers)
        for (int t = 0; t < 2; ++t)
            if ((rc = train_backward_tail(ctx, t, B, row_lo, 1, S[t])) != ASR_OK) return rc;
        if ((rc = train_pair_allreduce(ctx, 64)) != ASR_OK) return rc;
        for (int t = 0; t < 2; ++t)
            if ((rc = train_backward_tail(ctx, t, B, row_lo, 2, S[t])) != ASR_OK) return rc;
        for (int b = 7; b >= 0; --b) {
            for (int t = 0; t < 2; ++t)
                if ((rc = train_backward_block(ctx, t, B, b, 1, S[t])) != ASR_OK) return rc;
            if ((rc = train_pair_allreduce(ctx, 2 * ctx->tw[0].g[b].cout)) != ASR_OK) return rc;
            for (int t = 0; t < 2; ++t)
                if ((rc = train_backward_block(ctx, t, B, b, 2, S[t])) != ASR_OK) return rc;
        }
    } else {
        for (int t = 0; t < 2; ++t) {
            if ((rc = train_backward_tail(ctx, t, B, row_lo, 0, S[t])) != ASR_OK) return rc;
            for (int b = 7; b >= 0; --b)
                if ((rc = train_backward_block(ctx, t, B, b, 0, S[t])) != ASR_OK) return rc;
        }
    }
    for (int t = 0; t < 2; ++t) {
        TrainTower &tt = ctx->train->tw[t];
        hipStream_t st = train_stream(ctx, t);
        if (tt.wstream)                                          // the tower is done when its last weight gradients are
            for (int k = 0; k < 2; ++k)
                if (S[t].wg_pending[k]) ASR_HIP(ctx, hipStreamWaitEvent(st, tt.e_wg[k], 0));
        ASR_HIP(ctx, hipEventRecord(ctx->vdone[t], st));
        ctx->vpending[t] = true;
    }
    return ASR_OK;
}

int train_step_common(asr_ctx *ctx, const float *x1, const float *x2, int64_t B, float lr, float *loss, float *corr,
                      bool on_device, bool forward_only = false, float *lv1_out = nullptr, float *lv2_out = nullptr,
                      float *grads_out = nullptr) {
    if (!ctx) return ASR_ERR_INVALID;
    const auto t_begin = std::chrono::steady_clock::now();
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "train_step: call asr_train_begin first");
    if (!ctx->params_set) return fail(ctx, ASR_ERR_STATE, "train_step: asr_set_params has not been called");
    TrainState &T = *ctx->train;
    struct InTrain {
        asr_ctx *c;
        explicit InTrain(asr_ctx *cc) : c(cc) {
            c->in_train = true;
            for (int v = 0; v < 2; ++v) c->tstream[v] = train_stream(c, v);
        }
        ~InTrain() { c->in_train = false; }
    } in_train_guard(ctx);
    const bool dp = comm_active(ctx);
    if (B < (dp ? 1 : 2) || B > T.B)
        return fail(ctx, ASR_ERR_INVALID, "train_step: batch %lld outside [%d, %d]", (long long)B, dp ? 1 : 2, T.B);
    if (!x1 || !x2) return fail(ctx, ASR_ERR_INVALID, "train_step: NULL input");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int n = (int)B;
    const size_t b1 = (size_t)n * ctx->tw[0].in_h * ctx->tw[0].in_w * sizeof(float);
    const size_t b2 = (size_t)n * ctx->tw[1].in_h * ctx->tw[1].in_w * sizeof(float);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const int world = comm_world(ctx);
    if (world != T.world)
        return fail(ctx, ASR_ERR_STATE, "train_step: the communicator changed after asr_train_begin (world %d -> %d)",
                    T.world, world);
    // Rows of the whole batch and where this rank's rows sit in it.  Default: equal shards (batch * world).  After
    // asr_train_set_global_batch the shards follow the contiguous rule (the first n_global % world ranks hold one row
    // more), so that a batch that is not a multiple of the world size trains on ALL of its rows.
    int64_t n_global = (int64_t)n * world, row_lo = (int64_t)comm_rank(ctx) * n;
    int n_max = n;
    bool ragged = false;
    if (dp && T.global_batch > 0) {
        n_global = T.global_batch;
        const int64_t base = n_global / world, extra = n_global % world, r = comm_rank(ctx);
        row_lo = r * base + std::min<int64_t>(r, extra);
        n_max = (int)(base + (extra ? 1 : 0));
        ragged = extra != 0;
        if (n != base + (r < extra ? 1 : 0))
            return fail(ctx, ASR_ERR_INVALID, "train_step: rank %d of %d holds %lld rows of a batch of %lld, got %d",
                        (int)r, world, (long long)(base + (r < extra ? 1 : 0)), (long long)n_global, n);
        if (n_max > T.B || n_global < 2)
            return fail(ctx, ASR_ERR_INVALID, "train_step: global batch %lld does not fit the training state (%d rows per rank)",
                        (long long)n_global, T.B);
    }
    ctx->exch.n_local = n;
    ctx->exch.n_global = (int)n_global;
    for (int t = 0; t < 2; ++t)
        if (ctx->main_pending) ASR_HIP(ctx, hipStreamWaitEvent(train_stream(ctx, t), ctx->main_done, 0));
    ASR_HIP(ctx, hipMemcpyAsync(T.tw[0].x[0], x1, b1, kind, train_stream(ctx, 0)));
    ASR_HIP(ctx, hipMemcpyAsync(T.tw[1].x[0], x2, b2, kind, train_stream(ctx, 1)));
    // weight decay term of the reported loss: sum p^2 over the trainable parameters BEFORE the update
    // (train_dcca_pool.py:141-142).  A one-workgroup reduction (0.15 ms): it runs on the main stream while the towers
    // compute on theirs - the forward pass writes only running statistics, which the mask excludes
    if (!forward_only) ASR_HIP(ctx, asr::launch_l2_penalty(ctx->stream, T.pmaster, T.mask, T.poff[90], T.l2_dev));
    int rc;
    if ((rc = train_forward_towers(ctx, n)) != ASR_OK) return rc;
    if ((rc = join_views(ctx)) != ASR_OK) return rc;
    // data parallel: all-gather the tower outputs, every rank runs the CCALayer + loss on the FULL
    // batch (deterministic, cheap) and keeps its rows of dL/dH; rank r holds rows [r*n, (r+1)*n)
    const float *H1 = T.tw[0].H, *H2 = T.tw[1].H;
    float *lv1 = T.tw[0].lv, *lv2 = T.tw[1].lv, *dH1 = T.tw[0].dH, *dH2 = T.tw[1].dH;
    if (dp) {
        for (int t = 0; t < 2; ++t) {
            if (!ragged) {
                if ((rc = comm_allgather(ctx, ctx->stream, T.tw[t].H, T.Hg[t], (int64_t)n * 32 * sizeof(float))) != ASR_OK)
                    return rc;
                continue;
            }
            // shards of n_max or n_max - 1 rows: every rank sends n_max rows (the last one may be stale: never read),
            // then the valid rows of each slot are packed in rank order
            if ((rc = comm_allgather(ctx, ctx->stream, T.tw[t].H, T.Hpad[t], (int64_t)n_max * 32 * sizeof(float))) != ASR_OK)
                return rc;
            const int64_t base = n_global / world, extra = n_global % world;
            for (int r = 0; r < world; ++r) {
                const int64_t lo_r = r * base + std::min<int64_t>(r, extra), n_r = base + (r < extra ? 1 : 0);
                ASR_HIP(ctx, hipMemcpyAsync(T.Hg[t] + lo_r * 32, T.Hpad[t] + (size_t)r * n_max * 32, (size_t)n_r * 32 * sizeof(float),
                                            hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        H1 = T.Hg[0]; H2 = T.Hg[1]; lv1 = T.lvg[0]; lv2 = T.lvg[1]; dH1 = T.dHg[0]; dH2 = T.dHg[1];
    }
    {
        ProfScope ps(ctx, "train_cca_loss", 0, 0.0, 0.0);
        ASR_HIP(ctx, asr::launch_cca_train(ctx->stream, H1, H2, (int)n_global, pm(T, 90), pm(T, 90), ctx->cfg.r1,
                                           ctx->cfg.r2, ctx->cfg.rT, ctx->cfg.alpha, ctx->cfg.gamma, T.cca_ws,
                                           T.loss_dev, lv1, lv2, forward_only ? nullptr : dH1,
                                           forward_only ? nullptr : dH2));
    }
    if (dp) {                   // this rank's rows of the train-mode embeddings (debug tensor / burn-in output)
        const size_t off = (size_t)row_lo * 32, lb = (size_t)n * 32 * sizeof(float);
        ASR_HIP(ctx, hipMemcpyAsync(T.tw[0].lv, lv1 + off, lb, hipMemcpyDeviceToDevice, ctx->stream));
        ASR_HIP(ctx, hipMemcpyAsync(T.tw[1].lv, lv2 + off, lb, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (forward_only) {
        // burn-in (init_cca, utils/train_dcca_pool.py:160-162,170-182): only the default updates of the
        // train-mode graph happen - BN / CCALayer running values; no gradients, no Adam step
        if ((rc = train_repack(ctx)) != ASR_OK) return rc;
        T.master_dirty = true;
        const size_t lb = (size_t)n * 32 * sizeof(float);
        if (lv1_out) ASR_HIP(ctx, hipMemcpyAsync(lv1_out, T.tw[0].lv, lb, hipMemcpyDeviceToHost, ctx->stream));
        if (lv2_out) ASR_HIP(ctx, hipMemcpyAsync(lv2_out, T.tw[1].lv, lb, hipMemcpyDeviceToHost, ctx->stream));
        float host_loss[33];
        ASR_HIP(ctx, hipMemcpyAsync(host_loss, T.loss_dev, sizeof host_loss, hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (loss) *loss = host_loss[0];
        if (corr) memcpy(corr, host_loss + 1, 32 * sizeof(float));
        return mark_main(ctx);
    }
    ASR_HIP(ctx, hipEventRecord(T.cca_done, ctx->stream));
    if ((rc = train_backward_towers(ctx, n, row_lo)) != ASR_OK) return rc;
    if ((rc = join_views(ctx)) != ASR_OK) return rc;
    // data parallel: every rank holds the gradient of its rows' contribution to the full-batch loss - sum them
    if (dp && (rc = comm_allreduce(ctx, ctx->stream, T.pgrad, T.poff[90], ASR_DTYPE_F32)) != ASR_OK) return rc;
    if (grads_out) {
        // compute_gradients (train_dcca_pool.py:164): theano.grad of the train loss, no Adam step.  Like every
        // function compiled from the train-mode graph it still applies the graph's default updates (BatchNorm /
        // CCALayer running values), which the forward above already wrote into the master.
        const size_t nt = (size_t)T.poff[90];
        std::vector<float> pmh(nt);
        std::vector<unsigned char> mask(nt);
        ASR_HIP(ctx, hipMemcpyAsync(grads_out, T.pgrad, nt * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP(ctx, hipMemcpyAsync(pmh.data(), T.pmaster, nt * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP(ctx, hipMemcpyAsync(mask.data(), T.mask, nt, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = train_repack(ctx)) != ASR_OK) return rc;
        T.master_dirty = true;
        float host_loss[33];
        double host_l2 = 0.0;
        ASR_HIP(ctx, hipMemcpyAsync(host_loss, T.loss_dev, sizeof host_loss, hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP(ctx, hipMemcpyAsync(&host_l2, T.l2_dev, sizeof host_l2, hipMemcpyDeviceToHost, ctx->stream));
        ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < nt; ++i)       // the penalty's gradient, which the update path adds inside adam_kernel
            grads_out[i] = mask[i] ? grads_out[i] + 2.0f * ctx->cfg.l2 * pmh[i] : 0.0f;
        if (loss) *loss = host_loss[0] + ctx->cfg.l2 * (float)host_l2;
        if (corr) memcpy(corr, host_loss + 1, 32 * sizeof(float));
        return mark_main(ctx);
    }
    T.adam_t += 1;
    const double b1p = std::pow(0.9, (double)T.adam_t), b2p = std::pow(0.999, (double)T.adam_t);
    const float a_t = (float)((double)lr * std::sqrt(1.0 - b2p) / (1.0 - b1p));      // lasagne.updates.adam (A.7)
    {
        ProfScope ps(ctx, "train_adam", 0, 10.0 * T.poff[90], 28.0 * T.poff[90]);
        ASR_HIP(ctx, asr::launch_adam(ctx->stream, T.pmaster, T.pgrad, T.adam_m, T.adam_v, T.mask, T.poff[90], a_t,
                                      0.9, 0.999, 1e-8f, ctx->cfg.l2));
    }
    if ((rc = train_repack(ctx)) != ASR_OK) return rc;
    T.master_dirty = true;
    float host_loss[33];
    double host_l2 = 0.0;
    static const bool timing = getenv("ASR_TRAIN_HOST_TIMING") != nullptr;       // host time of the enqueue vs the whole step
    const auto t_enq = std::chrono::steady_clock::now();      // (the copies into pageable memory below wait for the stream)
    ASR_HIP(ctx, hipMemcpyAsync(host_loss, T.loss_dev, sizeof host_loss, hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP(ctx, hipMemcpyAsync(&host_l2, T.l2_dev, sizeof host_l2, hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (timing) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[asr] train_step B=%d: enqueue %.3f ms, then waited %.3f ms\n", n,
                std::chrono::duration<double, std::milli>(t_enq - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_enq).count());
    }
    if (loss) *loss = host_loss[0] + ctx->cfg.l2 * (float)host_l2;
    if (corr) memcpy(corr, host_loss + 1, 32 * sizeof(float));
    return mark_main(ctx);
}

}  // namespace

extern "C" {

int asr_debug_tune_report(asr_ctx *ctx, int32_t *checked, int32_t *mismatches, float *max_diff) {
    if (!ctx) return ASR_ERR_INVALID;
    if (checked) *checked = ctx->tune_checked;
    if (mismatches) *mismatches = ctx->tune_bad;
    if (max_diff) *max_diff = ctx->tune_max_diff;
    return ASR_OK;
}

// librccl is resolved at run time (single-GPU users never load it).  Order: ASR_RCCL_LIB (a path), the ROCm
// installation's own copy ($ROCM_PATH/lib, /opt/rocm/lib) - so that the same library runs whatever else the process has
// loaded (a bare dlopen("librccl.so") binds e.g. the copy inside a PyTorch wheel once torch is imported) - then the
// loader's search path.  asr_comm_library() reports what was bound.
static void *open_rccl() {
    if (const char *p = getenv("ASR_RCCL_LIB"))
        if (void *dl = dlopen(p, RTLD_NOW | RTLD_GLOBAL)) return dl;
    std::vector<std::string> cands;
    if (const char *r = getenv("ROCM_PATH")) cands.push_back(std::string(r) + "/lib/librccl.so");
    cands.push_back("/opt/rocm/lib/librccl.so");
    cands.push_back("librccl.so");
    cands.push_back("librccl.so.1");
    for (auto &c : cands)
        if (void *dl = dlopen(c.c_str(), RTLD_NOW | RTLD_GLOBAL)) return dl;
    return nullptr;
}

int asr_comm_library(asr_ctx *ctx, char *path, int cap) {
    if (!ctx || !path || cap < 1) return ASR_ERR_INVALID;
    path[0] = 0;
    if (!ctx->comm || !ctx->comm->dl || !ctx->comm->pAllGather) return ASR_OK;      // no RCCL communicator: ""
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(ctx->comm->pAllGather), &info) && info.dli_fname)
        snprintf(path, (size_t)cap, "%s", info.dli_fname);
    return ASR_OK;
}

int asr_comm_unique_id(void *id_out) {
    if (!id_out) return ASR_ERR_INVALID;
    void *dl = open_rccl();
    if (!dl) return fail(nullptr, ASR_ERR_STATE, "comm: cannot load librccl.so: %s", dlerror());
    auto get = reinterpret_cast<ncclResult_t (*)(ncclUniqueId *)>(dlsym(dl, "ncclGetUniqueId"));
    if (!get) return fail(nullptr, ASR_ERR_STATE, "comm: ncclGetUniqueId not found");
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) return fail(nullptr, ASR_ERR_HIP, "comm: ncclGetUniqueId failed");
    memcpy(id_out, &id, ASR_COMM_ID_BYTES);
    return ASR_OK;                      // the handle stays open: asr_comm_init re-uses the loaded library
}

int asr_comm_init(asr_ctx *ctx, int rank, int world, const void *unique_id) {
    if (!ctx || !unique_id) return ASR_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, ASR_ERR_INVALID, "comm: rank %d of %d", rank, world);
    if (ctx->train) return fail(ctx, ASR_ERR_STATE, "comm: initialise the communicator before asr_train_begin");
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    free_comm(ctx);
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world;
    c->dl = open_rccl();
    if (!c->dl) return fail(ctx, ASR_ERR_STATE, "comm: cannot load librccl.so: %s", dlerror());
    auto init = reinterpret_cast<ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>(dlsym(c->dl, "ncclCommInitRank"));
    c->pAllReduce = reinterpret_cast<decltype(c->pAllReduce)>(dlsym(c->dl, "ncclAllReduce"));
    c->pAllGather = reinterpret_cast<decltype(c->pAllGather)>(dlsym(c->dl, "ncclAllGather"));
    c->pCommDestroy = reinterpret_cast<decltype(c->pCommDestroy)>(dlsym(c->dl, "ncclCommDestroy"));
    c->pGetErrorString = reinterpret_cast<decltype(c->pGetErrorString)>(dlsym(c->dl, "ncclGetErrorString"));
    if (!init || !c->pAllReduce || !c->pAllGather || !c->pCommDestroy || !c->pGetErrorString) {
        dlclose(c->dl);
        return fail(ctx, ASR_ERR_STATE, "comm: librccl.so lacks a required symbol");
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, ASR_COMM_ID_BYTES);
    const ncclResult_t r = init(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        const char *msg = c->pGetErrorString(r);
        dlclose(c->dl);
        return fail(ctx, ASR_ERR_HIP, "comm: ncclCommInitRank: %s", msg);
    }
    install_comm(ctx, std::move(c));
    return ASR_OK;
}

int asr_comm_init_custom(asr_ctx *ctx, int rank, int world, asr_allreduce_fn allreduce, asr_allgather_fn allgather,
                         void *user) {
    if (!ctx || !allreduce || !allgather) return ASR_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, ASR_ERR_INVALID, "comm: rank %d of %d", rank, world);
    if (ctx->train) return fail(ctx, ASR_ERR_STATE, "comm: initialise the communicator before asr_train_begin");
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    free_comm(ctx);
    std::unique_ptr<Comm> c(new Comm());
    c->rank = rank; c->world = world; c->ar = allreduce; c->ag = allgather; c->user = user;
    install_comm(ctx, std::move(c));
    return ASR_OK;
}

int asr_comm_destroy(asr_ctx *ctx) {
    if (!ctx) return ASR_ERR_INVALID;
    if (ctx->train) return fail(ctx, ASR_ERR_STATE, "comm: call asr_train_end first");
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    free_comm(ctx);
    return ASR_OK;
}

int asr_comm_stats(asr_ctx *ctx, int64_t *counts, int reset) {
    if (!ctx || !counts) return ASR_ERR_INVALID;
    Comm *c = ctx->comm.get();
    counts[0] = c ? c->n_allreduce : 0;
    counts[1] = c ? c->b_allreduce : 0;
    counts[2] = c ? c->n_allgather : 0;
    counts[3] = c ? c->b_allgather : 0;
    if (c && reset) c->n_allreduce = c->b_allreduce = c->n_allgather = c->b_allgather = 0;
    return ASR_OK;
}

int asr_comm_info(asr_ctx *ctx, int *rank, int *world) {
    if (!ctx) return ASR_ERR_INVALID;
    if (rank) *rank = comm_rank(ctx);
    if (world) *world = comm_world(ctx);
    return ASR_OK;
}

int asr_comm_allreduce_dev(asr_ctx *ctx, void *buf_dev, int64_t count, int dtype) {
    if (!ctx) return ASR_ERR_INVALID;
    if (count < 0 || (count > 0 && !buf_dev) || (dtype != ASR_DTYPE_F32 && dtype != ASR_DTYPE_F64 && dtype != ASR_DTYPE_I32))
        return fail(ctx, ASR_ERR_INVALID, "comm_allreduce: bad argument (count %lld, dtype %d)", (long long)count, dtype);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    if ((rc = comm_allreduce(ctx, ctx->stream, buf_dev, count, dtype)) != ASR_OK) return rc;
    return mark_main(ctx);
}

int asr_comm_allgather_dev(asr_ctx *ctx, const void *send_dev, void *recv_dev, int64_t bytes_per_rank) {
    if (!ctx) return ASR_ERR_INVALID;
    if (bytes_per_rank < 0 || (bytes_per_rank > 0 && (!send_dev || !recv_dev)))
        return fail(ctx, ASR_ERR_INVALID, "comm_allgather: bad argument");
    if (bytes_per_rank == 0) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    if ((rc = comm_allgather(ctx, ctx->stream, send_dev, recv_dev, bytes_per_rank)) != ASR_OK) return rc;
    return mark_main(ctx);
}

/* pairs sharded by contiguous ranges: rank r holds queries / candidates [r*n_local, (r+1)*n_local) */
int asr_rank_sharded_dev(asr_ctx *ctx, const float *lv1_dev, const float *lv2_dev, int64_t n_local, float *lv2_all_dev,
                         int32_t *ranks, double *dstar, int32_t *ties) {
    if (!ctx || !lv1_dev || !lv2_dev || !lv2_all_dev) return ASR_ERR_INVALID;
    if (n_local < 0) return fail(ctx, ASR_ERR_INVALID, "rank_sharded: n_local %lld", (long long)n_local);
    if (n_local == 0) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = join_views(ctx);
    if (rc != ASR_OK) return rc;
    const int world = comm_world(ctx), rank = comm_rank(ctx);
    {
        ProfScope ps(ctx, "allgather_candidates", 0, 0.0, 128.0 * (double)n_local * world);
        rc = comm_allgather(ctx, ctx->stream, lv2_dev, lv2_all_dev, n_local * 32 * (int64_t)sizeof(float));
        if (rc != ASR_OK) return rc;
    }
    return asr_rank_dev(ctx, lv1_dev, n_local, 32, lv2_all_dev, n_local * world, 32, 32, (int64_t)rank * n_local,
                        n_local * world, ranks, dstar, ties);
}

int asr_train_begin(asr_ctx *ctx, int batch_size) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->params_set) return fail(ctx, ASR_ERR_STATE, "train_begin: asr_set_params has not been called");
    if (batch_size < (comm_active(ctx) ? 1 : 2) || batch_size > 8192)
        return fail(ctx, ASR_ERR_INVALID, "train_begin: batch size %d", batch_size);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    if (ctx->train && ctx->train->master_dirty && (rc = train_download_master(ctx)) != ASR_OK) return rc;
    rc = train_alloc(ctx, batch_size);
    if (rc != ASR_OK) free_train(ctx);
    return rc;
}

int asr_train_set_global_batch(asr_ctx *ctx, int64_t n_global) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "train_set_global_batch: call asr_train_begin first");
    if (n_global < 0 || (n_global > 0 && n_global < std::max(2, comm_world(ctx))))
        return fail(ctx, ASR_ERR_INVALID, "train_set_global_batch: %lld rows for %d ranks (every rank needs a row, the "
                    "batch two)", (long long)n_global, comm_world(ctx));
    ctx->train->global_batch = n_global;
    return ASR_OK;
}

int asr_train_end(asr_ctx *ctx) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!ctx->train) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    if (ctx->train->master_dirty && (rc = train_download_master(ctx)) != ASR_OK) return rc;
    // the Winograd-domain copies the embedding kernels read are derived data: rebuild them from the final master
    if ((rc = refresh_wino_weights(ctx)) != ASR_OK) return rc;
    free_train(ctx);
    return ASR_OK;
}

int asr_train_step(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float lr, float *loss, float *corr) {
    return train_step_common(ctx, x1, x2, batch, lr, loss, corr, false);
}
int asr_train_step_dev(asr_ctx *ctx, const float *x1_dev, const float *x2_dev, int64_t batch, float lr, float *loss,
                       float *corr) {
    return train_step_common(ctx, x1_dev, x2_dev, batch, lr, loss, corr, true);
}

int asr_burn_in(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float *lv1, float *lv2) {
    return train_step_common(ctx, x1, x2, batch, 0.0f, nullptr, nullptr, false, true, lv1, lv2);
}

int asr_compute_gradients(asr_ctx *ctx, const float *x1, const float *x2, int64_t batch, float *grads, int64_t n,
                          float *loss) {
    if (!ctx || !grads) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "compute_gradients: call asr_train_begin first");
    if (n != ctx->train->poff[90])
        return fail(ctx, ASR_ERR_INVALID, "compute_gradients: expected %lld values", (long long)ctx->train->poff[90]);
    return train_step_common(ctx, x1, x2, batch, 0.0f, loss, nullptr, false, false, nullptr, nullptr, grads);
}

int asr_valid_loss(asr_ctx *ctx, const float *x1, const float *x2, int64_t n, float *loss) {
    if (!ctx || !loss) return ASR_ERR_INVALID;
    if (n < 2) return fail(ctx, ASR_ERR_INVALID, "valid_loss: needs at least 2 pairs");
    std::vector<float> lv1((size_t)n * 32), lv2((size_t)n * 32);
    int rc = asr_embed_view1(ctx, x1, ASR_IN_F32_PREPARED, n, ASR_OUT_LATENT, lv1.data());
    if (rc != ASR_OK) return rc;
    rc = asr_embed_view2(ctx, x2, n, ASR_OUT_LATENT, lv2.data());
    if (rc != ASR_OK) return rc;
    float *d = nullptr;
    ASR_HIP(ctx, hipMalloc((void **)&d, ((size_t)n * 64 + 1) * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, lv1.data(), (size_t)n * 32 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + (size_t)n * 32, lv2.data(), (size_t)n * 32 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = asr::launch_rank_loss(ctx->stream, d, d + (size_t)n * 32, (int)n, ctx->cfg.gamma, d + (size_t)n * 64);
    if (e == hipSuccess) e = hipMemcpyAsync(loss, d + (size_t)n * 64, sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "valid_loss: %s", hipGetErrorString(e));
    return ASR_OK;
}

int asr_opt_state_size(asr_ctx *ctx, int64_t *n) {
    if (!ctx || !n) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "opt_state: call asr_train_begin first");
    *n = ctx->train->poff[90];
    return ASR_OK;
}

int asr_get_opt_state(asr_ctx *ctx, float *m, float *v, int64_t n, int32_t *t) {
    if (!ctx || !m || !v || !t) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "opt_state: call asr_train_begin first");
    TrainState &T = *ctx->train;
    if (n != T.poff[90]) return fail(ctx, ASR_ERR_INVALID, "opt_state: expected %lld values", (long long)T.poff[90]);
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, hipMemcpy(m, T.adam_m, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    ASR_HIP(ctx, hipMemcpy(v, T.adam_v, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    *t = T.adam_t;
    return ASR_OK;
}

int asr_set_opt_state(asr_ctx *ctx, const float *m, const float *v, int64_t n, int32_t t) {
    if (!ctx || !m || !v || t < 0) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "opt_state: call asr_train_begin first");
    TrainState &T = *ctx->train;
    if (n != T.poff[90]) return fail(ctx, ASR_ERR_INVALID, "opt_state: expected %lld values", (long long)T.poff[90]);
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, hipMemcpy(T.adam_m, m, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    ASR_HIP(ctx, hipMemcpy(T.adam_v, v, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    T.adam_t = t;
    return ASR_OK;
}

// kind: 0 z (raw conv out), 1 x (block input), 2 stats [mu|inv_std], 3 H, 4 dH, 5 lv (train-mode output),
//       6 grad of parameter `index`, 7 master value of parameter `index`, 8 [loss | corr(32)]
int asr_debug_train_tensor(asr_ctx *ctx, int kind, int view, int index, int64_t batch, float *out, int64_t cap,
                           int64_t *n_out) {
    if (!ctx || !n_out) return ASR_ERR_INVALID;
    if (!ctx->train) return fail(ctx, ASR_ERR_STATE, "debug_train_tensor: no training state");
    TrainState &T = *ctx->train;
    const float *src = nullptr;
    int64_t n = 0;
    if (kind <= 5 && (view < 1 || view > 2)) return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: view");
    if (kind <= 2 && (index < 0 || index > 8)) return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: block");
    if (kind >= 6 && kind <= 7 && (index < 0 || index >= (int)ctx->params.size()))
        return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: parameter index");
    const LayerGeom *g = (kind <= 2) ? &ctx->tw[view - 1].g[index] : nullptr;
    switch (kind) {
        case 0:
            if (index == 0 && train_recompute1())
                return fail(ctx, ASR_ERR_STATE, "debug_train_tensor: block 1's raw output is not materialised by the training "
                                                "step (ASR_TRAIN_RECOMPUTE1=0 keeps it)");
            src = T.tw[view - 1].z[index]; n = batch * g->H * g->W * g->cout; break;
        case 1: src = T.tw[view - 1].x[index]; n = batch * g->H * g->W * g->cin; break;
        case 2: src = T.tw[view - 1].stats[index]; n = 2 * g->cout; break;
        case 3: src = T.tw[view - 1].H; n = batch * 32; break;
        case 4: src = T.tw[view - 1].dH; n = batch * 32; break;
        case 5: src = T.tw[view - 1].lv; n = batch * 32; break;
        case 6: src = pg(T, index); n = (int64_t)ctx->params[index].size(); break;
        case 7: src = pm(T, index); n = (int64_t)ctx->params[index].size(); break;
        case 8: src = T.loss_dev; n = 33; break;
        case 9: {               // pooled blocks: the raw value of every pooling window's selected element (N, H/2, W/2, C)
            if (view < 1 || view > 2 || index < 0 || index > 7) return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: zsel block");
            const LayerGeom &gz = ctx->tw[view - 1].g[index];
            src = T.tw[view - 1].zsel[index];
            if (!src) return fail(ctx, ASR_ERR_STATE, "debug_train_tensor: block %d keeps no selected elements (not pooled, or "
                                  "ASR_TRAIN_ZSEL=0)", index + 1);
            n = batch * (gz.H / 2) * (gz.W / 2) * gz.cout;
            break;
        }
        default: return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: kind %d", kind);
    }
    *n_out = n;
    if (!out) return ASR_OK;
    if (cap < n) return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: buffer too small (%lld < %lld)", (long long)cap, (long long)n);
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    ASR_HIP(ctx, hipMemcpy(out, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return ASR_OK;
}

// CCALayer train branch + length norm + ranking loss alone (unit-test aid and building block):
// H1, H2 (B,32) host; cca_in 5184 floats (U V mean1 mean2 S12 S11 S22); outputs may be NULL.
int asr_cca_train_debug(asr_ctx *ctx, const float *H1, const float *H2, int64_t B, const float *cca_in, float *cca_out,
                        float *loss_corr, float *lv1, float *lv2, float *dH1, float *dH2) {
    if (!ctx || !H1 || !H2 || !cca_in || B < 2) return ASR_ERR_INVALID;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t hb = (size_t)B * 32;
    float *d = nullptr;
    void *ws = nullptr;
    auto cleanup = [&]() { (void)hipFree(d); (void)hipFree(ws); };
    hipError_t e = hipMalloc((void **)&d, (6 * hb + 2 * 5184 + 64) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&ws, asr::cca_train_ws_bytes((int)B));
    if (e == hipSuccess) e = hipMemsetAsync(ws, 0, asr::cca_train_ws_bytes((int)B), ctx->stream);
    float *dH1d = d + 2 * hb, *dH2d = d + 3 * hb, *lv1d = d + 4 * hb, *lv2d = d + 5 * hb;
    float *cin = d + 6 * hb, *cout = cin + 5184, *lossd = cout + 5184;
    if (e == hipSuccess) e = hipMemcpyAsync(d, H1, hb * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + hb, H2, hb * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(cin, cca_in, 5184 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = asr::launch_cca_train(ctx->stream, d, d + hb, (int)B, cin, cout, ctx->cfg.r1, ctx->cfg.r2, ctx->cfg.rT,
                                  ctx->cfg.alpha, ctx->cfg.gamma, ws, lossd, lv1d, lv2d, dH1 ? dH1d : nullptr,
                                  dH1 ? dH2d : nullptr);
    auto dl = [&](float *dst, const float *src, size_t n) {
        if (dst && e == hipSuccess) e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    };
    dl(cca_out, cout, 5184); dl(loss_corr, lossd, 33); dl(lv1, lv1d, hb); dl(lv2, lv2d, hb);
    if (dH1) { dl(dH1, dH1d, hb); dl(dH2, dH2d, hb); }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "cca_train_debug: %s", hipGetErrorString(e));
    return ASR_OK;
}

}  // extern "C"
