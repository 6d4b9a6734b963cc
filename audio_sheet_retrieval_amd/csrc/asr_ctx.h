// Internal header of the C-ABI layer: the context structure and the helpers its translation units share
//   asr_api.hip            context, towers, run-time tuner, embedding, CCA fit, device allocations, profile
//   asr_api_train.hip      training step (forward, backward, Adam), collectives
//   asr_api_retrieval.hip  ranking, top-k, the resident code data base, shard entry points, alignment, piece vote,
//                          audio front-end
// Not part of the boundary - that is include/asr_hip.h.
#pragma once
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

namespace asr_detail {


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
    uint8_t *ztie[9] = {};          // with zsel under ASR_POOL_TIES_ALL: how many window elements share the maximum (2 bits/channel)
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
    // the convolutions' own BatchNorm-statistics table: all-zero between uses (its consumer, colsum_final_kernel, clears
    // the rows it read), so a convolution that leaves rows of idle waves untouched needs no memset in front of it
    double *fstats = nullptr;
    size_t fstats_doubles = 0;      // rows x columns of that table; the staged rows of its reduction follow
    unsigned *ticket = nullptr;     // last-arriver counter of the fused reductions (zero between launches)
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
    // block gates (single-GPU step, towers on their own streams): "the spectrogram tower has finished block b" of the
    // forward [0..8] and of the backward [9 + b, b = 8 (tail) .. 0] pass; the sheet tower's stream waits on them so that
    // the small tower's chain of short kernels runs BESIDE the sheet tower's early blocks instead of behind them
    hipEvent_t gate[18] = {};
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
    // asr_comm_timing: every collective bracketed by two HIP events on the stream it is enqueued on (RCCL) or timed on the
    // host clock (callbacks, which are host-synchronous); read and reset by asr_comm_timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    double host_ms = 0.0;
    int64_t timed_calls = 0;
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


}  // namespace asr_detail

using namespace asr_detail;

struct asr_ctx {
    asr_config cfg{};
    float loss_weight = 1.0f;       // asr_set_objective: get_contrastive_cos_loss(weight, gamma, symmetric)
    int loss_symmetric = 0;
    int num_cus = 256;
    hipStream_t stream = nullptr;             // main stream: ranking, CCA fit, copies
    hipStream_t vstream[2] = {nullptr, nullptr};   // one per tower: the training step overlaps the two towers
    // The side stream of the sheet tower's weight gradients (training step), created WITH the context, right behind the
    // three streams above: the runtime maps streams onto four hardware queues by creation (least-used queue first), so
    // these four take one queue each.  Created in asr_train_begin it came after whatever copy streams the host-buffer
    // entry points had opened in between and could land on a tower's queue - the update of an engine that had run
    // asr_eval_batches before took 10.0 ms instead of 9.65 (round 6, bench.py's secondary leg against a fresh engine).
    hipStream_t wside_stream = nullptr;
    // The copy streams of the two host-buffer pipelines ([0], [1]: asr_eval_batches' H2D / D2H; [2]: asr_embed_*'s H2D) and
    // a spare, created with the context as well, in this order.  Which hardware queue a copy stream shares matters: moved
    // three places along the runtime's least-used order (ASR_COPY_STEER probe, round 6) the page-locked pipeline ran at
    // 221-226 k pairs/s instead of 281-284 k - a copy stream on the queue the towers compute on.  Created lazily, their place
    // depended on what else the process had opened in between.  With 4 + 4 streams per context every context of a process
    // starts at the same queue alignment and lands in the layout that was measured (ASR_EARLY_COPY=0: lazily, as in
    // rounds 2-5).
    hipStream_t copy_streams[4] = {nullptr, nullptr, nullptr, nullptr};
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
    unsigned *topk_tickets = nullptr;         // 1024 last-arriver counters of the top-k call (zero between calls)
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

namespace asr_detail {

// the last error of a context (or, without one, of the calling thread), returned code passed through
int fail(asr_ctx *ctx, int code, const char *fmt, ...);

#define ASR_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, ASR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

ProfRec *prof_rec(asr_ctx *ctx, const std::string &name, double flops, double bytes);
void prof_fold(ProfRec *r);

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

// stream bookkeeping: wait for the towers / mark the main stream busy / drain everything
int join_views(asr_ctx *ctx);
int mark_main(asr_ctx *ctx);
int sync_all(asr_ctx *ctx);
// float64 row-norm scratch of the two sides of a ranking / top-k call
int ensure_norms(asr_ctx *ctx, int64_t n1, int64_t n2);
int rank_check(asr_ctx *ctx, int64_t n1, int64_t ld1, int64_t n2, int64_t ld2, int dim, int64_t query_offset,
               int64_t n1_global);
// towers (asr_api.hip) <-> training step and collectives (asr_api_train.hip)
hipError_t launch_conv_any(asr_ctx *ctx, hipStream_t st, const asr::ConvPlan &p, const float *in, const float *w,
                           const float *bn, float *out, int n, const asr::Fuse1Args *f1 = nullptr, double *stats = nullptr,
                           int *stats_rows = nullptr, bool stats_clean = false, const asr::BnBwdFuse *bf = nullptr);
int tune_cache_tag();
void free_train(asr_ctx *ctx);
void free_comm(asr_ctx *ctx);
int refresh_wino_weights(asr_ctx *ctx);
int train_upload_master(asr_ctx *ctx);        // host mirror of the parameters -> device master (+ derived layouts)
int train_download_master(asr_ctx *ctx);      // device master -> host mirror

}  // namespace asr_detail
