// C-ABI layer, training side: the fused training step (train-mode forward, CCALayer + loss, backward, Adam), its
// schedule tuner and weight re-layouts, the RCCL / host-callback communicator and the data-parallel exchange
// (include/asr_hip.h for the contract and the reference interfaces each entry point replaces).  Context and shared
// helpers: asr_ctx.h.
#include "asr_ctx.h"
// ===========================================================================
// training step (utils/train_dcca_pool.py:85-167 compiled `train` / `valid`)
// ===========================================================================
namespace asr_detail {

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
    // (read at every asr_train_begin, not latched: a caller that switches the tuner off for one engine gets that)
    const bool on = !(getenv("ASR_AUTOTUNE") && getenv("ASR_AUTOTUNE")[0] == '0') &&
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
                // F(4x4) FORWARD builds only under ASR_POOL_TIES_FIRST.  The "every tied element" rule compares activations
                // for equality, and what makes the device's ties the reference's (Theano's CorrMM gives equal patches equal
                // outputs) is that an F(2x2) output depends on its own 3x3 patch only: a uniform or axis-constant patch takes
                // the exact path through the one non-zero transform position whatever else its tile holds.  An F(4x4) output
                // also carries rounding noise from the two tile columns / rows beyond its patch - measured on pages with
                // large white areas: 11 % / 4 % of the pooling windows of blocks 6 / 8 lost ties float64 has (1e-6 with
                // F(2x2)).  Data gradients are not compared: they keep their F(4x4) builds.
                if (dir == 1 || ctx->cfg.pool_ties == ASR_POOL_TIES_FIRST)
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

}  // namespace asr_detail

namespace asr_detail {
int train_upload_master(asr_ctx *ctx) {
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

int train_download_master(asr_ctx *ctx) {
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
}  // namespace asr_detail

namespace asr_detail {

// ---- collectives ------------------------------------------------------------------------------------
// RCCL calls are enqueued on the stream; a host callback is host-synchronous (the stream is drained first and the
// callback returns with the result in place).
// asr_comm_timing: an event pair around one enqueued collective (host clock for callback transports).  At most 8192
// pairs are kept between two reads: a caller that switches the timing on and never reads it must not grow without bound.
struct CommTimer {
    Comm *c; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    bool host = false;
    std::chrono::steady_clock::time_point t0;
    CommTimer(Comm *cc, hipStream_t s, bool on_host) : c(cc && cc->timing ? cc : nullptr), st(s), host(on_host) {
        if (!c) return;
        if (host) { c->timed_calls += 1; t0 = std::chrono::steady_clock::now(); return; }
        if (c->timed.size() >= 8192 || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
            hipEventRecord(e0, st) != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            e0 = e1 = nullptr;
            c = nullptr;                                   // this collective is not timed (and not counted)
            return;
        }
        c->timed_calls += 1;
    }
    ~CommTimer() {
        if (!c) return;
        if (host) {
            c->host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        } else {
            (void)hipEventRecord(e1, st);
            c->timed.emplace_back(e0, e1);
        }
    }
};

int comm_allreduce(asr_ctx *ctx, hipStream_t st, void *buf, int64_t count, int dtype) {
    Comm *c = ctx->comm.get();
    if (!c || (c->world <= 1 && !c->force) || count <= 0) return ASR_OK;
    if (dtype != ASR_DTYPE_F32 && dtype != ASR_DTYPE_F64 && dtype != ASR_DTYPE_I32)
        return fail(ctx, ASR_ERR_INVALID, "comm: all-reduce dtype %d", dtype);
    c->n_allreduce += 1;
    c->b_allreduce += count * (dtype == ASR_DTYPE_F64 ? 8 : 4);
    if (c->ar) {
        ASR_HIP(ctx, hipStreamSynchronize(st));
        CommTimer tm(c, st, true);
        if (c->ar(c->user, buf, count, dtype) != 0) return fail(ctx, ASR_ERR_STATE, "comm: all-reduce callback failed");
        return ASR_OK;
    }
    CommTimer tm(c, st, false);
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
        CommTimer tm(c, st, true);
        if (c->ag(c->user, send, recv, bytes_per_rank) != 0)
            return fail(ctx, ASR_ERR_STATE, "comm: all-gather callback failed");
        return ASR_OK;
    }
    CommTimer tm(c, st, false);
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
            // [mu | inv_std] of the batch + (pooled blocks) [gamma * inv_std | beta] as the forward apply pass used them
            ASR_HIP(ctx, hipMalloc((void **)&tt.stats[b], (size_t)4 * g.cout * sizeof(float)));
            // pooled blocks: the raw value of every pooling window's selected element, written by the forward apply pass
            // for the reduce pass of the BatchNorm backward (ASR_TRAIN_ZSEL=0: that pass re-reads the four window elements)
            static const bool use_zsel = !(getenv("ASR_TRAIN_ZSEL") && getenv("ASR_TRAIN_ZSEL")[0] == '0');
            if (b < 8 && g.pool && use_zsel) {
                ASR_HIP(ctx, hipMalloc((void **)&tt.zsel[b], (size_t)B * (g.H / 2) * (g.W / 2) * g.cout * sizeof(float)));
                // "every tied element" pooling gradient: the multiplicity of each window's maximum, two bits per channel
                if (ctx->cfg.pool_ties == ASR_POOL_TIES_ALL)
                    ASR_HIP(ctx, hipMalloc((void **)&tt.ztie[b], (size_t)B * (g.H / 2) * (g.W / 2) * (g.cout / 4)));
            }
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
            // Only the sheet tower gets a side stream.  The runtime maps streams onto FOUR hardware queues (main, the two
            // towers', one more): a second side stream shared the sheet tower's queue, the spectrogram tower's weight
            // gradients queued up behind ~4 ms of the sheet tower's, its main stream waited for them (dz buffer re-use)
            // and its backward pass finished 1 ms AFTER everything else, alone on the GPU (profiles/r05_train_timeline.txt).
            // Its weight gradients are 0.5 ms in total: they run on its own stream now.
            static const bool wside2 = getenv("ASR_TRAIN_WGRAD_STREAM2") && getenv("ASR_TRAIN_WGRAD_STREAM2")[0] == '1';
            if (wside && !one && !comm_active(ctx) && (t == 0 || wside2)) {
                ASR_HIP(ctx, hipMalloc((void **)&tt.dz2, max_z * sizeof(float)));
                if (t == 0 && ctx->wside_stream) tt.wstream = ctx->wside_stream;       // created with the context (asr_ctx.h)
                else ASR_HIP(ctx, hipStreamCreateWithFlags(&tt.wstream, hipStreamNonBlocking));
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
        {
            // the convolutions' statistics table (zero between uses) + the staged rows of its reduction behind it
            const size_t frows = (size_t)std::max(std::max(asr::conv_wino_stats_rows_max(ctx->num_cus),
                                                           asr::conv_wino4_stats_rows_max(ctx->num_cus)), 4096);
            tt.fstats_doubles = frows * 2 * 128;
            const size_t fbytes = (tt.fstats_doubles + (frows / 32 + 2) * 256) * sizeof(double);
            ASR_HIP(ctx, hipMalloc((void **)&tt.fstats, fbytes));
            ASR_HIP(ctx, hipMemsetAsync(tt.fstats, 0, fbytes, ctx->stream));
            ASR_HIP(ctx, hipMalloc((void **)&tt.ticket, 4 * sizeof(unsigned)));
            ASR_HIP(ctx, hipMemsetAsync(tt.ticket, 0, 4 * sizeof(unsigned), ctx->stream));
        }
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
        static const bool fused_reduce = (getenv("ASR_TRAIN_FUSED_REDUCE") && getenv("ASR_TRAIN_FUSED_REDUCE")[0] == '1');
        const bool own_table = fused_reduce && !ex && 2 * g.cout <= 256;
        double *stab = own_table ? tt.fstats : tt.partial;
        int srows = 0;
        if (phase != 2) {
            // algorithmic bytes: input read, raw output written - block 1 in the recompute form stores nothing (its
            // statistics pass only reads the image)
            ProfScope ps(ctx, name, view, 2.0 * rows * g.k * g.k * g.cin * g.cout,
                         4.0 * rows * (g.cin + ((b == 0 && train_recompute1()) ? 0 : g.cout)),
                         b >= 1 && b < 8 ? tt.fplan[b].symbol : "");
            // ASR_TRAIN_FUSED_REDUCE=1 (single GPU): the convolutions write their statistics rows into a table that is
            // all-zero between uses (no memset in front of them) and ONE last-arriver launch reduces it (colsum_final_kernel):
            // 229 -> 184 kernels per update - and 10.62 ms against 10.50 (bench harness; 10.57 / 10.53 stand-alone): the
            // agent-scope release / acquire around the ticket costs what the launches saved.  Default: round 4's memset /
            // stage / final launches, here and in the BatchNorm backward.
            if (b == 0)
                ASR_HIP(ctx, asr::launch_conv1_raw(st, tt.x[0], tw.w_dev[0], tt.z[0], B, g.H, g.W, g.cout,
                                                   fuse_stats ? stab : nullptr, &srows, train_recompute1() ? 1 : 0));
            else if (b < 8)
                ASR_HIP(ctx, launch_conv_any(ctx, st, tt.fplan[b], tt.x[b], tw.w_dev[b], nullptr, tt.z[b], B, nullptr,
                                             fuse_stats ? stab : nullptr, &srows, own_table));
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
            ASR_HIP(ctx, asr::launch_bn_stats_final(st, stab, std::max(srows, 1), rows, g.cout, tt.stats[b], pm(T, base + 3),
                                                    pm(T, base + 4), 1e-4f, 0.1f, ex, tt.sums, own_table ? tt.ticket : nullptr,
                                                    own_table, own_table ? tt.fstats + tt.fstats_doubles : nullptr));
        else
            ASR_HIP(ctx, asr::launch_bn_stats(st, tt.z[b], rows, g.cout, tt.partial, tt.stats[b], pm(T, base + 3),
                                              pm(T, base + 4), 1e-4f, 0.1f, ex, tt.sums, ex ? nullptr : tt.ticket));
        if (phase == 1) { ctx->exch.phase = 0; return ASR_OK; }
        if (b == 0 && train_recompute1())
            ASR_HIP(ctx, asr::launch_conv1_raw(st, tt.x[0], tw.w_dev[0], tt.x[1], B, g.H, g.W, g.cout, nullptr, nullptr, 2,
                                               tt.stats[0], pm(T, base + 2), pm(T, base + 1)));
        else if (b < 8)
            ASR_HIP(ctx, asr::launch_bn_apply(st, tt.z[b], tt.stats[b], pm(T, base + 2), pm(T, base + 1), tt.x[b + 1],
                                              B, g.H, g.W, g.cout, g.pool, 1, tt.zsel[b], tt.ztie[b],
                                              tt.stats[b] + 2 * g.cout));
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

// ASR_TRAIN_GATE_FWD / ASR_TRAIN_GATE_BWD = lead (>= 0; unset or negative: no gates).  The spectrogram tower is ~1/5 of
// the sheet tower's work in kernels that fill a fraction of the GPU; left to the hardware's queue arbitration its
// stream is served last (round-5 timeline: its forward ends 0.2 ms after the sheet tower's, its backward 0.75 ms after,
// alone on the GPU).  With gates the sheet tower's block b waits until the spectrogram tower is `lead` blocks further.
static int train_gate(int dir) {
    static const int g[2] = {getenv("ASR_TRAIN_GATE_FWD") ? atoi(getenv("ASR_TRAIN_GATE_FWD")) : -1,
                             getenv("ASR_TRAIN_GATE_BWD") ? atoi(getenv("ASR_TRAIN_GATE_BWD")) : -1};
    return g[dir];
}

static int gate_record(asr_ctx *ctx, hipEvent_t &e, hipStream_t st) {
    if (!e) ASR_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ASR_HIP(ctx, hipEventRecord(e, st));
    return ASR_OK;
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
    } else if (train_gate(0) >= 0 && train_stream(ctx, 0) != train_stream(ctx, 1)) {
        // gated order: the spectrogram tower is enqueued first and marks the end of each of its blocks; the sheet tower's
        // block b starts only when the spectrogram tower has finished block min(8, b + lead) - see train_gate()
        const int lead = train_gate(0);
        TrainState &T = *ctx->train;
        for (int b = 0; b < 9; ++b) {
            if ((rc = train_forward_block(ctx, 1, B, b, 0)) != ASR_OK) return rc;
            if ((rc = gate_record(ctx, T.gate[b], train_stream(ctx, 1))) != ASR_OK) return rc;
        }
        int waited = -1;
        for (int b = 0; b < 9; ++b) {
            const int need = std::min(8, b + lead);
            if (need > waited) { ASR_HIP(ctx, hipStreamWaitEvent(train_stream(ctx, 0), T.gate[need], 0)); waited = need; }
            if ((rc = train_forward_block(ctx, 0, B, b, 0)) != ASR_OK) return rc;
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
    int pre_rows = 0;               // > 0: the data gradient that wrote dA also left the BatchNorm-backward sums of the block
                                    // about to be processed in the tower's statistics table (asr::BnBwdFuse), that many rows
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
                                    : (g.pool && tt.zsel[b])       ? 4.0 * rows * g.cout * (2.75 + (tt.ztie[b] ? 1.0 / 64 : 0.0))   // (zsel + dA [+ ztie]) + (z + dA + dz)
                                                                   : 4.0 * rows * g.cout * (3.0 + (g.pool ? 0.5 : 2.0));
            ProfScope ps(ctx, bname, view, 12.0 * rows * g.cout + ((b == 0 && train_recompute1()) ? 2.0 * rows * 9.0 * g.cout : 0.0),
                         bn_bytes);
            // block 1: the apply pass is fused into the weight-gradient kernel, dz's only reader there
            if (b == 0 && train_recompute1())
                ASR_HIP(ctx, asr::launch_bn_bwd_conv1(st, tt.x[0], tw.w_dev[0], dA, tt.stats[0], pm(T, base + 2), pm(T, base + 1),
                                                      tt.partial, tt.sums, pg(T, base + 1), pg(T, base + 2), B, g.H, g.W,
                                                      g.cout, ex));
            else {
            const int pre = (phase != 2) ? S.pre_rows : 0;
            S.pre_rows = 0;
            ASR_HIP(ctx, asr::launch_bn_bwd(st, tt.z[b], (b == 0 && fuse1) ? nullptr : dz, dA, tt.stats[b], pm(T, base + 2), pm(T, base + 1),
                                            tt.partial, tt.sums, pg(T, base + 1), pg(T, base + 2), B, g.H, g.W, g.cout,
                                            g.pool, 1, ex, tt.zsel[b], tt.ztie[b],
                                            ctx->cfg.pool_ties == ASR_POOL_TIES_FIRST ? 1 : 0, tt.ticket,
                                            pre > 0 ? tt.fstats : nullptr, pre, pre > 0 ? tt.fstats + tt.fstats_doubles : nullptr));
            }
        }
        ctx->exch.phase = 0;
        if (phase == 1) return ASR_OK;
        char name[32];
        // ASR_TRAIN_WGRAD_LATE=1: the weight gradient of block b (side stream) is started AFTER its data gradient, so that
        // it lies beside the HBM-bound BatchNorm backward of block b - 1 instead of beside another MFMA-bound kernel.
        // Measured (round 5, three runs each): 10.78 ms against 10.40 ms - the two MFMA kernels share the matrix pipes
        // better than the model says (each keeps them < 50 % busy), and the late order leaves the last two weight
        // gradients (conv2, conv1: 0.76 ms) behind the end of the main stream's chain.  Default: round 4's order.
        static const bool wgrad_late = getenv("ASR_TRAIN_WGRAD_LATE") && getenv("ASR_TRAIN_WGRAD_LATE")[0] == '1';
        auto enqueue_wgrad = [&]() -> int {
        if (tt.wstream) {
            ASR_HIP(ctx, hipEventRecord(tt.e_dz[cur], st));
            ASR_HIP(ctx, hipStreamWaitEvent(ws, tt.e_dz[cur], 0));
        }
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
            return ASR_OK;
        };
        if (!(wgrad_late && b >= 1)) {
            const int rcw = enqueue_wgrad();
            if (rcw != ASR_OK) return rcw;
        }
        if (b >= 1) {
            snprintf(name, sizeof name, "train_dgrad_conv%d", b + 1);
            ProfScope ps(ctx, name, view, 2.0 * rows * 9.0 * g.cin * g.cout, 4.0 * rows * (g.cin + g.cout),
                         tt.dplan[b].symbol);
            // The data gradient of block b writes dL/d(output of block b-1): it can run the REDUCE pass of that block's
            // BatchNorm backward in its epilogue (asr::BnBwdFuse: reads z[b-1] - pooled blocks: zsel[b-1] - at the index of
            // every value it stores), so that bn_bwd_reduce_kernel and its re-read of the gradient leave the main stream's
            // chain.  Built, parity-green under both pooling rules - and measured
            // SLOWER: batch 512 update 12.36 ms against 10.57 ms with the separate pass.  The epilogue's loads of z
            // are used at once by a wave that has nothing else to issue (1-2 workgroups per CU): +70 % on the data
            // gradients, against 1.25 ms of HBM-bound reduce passes that mostly hide under the other stream's MFMA
            // kernels anyway.  Off by default (ASR_TRAIN_BNB_FUSE=1 switches it on; kept for a prefetching epilogue).
            // (round 6: compiled out unless the library is built with -DASR_BNB_FUSE_BUILD=1, see asr_kernels.h)
            static const bool bnb_fuse = ASR_BNB_FUSE_BUILD && (getenv("ASR_TRAIN_BNB_FUSE") && getenv("ASR_TRAIN_BNB_FUSE")[0] == '1') &&
                                         (getenv("ASR_TRAIN_FUSED_REDUCE") && getenv("ASR_TRAIN_FUSED_REDUCE")[0] == '1');
            const LayerGeom &gp = tw.g[b - 1];
            asr::BnBwdFuse bf{nullptr, nullptr, nullptr};
            if (bnb_fuse && !ex && b >= 2 && 2 * gp.cout <= 256 && (!gp.pool || tt.zsel[b - 1]) &&
                (!gp.pool || ctx->cfg.pool_ties == ASR_POOL_TIES_FIRST || tt.ztie[b - 1])) {
                bf.z = gp.pool ? tt.zsel[b - 1] : tt.z[b - 1];
                bf.tie = (gp.pool && ctx->cfg.pool_ties == ASR_POOL_TIES_ALL) ? tt.ztie[b - 1] : nullptr;
                bf.cst = tt.stats[b - 1];
            }
            int prow = 0;
            ASR_HIP(ctx, launch_conv_any(ctx, st, tt.dplan[b], dz, tt.wdgrad[b], nullptr, dB, B, nullptr,
                                         bf.z ? tt.fstats : nullptr, &prow, true, bf.z ? &bf : nullptr));
            S.pre_rows = bf.z ? prow : 0;           // 0: this plan writes no sums (direct-form build) - the reduce pass runs
            std::swap(dA, dB);
        }
        if (wgrad_late && b >= 1) {
            const int rcw = enqueue_wgrad();
            if (rcw != ASR_OK) return rcw;
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
    } else if (train_gate(1) >= 0 && train_stream(ctx, 0) != train_stream(ctx, 1)) {
        // gated order (see train_gate): spectrogram tower first, one gate after its tail (index 8) and after each block;
        // the sheet tower's block b (tail: b = 8) waits for the spectrogram tower's block max(0, b - lead)
        const int lead = train_gate(1);
        TrainState &T = *ctx->train;
        if ((rc = train_backward_tail(ctx, 1, B, row_lo, 0, S[1])) != ASR_OK) return rc;
        if ((rc = gate_record(ctx, T.gate[9 + 8], train_stream(ctx, 1))) != ASR_OK) return rc;
        for (int b = 7; b >= 0; --b) {
            if ((rc = train_backward_block(ctx, 1, B, b, 0, S[1])) != ASR_OK) return rc;
            if ((rc = gate_record(ctx, T.gate[9 + b], train_stream(ctx, 1))) != ASR_OK) return rc;
        }
        int waited = 9;
        for (int b = 8; b >= 0; --b) {
            const int need = std::max(0, b - lead);
            if (need < waited) { ASR_HIP(ctx, hipStreamWaitEvent(train_stream(ctx, 0), T.gate[9 + need], 0)); waited = need; }
            if (b == 8) rc = train_backward_tail(ctx, 0, B, row_lo, 0, S[0]);
            else rc = train_backward_block(ctx, 0, B, b, 0, S[0]);
            if (rc != ASR_OK) return rc;
        }
    } else {
        // ASR_TRAIN_BWD_ORDER=1 enqueues the spectrogram tower's backward pass (~1.5 ms of small kernels) first.  Measured:
        // 10.31 ms against 10.33 ms - the towers share the GPU either way, the order only decides which one crawls.
        static const bool spec_first = getenv("ASR_TRAIN_BWD_ORDER") && getenv("ASR_TRAIN_BWD_ORDER")[0] == '1';
        for (int k = 0; k < 2; ++k) {
            const int t = spec_first ? 1 - k : k;
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
    // The CCALayer + loss chain runs on the SHEET TOWER's stream when the towers have streams of their own (round 6): that
    // tower ends the forward pass last and starts the backward pass first, and with the chain on the main stream its
    // critical path crossed streams twice (towers -> main -> towers), ~25 us of event latency each way with the GPU idle.
    // Now only the spectrogram tower's "forward done" has to reach the sheet tower's stream, and the sheet tower's backward
    // pass follows the chain in stream order.  ASR_TRAIN_CCA_MAIN=1: the chain on the main stream as in rounds 2-5.
    static const bool cca_main = getenv("ASR_TRAIN_CCA_MAIN") && getenv("ASR_TRAIN_CCA_MAIN")[0] == '1';
    hipStream_t cs = ctx->stream;
    if (!dp && !forward_only && !cca_main && train_stream(ctx, 0) != ctx->stream && train_stream(ctx, 0) != train_stream(ctx, 1)) {
        cs = train_stream(ctx, 0);
        ASR_HIP(ctx, hipStreamWaitEvent(cs, ctx->vdone[1], 0));       // (recorded at the end of train_forward_towers)
    } else if ((rc = join_views(ctx)) != ASR_OK) return rc;
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
        ProfScope ps(ctx, "train_cca_loss", 0, 0.0, 0.0, "", cs);
        ASR_HIP(ctx, asr::launch_cca_train(cs, H1, H2, (int)n_global, pm(T, 90), pm(T, 90), ctx->cfg.r1,
                                           ctx->cfg.r2, ctx->cfg.rT, ctx->cfg.alpha, ctx->cfg.gamma, T.cca_ws,
                                           T.loss_dev, lv1, lv2, forward_only ? nullptr : dH1,
                                           forward_only ? nullptr : dH2, ctx->loss_weight, ctx->loss_symmetric));
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
    ASR_HIP(ctx, hipEventRecord(T.cca_done, cs));
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

}  // namespace asr_detail

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

int asr_comm_timing(asr_ctx *ctx, int enable, double *ms, int64_t *calls) {
    if (!ctx) return ASR_ERR_INVALID;
    Comm *c = ctx->comm.get();
    if (ms) *ms = 0.0;
    if (calls) *calls = 0;
    if (!c) return ASR_OK;
    ASR_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = sync_all(ctx);
    if (rc != ASR_OK) return rc;
    double total = c->host_ms;
    for (auto &pr : c->timed) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) total += (double)t;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    c->timed.clear();
    if (ms) *ms = total;
    if (calls) *calls = c->timed_calls;
    c->host_ms = 0.0;
    c->timed_calls = 0;
    c->timing = enable != 0;
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

int asr_set_objective(asr_ctx *ctx, float weight, float gamma, int symmetric) {
    if (!ctx) return ASR_ERR_INVALID;
    if (!(weight > 0.0f) || !(gamma == gamma) || (symmetric != 0 && symmetric != 1))
        return fail(ctx, ASR_ERR_INVALID, "set_objective: weight must be positive, symmetric 0 or 1");
    ctx->loss_weight = weight;
    ctx->cfg.gamma = gamma;
    ctx->loss_symmetric = symmetric;
    return ASR_OK;
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
    if (e == hipSuccess) e = asr::launch_rank_loss(ctx->stream, d, d + (size_t)n * 32, (int)n, ctx->cfg.gamma, d + (size_t)n * 64,
                                                   ctx->loss_weight, ctx->loss_symmetric);
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
//       6 grad of parameter `index`, 7 master value of parameter `index`, 8 [loss | corr(32)],
//       9 zsel (selected raw value per pooling window), 10 the set of window elements that share the maximum
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
        case 10: {              // pooled blocks: which window elements share the maximum y (4-bit sets as floats, N x H/2 x W/2 x C)
            if (view < 1 || view > 2 || index < 1 || index > 7 || !ctx->tw[view - 1].g[index].pool)
                return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: pool mask of a block that is not pooled");
            const LayerGeom &gz = ctx->tw[view - 1].g[index];
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
    if (kind == 10) {           // derived on request from z and the batch statistics of the last training forward
        if (batch < 1 || batch > T.B) return fail(ctx, ASR_ERR_INVALID, "debug_train_tensor: batch");
        const LayerGeom &gz = ctx->tw[view - 1].g[index];
        float *tmp = nullptr;
        ASR_HIP(ctx, hipMalloc((void **)&tmp, (size_t)n * sizeof(float)));
        hipError_t e = asr::launch_pool_mask(ctx->stream, T.tw[view - 1].z[index], T.tw[view - 1].stats[index],
                                             T.tw[view - 1].stats[index] + 2 * gz.cout, tmp, (int)batch, gz.H, gz.W, gz.cout);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(out, tmp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail(ctx, ASR_ERR_HIP, "debug_train_tensor: pool mask: %s", hipGetErrorString(e));
        return ASR_OK;
    }
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
                                  dH1 ? dH2d : nullptr, ctx->loss_weight, ctx->loss_symmetric);
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
