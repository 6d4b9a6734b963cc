#!/usr/bin/env python
"""Secondary measurements, one function each - `bench.py` calls them for the "secondary" block of its JSON
line (N = 1) and this file prints them as JSON lines when run by itself:

  train    BASELINE configs[2]: full training step at batch 512 (updates/s, TFLOP/s vs the fp32-MFMA peak)
  cca      configs[3]: CCA('svd').fit on 25 000 x 32 features (ms, GB/s vs HBM)
  topk     configs[4] per-GPU shard: top-25 of 1024 queries against 250 k codes and of 64 queries against 2 M codes
  rank     eval_retrieval's ranking at n = 1000 / 2000 / 8000
  dropin   the reference's own API: RetrievalWrapper.compute_view_1 + compute_view_2 + eval_retrieval at n = 2000
           (eval_models.sh:15), host arrays in, host arrays out
  refine   refine_cca.py's work on 25 000 host pairs (tower outputs + CCA fit + write-back)

    python tools/bench_secondary.py [train] [cca] [topk] [rank] [dropin] [refine]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from audio_sheet_retrieval_amd import _lib  # noqa: E402
from audio_sheet_retrieval_amd.utils import synth_data  # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402

MODEL = os.environ.get("ASR_BENCH_MODEL", "mutopia_ccal_cont")          # _rsz for side measurements


def fwd_flop_per_pair(model):
    """conv MACs x 2 of both towers, per pair"""
    return 552594048 if model.endswith("_rsz") else 425302464


FWD_FLOP_PER_PAIR = fwd_flop_per_pair(MODEL)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def timeit(fn, sync, reps, warm=2):
    """median over 5 groups of back-to-back calls (the box shows one 50-80 ms stall every few dozen launches of any
    kernel; a mean over one run of calls reports it as a 3x slower kernel)"""
    for _ in range(warm):
        fn()
    sync()
    per = max(1, reps // 5)
    groups = []
    for _ in range(5 if reps >= 5 else 1):
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        sync()
        groups.append((time.perf_counter() - t0) / per)
    return float(np.median(groups))


def measure_train(eng, B=512, verbose=False, model=None):
    import ctypes
    model = model or MODEL
    sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
    x1 = (sheet.astype(np.float32) / np.float32(255))
    if model.endswith("_rsz"):                 # the training step takes PREPARED sheets: network resolution
        x1 = np.ascontiguousarray(0.25 * (x1[:, :, 0::2, 0::2] + x1[:, :, 0::2, 1::2] + x1[:, :, 1::2, 0::2] + x1[:, :, 1::2, 1::2]))
    eng.train_begin(B)
    d1 = eng.alloc(x1.nbytes).upload(x1)
    d2 = eng.alloc(spec.nbytes).upload(spec)
    loss = ctypes.c_float()
    corr = np.empty(32, np.float32)

    def step():
        eng._check(eng.lib.asr_train_step_dev(eng.ctx, d1.ptr, d2.ptr, B, 0.002, ctypes.byref(loss), corr.ctypes.data))
    eng.profile_enable(False)
    dt = timeit(step, eng.sync, 10)
    eng.profile_reset(); eng.profile_enable(True)
    for _ in range(5):
        step()
    eng.sync(); eng.profile_enable(False)
    prof = sorted([p for p in eng.profile() if p["launches"] > 0], key=lambda p: -p["total_ms"])
    top = {p["name"]: round(p["total_ms"] / 5, 3) for p in prof[:24]}      # ms per step, summed over launches
    if verbose or os.environ.get("ASR_TRAIN_PROFILE_ALL"):                 # every stage, with its algorithmic GB/s
        for p in prof:
            sys.stderr.write("%-28s %8.3f ms  %7.1f GB/s  %6.1f TFLOP/s\n" % (
                p["name"], p["total_ms"] / 5, p["bytes"] * p["launches"] / max(p["total_ms"], 1e-9) / 1e6,
                p["flops"] * p["launches"] / max(p["total_ms"], 1e-9) / 1e9))
    top["_sum_all"] = round(sum(p["total_ms"] for p in prof) / 5, 3)
    top["_sum_v1"] = round(sum(p["total_ms"] for p in prof if p["name"].endswith("_v1")) / 5, 3)
    top["_sum_v2"] = round(sum(p["total_ms"] for p in prof if p["name"].endswith("_v2")) / 5, 3)
    eng.train_end()
    d1.free(); d2.free()
    tfl = 3.0 * B * fwd_flop_per_pair(model) / dt / 1e12
    dom = prof[0] if prof else None
    rsz = model.endswith("_rsz")
    return {"what": "train_step", "config": "BASELINE configs[2]: full training step, batch 512, %s" % model,
            "pool_ties": eng.pool_ties,
            "batch": B, "ms_per_step": dt * 1e3, "updates_per_s": 1.0 / dt, "pairs_per_s": B / dt,
            "loss": float(loss.value), "tflops_fwd_bwd": tfl,
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / PEAK_F32_MFMA_TFLOPS,
                         "work": "3 x forward conv FLOP (fwd + dgrad + wgrad, SURVEY 8d) x 512 pairs per step",
                         "dominant_stage": None if dom is None else dom["name"],
                         "dominant_stage_ms": None if dom is None else round(dom["total_ms"] / 5, 3)},
            "parity_test": "tests/test_gpu_train_routed.py::test_routed_gradients_at_batch_512[%s]" % model if rsz else
                           "tests/test_gpu_bench_sizes.py::test_full_training_step_batch_512_matches_oracle",
            "kernel_ms": top}


def measure_model_headline(model, n=1000, steps=20, train_batch=512):
    """The headline step (both towers + CCA + ranking of n resident pairs) and the batch-512 training step for ANOTHER
    model variant - mutopia_ccal_cont_rsz, the variant the reference ships weights for and evaluates
    (eval_models.sh:5, tutorials/params_all_split_mutopia_full_aug.pkl) - on an engine of its own."""
    eng = _lib.Engine(model)
    eng.set_params(synth_data.synth_params(param_shapes(model), seed=1, trained_like=True))
    sheet, spec = synth_data.synth_pairs(np.arange(n), seed=23)
    d_sheet, d_spec = eng.alloc(sheet.nbytes).upload(sheet), eng.alloc(spec.nbytes).upload(spec)
    d_lv1, d_lv2 = eng.alloc(n * 128), eng.alloc(n * 128)
    d_ranks, d_dstar, d_ties = eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4)

    def step():
        eng.embed_view1_dev(d_sheet.ptr, _lib.IN_U8_RAW, n, d_lv1.ptr)
        eng.embed_view2_dev(d_spec.ptr, n, d_lv2.ptr)
        eng.rank_dev(d_lv1.ptr, n, d_lv2.ptr, n, d_ranks.ptr, d_dstar.ptr, d_ties.ptr)
    for _ in range(3):
        step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    for b in (d_sheet, d_spec, d_lv1, d_lv2, d_ranks, d_dstar, d_ties):
        b.free()
    tfl = n * fwd_flop_per_pair(model) / dt / 1e12
    out = {"what": "headline step of another model variant", "model": model, "pairs": n, "steps": steps,
           "value": n / dt, "unit": "pairs/s", "ms_per_step": dt * 1e3,
           "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": tfl / PEAK_F32_MFMA_TFLOPS,
                        "work": "%d FLOP per pair (SURVEY 8d, both towers, direct form) - whole step, all kernels"
                                % fwd_flop_per_pair(model)},
           "parity_test": "tests/test_gpu_bench_sizes.py::test_bench_launch_matches_oracle[%s-0]" % model}
    try:
        out["train_step_b%d" % train_batch] = measure_train(eng, train_batch, model=model)
    except Exception as e:
        out["train_step_b%d" % train_batch] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    return out


def measure_cca(eng, n=25000):
    rng = np.random.default_rng(0)
    z = rng.standard_normal((n, 32))
    H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
    H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
    dH1, dH2 = eng.alloc(H1.nbytes).upload(H1), eng.alloc(H2.nbytes).upload(H2)
    dU, dV, dm, dc = eng.alloc(4096), eng.alloc(4096), eng.alloc(256), eng.alloc(256)
    dt = timeit(lambda: eng.cca_fit_dev(dH1.ptr, dH2.ptr, n, dU.ptr, dV.ptr, dm.ptr, dc.ptr), eng.sync, 20)
    eng.cca_fit(H1, H2)                                   # first call: workspace allocation
    t0 = time.perf_counter(); eng.cca_fit(H1, H2); host = time.perf_counter() - t0
    for b in (dH1, dH2, dU, dV, dm, dc):
        b.free()
    gbs = 256.0 * n / dt / 1e9
    return {"what": "cca_fit", "config": "BASELINE configs[3]: 25000-sample CCA re-estimation", "n": n,
            "ms_device_resident": dt * 1e3, "ms_host_buffers": host * 1e3,
            # 6.4 MB: neither roof is near.  The time is a chain of dependent steps - two passes over the samples (means,
            # centred second moments: ~0.06 ms) and the 32x32 float64 solve (two symmetric eigen-decompositions + one
            # SVD by cyclic Jacobi in one workgroup: ~0.4 ms, profiles/r04 cca_fit trace) - so the bound is latency
            "roofline": {"bound": "latency", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "solve_share": 0.8,
                         "work": "256 B per sample (two (n,32) float32 arrays read once, SURVEY 8d); the kernels read "
                                 "them twice (means, then centred second moments) and the 32x32 float64 solve is a "
                                 "latency chain of ~0.4 ms of the ~0.5: `frac` against HBM is reported for "
                                 "completeness, the size is latency-bound"},
            "parity_test": "tests/test_reference_golden.py::test_device_cca_fit_matches_reference[cca_25000]"}


def measure_topk(eng, n_db=250000, n_q=1024, k=25, reps=3):
    rng = np.random.default_rng(1)
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = (db[rng.integers(0, n_db, n_q)] + 0.1 * rng.standard_normal((n_q, 32)).astype(np.float32)).astype(np.float32)
    ddb, dq = eng.alloc(db.nbytes).upload(db), eng.alloc(q.nbytes).upload(q)
    di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
    dt_stateless = timeit(lambda: eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr), eng.sync, reps, warm=1)
    # the live server's shape (audio_sheet_server.py:496-522, 530-563): the data base is loaded once (asr_db_create: norms,
    # unit-length copy) and queried per frame - this is the figure the roofline below is quoted on
    pool = eng.db_create(ddb.ptr, n_db)
    # 5 groups of 20 back-to-back calls (the server's frames arrive back to back; with 2 calls per group the ~20 us of the
    # closing synchronisation were half of them in every call's figure: 64 x 2 M read 0.123 ms here, 0.112 in tools/ab_topk.py)
    dt = timeit(lambda: pool.topk_dev(dq.ptr, n_q, k, di.ptr, dd.ptr), eng.sync, max(reps, 100), warm=2)
    pool.close()
    for b in (ddb, dq, di, dd):
        b.free()
    tfl = 64.0 * n_db * n_q / dt / 1e12
    gbs = 128.0 * n_db * max(1, -(-n_q // 16)) / dt / 1e9        # a workgroup streams its pool slice once per 16 queries
    few = n_q <= 64
    if n_q <= 2:
        test = "tests/test_gpu_code_db.py::test_single_query_scan_path_equals_the_oracle"
    else:
        test = "tests/test_gpu_bench_sizes.py::test_topk_at_config5_pool_sizes[%d]" % n_db
    return {"what": "topk", "config": "BASELINE configs[4] per-GPU shard: %d codes (of a 2M pool), %d queries, k=%d"
                                      % (n_db, n_q, k),
            "n_db": n_db, "n_q": n_q, "k": k, "ms": dt * 1e3, "ms_stateless_call": dt_stateless * 1e3,
            "what_is_timed": "asr_topk_db_dev against a resident data base (asr_db_create once); ms_stateless_call: "
                             "asr_topk_dev, which derives the pool's norms on every call",
            "queries_per_s": n_q / dt,
            "pair_distances_per_s": n_db * n_q / dt,
            "roofline": ({"bound": "hbm", "achieved": 128.0 * n_db / dt / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": 128.0 * n_db / dt / 1e9 / PEAK_HBM_GBS,
                          "work": "128 B per pool code, streamed once (SURVEY 8d: few queries against a large pool)"}
                         if few else
                         {"bound": "mfma", "achieved": tfl, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                          "frac": tfl / PEAK_F32_MFMA_TFLOPS, "work": "64 FLOP per (query, code) pair (SURVEY 8d)",
                          "l2_stream_gbs": gbs}),
            "parity_test": test}


def measure_rank(eng, n):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((n, 32)).astype(np.float32)
    b = (rng.standard_normal((n, 32)) + a).astype(np.float32)
    da, db_ = eng.alloc(a.nbytes).upload(a), eng.alloc(b.nbytes).upload(b)
    dr, dd, dti = eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4)
    dt = timeit(lambda: eng.rank_dev(da.ptr, n, db_.ptr, n, dr.ptr, dd.ptr, dti.ptr), eng.sync, 10)
    for x in (da, db_, dr, dd, dti):
        x.free()
    return {"what": "rank", "n": n, "ms": dt * 1e3, "pair_distances_per_s": n * n / dt}


def _param_pickle(tmpdir):
    import pickle
    params = synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=True)
    path = os.path.join(tmpdir, "params_bench.pkl")
    with open(path, "wb") as fp:
        pickle.dump(params, fp, protocol=2)
    return path


def measure_dropin(n=2000, repeats=5, sheets_u8=None, specs=None):
    """The reference's embedding API end to end on HOST arrays (what audio_sheet_server.py / run_eval.py do):
    RetrievalWrapper(model, pickle, prepare_view_1=model.prepare); compute_view_1(uint8 sheets) + compute_view_2(specs)
    + eval_retrieval - wall clock per call sequence, pageable caller memory, every result back in host arrays."""
    import importlib
    import tempfile
    from audio_sheet_retrieval_amd.retrieval_wrapper import RetrievalWrapper
    from audio_sheet_retrieval_amd.utils.train_dcca_pool import eval_retrieval
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + MODEL)
    if sheets_u8 is None:
        sheets_u8, specs = synth_data.synth_pairs(np.arange(n), seed=23)
    sheets_u8, specs = np.ascontiguousarray(sheets_u8[:n]), np.ascontiguousarray(specs[:n])
    with tempfile.TemporaryDirectory(prefix="asr_bench_") as tmp:
        rw = RetrievalWrapper(model, _param_pickle(tmp), prepare_view_1=model.prepare)
    eng = rw.compute_v1_latent.engine
    out = {}
    for label, X in (("u8", sheets_u8), ("f32", sheets_u8.astype(np.float32))):
        rw.compute_view_1(X[:600]); rw.compute_view_2(specs[:600])          # tuner, staging slots, first touch
        times, parts = [], []
        for _ in range(repeats):
            t0 = time.perf_counter()
            lv1 = rw.compute_view_1(X)
            t1 = time.perf_counter()
            lv2 = rw.compute_view_2(specs)
            t2 = time.perf_counter()
            res = eval_retrieval(lv1, lv2, engine=eng)
            t3 = time.perf_counter()
            times.append(t3 - t0)
            parts.append((t1 - t0, t2 - t1, t3 - t2))
        k = int(np.argsort(times)[len(times) // 2])
        out[label] = {"pairs_per_s": n / times[k], "ms": times[k] * 1e3, "min_ms": min(times) * 1e3, "max_ms": max(times) * 1e3,
                      "ms_compute_view_1": parts[k][0] * 1e3, "ms_compute_view_2": parts[k][1] * 1e3,
                      "ms_eval_retrieval": parts[k][2] * 1e3,
                      "h2d_bytes": int(X.nbytes + specs.nbytes), "median_rank": float(res[1])}
    eng.close()
    return {"what": "dropin_api", "n": n, "model": MODEL,
            "config": "RetrievalWrapper.compute_view_1 + compute_view_2 + eval_retrieval on %d host pairs "
                      "(eval_models.sh:15), pageable NumPy arrays in and out" % n,
            "value_dropin_api": out["u8"]["pairs_per_s"], "unit": "pairs/s",
            "uint8_sheets": out["u8"], "float32_sheets": out["f32"],
            "parity_test": "tests/test_gpu_dropin_api.py::test_compute_view_fast_path_is_bit_identical_to_the_chunked_host_path"}


def measure_refine(n=25000, unique=1000, sheets_u8=None, specs=None):
    """refine_cca.py:92-107 on n host pairs: tower outputs of both views (the reference: 2 x n/10 calls of 10 samples),
    CCA('svd').fit, write-back of mean1/mean2/U/V.  The host arrays are float32 sheets holding 0..255 (what the pool
    yields, utils/data_pools.py:203-228) built by repeating `unique` synthetic pairs; their synthesis is not timed."""
    import importlib
    from audio_sheet_retrieval_amd import network, refine_cca
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + MODEL)
    if sheets_u8 is None:
        sheets_u8, specs = synth_data.synth_pairs(np.arange(unique), seed=23)
    reps = -(-n // sheets_u8.shape[0])
    out = {}
    layers = model.build_model(show_model=False)
    network.set_all_param_values(layers, synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=True))
    for label, dtype in (("f32", np.float32), ("u8", np.uint8)):
        X = np.concatenate([sheets_u8.astype(dtype)] * reps)[:n]
        Z = np.concatenate([specs] * reps)[:n]
        refine_cca.estimate(layers, X[:600], Z[:600], model.prepare, verbose=False)        # warm-up
        t0 = time.perf_counter()
        cca = refine_cca.estimate(layers, X, Z, model.prepare, verbose=False)
        dt = time.perf_counter() - t0
        out[label] = {"s": dt, "pairs_per_s": n / dt, "h2d_bytes": int(X.nbytes + Z.nbytes),
                      "h2d_gbs": (X.nbytes + Z.nbytes) / dt / 1e9}
        del X, Z
    layers[0].net.engine.close()
    return {"what": "refine_cca", "n": n, "model": MODEL,
            "config": "refine_cca.py on %d host pairs (README.md:104-107): tower outputs + CCA('svd').fit + write-back" % n,
            "refine_cca_s": out["f32"]["s"], "float32_sheets": out["f32"], "uint8_sheets": out["u8"],
            "first_coeff_finite": bool(np.isfinite(cca.U).all()),
            "parity_test": "tests/test_gpu_dropin_api.py::test_run_eval_and_refine_cca_fast_path_equal_the_chunked_route"}


def main():
    which = sys.argv[1:] or ["train", "cca", "topk", "rank", "dropin", "refine"]
    eng = _lib.Engine(MODEL)
    eng.set_params(synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=False))
    if "train" in which:
        print(json.dumps(measure_train(eng)), flush=True)
    if "cca" in which:
        print(json.dumps(measure_cca(eng)), flush=True)
    if "topk" in which:
        print(json.dumps(measure_topk(eng, 250000, 1024)), flush=True)
        print(json.dumps(measure_topk(eng, 2000000, 64)), flush=True)
    if "rank" in which:
        for n in (1000, 2000, 8000):
            print(json.dumps(measure_rank(eng, n)), flush=True)
    eng.close()
    if "dropin" in which:
        print(json.dumps(measure_dropin()), flush=True)
    if "refine" in which:
        print(json.dumps(measure_refine()), flush=True)


if __name__ == "__main__":
    main()
