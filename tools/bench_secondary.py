#!/usr/bin/env python
"""Secondary measurements: training updates/s at B=512 (BASELINE config 3),
CCA re-estimation at n=25000 (config 4), top-k against one GPU's shard of a 2M pool
(config 5), ranking at the reference's default n_test=2000.  One JSON line each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_sheet_retrieval_amd import _lib  # noqa: E402
from audio_sheet_retrieval_amd.utils import synth_data  # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402

MODEL = os.environ.get("ASR_BENCH_MODEL", "mutopia_ccal_cont")          # _rsz for side measurements
FWD_FLOP_PER_PAIR = 552594048 if MODEL.endswith("_rsz") else 425302464


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


def main():
    which = sys.argv[1:] or ["train", "cca", "topk", "rank"]
    eng = _lib.Engine(MODEL)
    eng.set_params(synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=False))
    if "train" in which:
        B = 512
        sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
        x1 = (sheet.astype(np.float32) / np.float32(255))
        if MODEL.endswith("_rsz"):                 # the training step takes PREPARED sheets: network resolution
            x1 = np.ascontiguousarray(0.25 * (x1[:, :, 0::2, 0::2] + x1[:, :, 0::2, 1::2] + x1[:, :, 1::2, 0::2] + x1[:, :, 1::2, 1::2]))
        eng.train_begin(B)
        d1 = eng.alloc(x1.nbytes).upload(x1)
        d2 = eng.alloc(spec.nbytes).upload(spec)
        import ctypes
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
        prof = sorted(eng.profile(), key=lambda p: -p["total_ms"])
        top = {p["name"]: round(p["total_ms"] / 5, 3) for p in prof[:24]}      # ms per step, summed over launches
        if os.environ.get("ASR_TRAIN_PROFILE_ALL"):                             # every stage, with its algorithmic GB/s
            for p in prof:
                sys.stderr.write("%-28s %8.3f ms  %7.1f GB/s  %6.1f TFLOP/s\n" % (
                    p["name"], p["total_ms"] / 5, p["bytes"] * p["launches"] / max(p["total_ms"], 1e-9) / 1e6,
                    p["flops"] * p["launches"] / max(p["total_ms"], 1e-9) / 1e9))
        top["_sum_all"] = round(sum(p["total_ms"] for p in prof) / 5, 3)
        top["_sum_v1"] = round(sum(p["total_ms"] for p in prof if p["name"].endswith("_v1")) / 5, 3)
        top["_sum_v2"] = round(sum(p["total_ms"] for p in prof if p["name"].endswith("_v2")) / 5, 3)
        print(json.dumps({"what": "train_step", "config": "BASELINE configs[2]: full training step, batch 512, %s" % MODEL,
                          "batch": B, "ms_per_step": dt * 1e3, "updates_per_s": 1.0 / dt, "pairs_per_s": B / dt,
                          "loss": float(loss.value),
                          "tflops_fwd_bwd": 3.0 * B * FWD_FLOP_PER_PAIR / dt / 1e12,
                          "kernel_ms": top}))
        eng.train_end()
    if "cca" in which:
        rng = np.random.default_rng(0)
        n = 25000
        z = rng.standard_normal((n, 32))
        H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
        H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
        dH1, dH2 = eng.alloc(H1.nbytes).upload(H1), eng.alloc(H2.nbytes).upload(H2)
        dU, dV, dm, dc = eng.alloc(4096), eng.alloc(4096), eng.alloc(256), eng.alloc(256)
        dt = timeit(lambda: eng.cca_fit_dev(dH1.ptr, dH2.ptr, n, dU.ptr, dV.ptr, dm.ptr, dc.ptr), eng.sync, 20)
        t0 = time.perf_counter(); eng.cca_fit(H1, H2); host = time.perf_counter() - t0
        print(json.dumps({"what": "cca_fit", "config": "BASELINE configs[3]: 25000-sample CCA re-estimation", "n": n,
                          "ms_device_resident": dt * 1e3, "ms_host_buffers": host * 1e3,
                          "GBps": 2 * 2 * n * 32 * 4 / dt / 1e9}))
    if "topk" in which:
        rng = np.random.default_rng(1)
        n_db, n_q, k = 250000, 1024, 25
        db = rng.standard_normal((n_db, 32)).astype(np.float32)
        db /= np.linalg.norm(db, axis=1, keepdims=True)
        q = db[rng.integers(0, n_db, n_q)] + 0.1 * rng.standard_normal((n_q, 32)).astype(np.float32)
        ddb, dq = eng.alloc(db.nbytes).upload(db), eng.alloc(q.astype(np.float32).nbytes).upload(q.astype(np.float32))
        di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
        dt = timeit(lambda: eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr), eng.sync, 3, warm=1)
        print(json.dumps({"what": "topk", "config": "BASELINE configs[4] per-GPU shard: 250k of a 2M pool, k=25",
                          "n_db": n_db, "n_q": n_q, "k": k, "ms": dt * 1e3, "queries_per_s": n_q / dt,
                          "pair_distances_per_s": n_db * n_q / dt}))
    if "rank" in which:
        rng = np.random.default_rng(2)
        for n in (1000, 2000, 8000):
            a = rng.standard_normal((n, 32)).astype(np.float32)
            b = (rng.standard_normal((n, 32)) + a).astype(np.float32)
            da, db_ = eng.alloc(a.nbytes).upload(a), eng.alloc(b.nbytes).upload(b)
            dr, dd, dti = eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4)
            dt = timeit(lambda: eng.rank_dev(da.ptr, n, db_.ptr, n, dr.ptr, dd.ptr, dti.ptr), eng.sync, 10)
            print(json.dumps({"what": "rank", "n": n, "ms": dt * 1e3, "pair_distances_per_s": n * n / dt}))
    eng.close()


if __name__ == "__main__":
    main()
