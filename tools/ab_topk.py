#!/usr/bin/env python
"""Timing of the retrieval entry points on one shape, for A/B runs under different ASR_TOPK_* / ASR_RANK_* settings and
under rocprofv3:  python tools/ab_topk.py <n_db> <n_q> [k] [mode: db|stateless|fused|rank] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_sheet_retrieval_amd import _lib  # noqa: E402


def main():
    n_db, n_q = int(sys.argv[1]), int(sys.argv[2])
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    mode = sys.argv[4] if len(sys.argv) > 4 else "db"
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    rng = np.random.default_rng(1)
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    kk = max(1, n_db // n_q)
    q = (db[(np.arange(n_q) * kk) % n_db] + 0.1 * rng.standard_normal((n_q, 32)).astype(np.float32)).astype(np.float32)
    alt = os.environ.get("ASR_LIB_PATH")              # an ablation build of the library (tools/ab_topk_abl.sh)
    eng = _lib.Engine("mutopia_ccal_cont", lib=_lib.load_library(alt) if alt else None)
    ddb, dq = eng.alloc(db.nbytes).upload(db), eng.alloc(q.nbytes).upload(q)
    di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
    dr, ds, dt = eng.alloc(n_q * 4), eng.alloc(n_q * 8), eng.alloc(n_q * 4)
    pool = eng.db_create(ddb.ptr, n_db)
    fn = {"db": lambda: pool.topk_dev(dq.ptr, n_q, k, di.ptr, dd.ptr),
          "stateless": lambda: eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr),
          "fused": lambda: pool.topk_rank_dev(dq.ptr, n_q, k, di.ptr, dd.ptr, dr.ptr, ds.ptr, dt.ptr),
          "rank": lambda: pool.rank_dev(dq.ptr, n_q, dr.ptr, ds.ptr, dt.ptr)}[mode]
    for _ in range(3):
        fn()
    eng.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        ts.append((time.perf_counter() - t0) / reps)
    env = " ".join("%s=%s" % (a, b) for a, b in sorted(os.environ.items()) if a.startswith(("ASR_TOPK", "ASR_RANK")))
    if os.environ.get("ASR_TF_TRACE_WGS"):
        print_trace(eng, int(os.environ["ASR_TF_TRACE_WGS"]))
    print("%s n_db=%d n_q=%d k=%d [%s]: median %.4f ms (min %.4f)" % (mode, n_db, n_q, k, env, np.median(ts) * 1e3,
                                                                      min(ts) * 1e3), flush=True)


def print_trace(eng, n_wg):
    """trace build (tools/ab_topk_abl.sh 4): where the filter workgroups of the LAST launch spent their time"""
    import ctypes
    lib = eng.lib
    if not hasattr(lib, "asr_debug_tf_trace"):
        return
    n_wg = min(n_wg, 8192)
    buf = (ctypes.c_ulonglong * (8192 * 8))()
    lib.asr_debug_tf_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.asr_debug_tf_trace(buf, 8192 * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8)[:n_wg].astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda ticks: ticks / 100.0                                   # wall_clock64: 100 MHz
    names = ["prologue", "main loop", "final compaction", "write-out"]
    print("trace of %d workgroups: first start .. last end %.1f us; starts spread over %.1f us" %
          (n_wg, us(t[:, 4].max() - t0), us(t[:, 0].max() - t0)))
    for i, nm in enumerate(names):
        d = us(t[:, i + 1] - t[:, i])
        print("   %-17s median %6.1f us   p10 %6.1f   p90 %6.1f   max %6.1f" % (nm, np.median(d), np.percentile(d, 10),
                                                                                 np.percentile(d, 90), d.max()))
    life = us(t[:, 4] - t[:, 0])
    print("   %-17s median %6.1f us   p10 %6.1f   p90 %6.1f   max %6.1f" % ("lifetime", np.median(life), np.percentile(life, 10),
                                                                             np.percentile(life, 90), life.max()))
    print("   rounds median %d max %d; compactions median %d max %d; tiles per workgroup %d" %
          (np.median(t[:, 5]), t[:, 5].max(), np.median(t[:, 6]), t[:, 6].max(), np.median(t[:, 7])))
    order = np.argsort(t[:, 0])
    st = us(t[order, 0] - t0)
    print("   start times (us) of workgroups in start order, every 64th:", " ".join("%.0f" % v for v in st[::64]))


if __name__ == "__main__":
    main()
