import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_sheet_retrieval_amd import _lib
eng = _lib.Engine("mutopia_ccal_cont")
rng = np.random.default_rng(1)
n_db = 250000
db = rng.standard_normal((n_db, 32)).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
ddb = eng.alloc(db.nbytes).upload(db)
for n_q in (16, 64, 256, 1024, 4096):
    for k in (25, 128):
        q = (db[rng.integers(0, n_db, n_q)] + 0.1 * rng.standard_normal((n_q, 32))).astype(np.float32)
        dq = eng.alloc(q.nbytes).upload(q)
        di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
        for _ in range(2): eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr)
        eng.sync(); t0 = time.perf_counter()
        for _ in range(5): eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr)
        eng.sync(); dt = (time.perf_counter() - t0) / 5
        print("n_q %5d k %3d: %.3f ms  (%.1f G pair-dist/s)" % (n_q, k, dt * 1e3, n_db * n_q / dt / 1e9))
