import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_sheet_retrieval_amd import _lib
n_q, k = int(sys.argv[1]), int(sys.argv[2])
eng = _lib.Engine("mutopia_ccal_cont")
rng = np.random.default_rng(1)
n_db = int(sys.argv[3]) if len(sys.argv) > 3 else 250000
db = rng.standard_normal((n_db, 32)).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
ddb = eng.alloc(db.nbytes).upload(db)
q = (db[rng.integers(0, n_db, n_q)] + 0.1 * rng.standard_normal((n_q, 32))).astype(np.float32)
dq = eng.alloc(q.nbytes).upload(q)
di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
for _ in range(5): eng.topk_dev(ddb.ptr, n_db, dq.ptr, n_q, k, di.ptr, dd.ptr)
eng.sync()
