import sys, time, numpy as np
sys.path.insert(0, '.')
from audio_sheet_retrieval_amd import _lib
eng = _lib.Engine("mutopia_ccal_cont")
rng = np.random.default_rng(0); n = 25000
z = rng.standard_normal((n, 32))
H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32))).astype(np.float32)
dH1, dH2 = eng.alloc(H1.nbytes).upload(H1), eng.alloc(H2.nbytes).upload(H2)
dU, dV, dm, dc = eng.alloc(4096), eng.alloc(4096), eng.alloc(256), eng.alloc(256)
ts = []
for i in range(40):
    eng.sync(); t0 = time.perf_counter(); eng.cca_fit_dev(dH1.ptr, dH2.ptr, n, dU.ptr, dV.ptr, dm.ptr, dc.ptr); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join('%.2f' % t for t in ts))
