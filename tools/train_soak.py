import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
from audio_sheet_retrieval_amd import _lib
from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
M = "mutopia_ccal_cont"
eng = _lib.Engine(M)
eng.set_params(synth_data.synth_params(param_shapes(M), seed=1, trained_like=False))
B = 512
eng.train_begin(B)
bufs = []
for k in range(4):                      # four different batches, rotated
    sheet, spec = synth_data.synth_pairs(np.arange(B) + 1000 * k, seed=23 + k)
    x1 = sheet.astype(np.float32) / np.float32(255)
    bufs.append((eng.alloc(x1.nbytes).upload(x1), eng.alloc(spec.nbytes).upload(spec)))
loss = ctypes.c_float(); corr = np.empty(32, np.float32)
n = int(sys.argv[1]); losses = []
t0 = time.perf_counter()
for i in range(n):
    d1, d2 = bufs[i % 4]
    eng._check(eng.lib.asr_train_step_dev(eng.ctx, d1.ptr, d2.ptr, B, 0.002, ctypes.byref(loss), corr.ctypes.data))
    if i % 50 == 49:
        eng.sync(); losses.append(float(loss.value))
eng.sync(); dt = time.perf_counter() - t0
print("steps %d in %.1f s = %.2f ms/step; loss first %.5f last %.5f min %.5f max %.5f finite %s corr[-1] %.3f" % (
    n, dt, dt / n * 1e3, losses[0], losses[-1], min(losses), max(losses), bool(np.isfinite(losses).all()), float(corr.max())))
p = eng.get_params(); print("params finite:", all(np.isfinite(a).all() for a in p))
