"""per-step wall times of the batch-512 training step (outlier hunt):  python tools/train_series.py [steps]"""
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
sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
x1 = sheet.astype(np.float32) / np.float32(255)
eng.train_begin(B)
d1 = eng.alloc(x1.nbytes).upload(x1); d2 = eng.alloc(spec.nbytes).upload(spec)
loss = ctypes.c_float(); corr = np.empty(32, np.float32)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    eng.sync(); t0 = time.perf_counter()
    eng._check(eng.lib.asr_train_step_dev(eng.ctx, d1.ptr, d2.ptr, B, 0.002, ctypes.byref(loss), corr.ctypes.data))
    eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts[3:])
print("steps %d  median %.2f  min %.2f  max %.2f  >1.5x median: %d" % (len(ts), np.median(ts), ts.min(), ts.max(), int((ts > 1.5 * np.median(ts)).sum())))
