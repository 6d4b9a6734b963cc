import sys, ctypes, numpy as np
sys.path.insert(0, '.')
from audio_sheet_retrieval_amd import _lib
from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
M = "mutopia_ccal_cont"
params = synth_data.synth_params(param_shapes(M), seed=1, trained_like=False)
sheet, spec = synth_data.synth_pairs(np.arange(64), seed=23)
x1 = sheet.astype(np.float32) / np.float32(255)
vals = []
for it in range(12):
    eng = _lib.Engine(M, max_chunk=64)
    eng.set_params(params)
    eng.embed_view1(x1, prepared=True); eng.embed_view2(spec)
    eng.train_begin(64)
    for _ in range(3): eng.train_step(x1, spec, lr=0.002)
    eng.train_end()
    eng.embed_view1(x1, prepared=True)
    eng.close()
    vals.append(free_mb())
print("free MB after each cycle:", ' '.join('%.0f' % v for v in vals))
print("drift MB (last - second):", vals[-1] - vals[1])
