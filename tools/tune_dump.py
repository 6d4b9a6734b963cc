"""Print the autotuner's timing table (ASR_DEBUG) and its cross-check of every candidate (ASR_TUNE_VERIFY) for one model.
usage: python tools/tune_dump.py [model] [chunk]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("ASR_DEBUG", "1")
os.environ.setdefault("ASR_TUNE_VERIFY", "1")
import numpy as np                                                    # noqa: E402
from audio_sheet_retrieval_amd import _lib                            # noqa: E402
from audio_sheet_retrieval_amd.utils import synth_data                # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "mutopia_ccal_cont"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = _lib.Engine(model, max_chunk=chunk)
eng.set_params(synth_data.synth_params(param_shapes(model), seed=1, trained_like=True))
sheet, spec = synth_data.synth_pairs(np.arange(chunk), seed=23)
lv1 = eng.embed_view1(sheet, prepared=False)
lv2 = eng.embed_view2(spec)
print("tune report:", eng.tune_report())
from oracle import network as onet                                   # noqa: E402  (checker only)
params = synth_data.synth_params(param_shapes(model), seed=1, trained_like=True)
k = min(chunk, 48)
r1, r2 = onet.compute_output(onet.prepare(sheet[:k], model), spec[:k], params)
print("embedding difference to the CPU oracle: view1 %.3e view2 %.3e" % (np.abs(lv1[:k] - r1).max(), np.abs(lv2[:k] - r2).max()))
