import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_secondary as bs
from audio_sheet_retrieval_amd import _lib
from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
for B in (100, 512):
    eng = _lib.Engine(bs.MODEL)
    eng.set_params(synth_data.synth_params(param_shapes(bs.MODEL), seed=1, trained_like=False))
    r = bs.measure_train(eng, B=B)
    print(B, round(r["ms_per_step"], 3), r["loss"])
    eng.close()
