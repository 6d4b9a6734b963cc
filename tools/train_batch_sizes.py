"""Probe (GPU box): the plain training update at several batch sizes (ms per update) - what a rank of a data-parallel job
computes per update at batch 512 / world.  Usage: [ASR_LIB_PATH=<variant .so>] python tools/train_batch_sizes.py [B ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_secondary as bs  # noqa: E402
from audio_sheet_retrieval_amd import _lib  # noqa: E402
from audio_sheet_retrieval_amd.utils import synth_data  # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402

ALT = os.environ.get("ASR_LIB_PATH")             # a variant build (tools/build_variant.sh) for same-box A/B runs
for B in [int(a) for a in sys.argv[1:]] or (64, 100, 512):
    eng = _lib.Engine(bs.MODEL, lib=_lib.load_library(ALT) if ALT else None)
    eng.set_params(synth_data.synth_params(param_shapes(bs.MODEL), seed=1, trained_like=False))
    r = bs.measure_train(eng, B=B)
    print("batch %d: %.3f ms per update, loss %.5f" % (B, r["ms_per_step"], r["loss"]))
    eng.close()
