#!/usr/bin/env python
"""Probe: how much do two independent training steps (two engines, two host threads, batch B/2 each) overlap on one
GPU, against one engine at batch B?  Prints one JSON line.  (An experiment behind DESIGN.md's training notes - the step
is a chain of kernels that are either HBM-bound or MFMA-bound; this measures what co-scheduling them can give.)"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_sheet_retrieval_amd import _lib  # noqa: E402
from audio_sheet_retrieval_amd.utils import synth_data  # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402

MODEL = "mutopia_ccal_cont"


def engine(B):
    eng = _lib.Engine(MODEL)
    eng.set_params(synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=False))
    sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
    x1 = sheet.astype(np.float32) / np.float32(255)
    eng.train_begin(B)
    d1 = eng.alloc(x1.nbytes).upload(x1)
    d2 = eng.alloc(spec.nbytes).upload(spec)
    loss = ctypes.c_float()
    corr = np.empty(32, np.float32)

    def step():
        eng._check(eng.lib.asr_train_step_dev(eng.ctx, d1.ptr, d2.ptr, B, 0.002, ctypes.byref(loss), corr.ctypes.data))
    return eng, step, (d1, d2)


def run(steps, K):
    for s in steps:
        for _ in range(3):
            s()
    t0 = time.perf_counter()
    if len(steps) == 1:
        for _ in range(K):
            steps[0]()
    else:
        th = [threading.Thread(target=lambda s=s: [s() for _ in range(K)]) for s in steps]
        for t in th:
            t.start()
        for t in th:
            t.join()
    return (time.perf_counter() - t0) / K * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    K = 20
    e1, s1, k1 = engine(B)
    t_full = min(run([s1], K) for _ in range(3))
    e1.close()
    ea, sa, ka = engine(B // 2)
    eb, sb, kb = engine(B // 2)
    t_half = min(run([sa], K) for _ in range(3))
    t_pair = min(run([sa, sb], K) for _ in range(3))
    print(json.dumps({"B": B, "ms_one_engine_B": round(t_full, 3), "ms_one_engine_half": round(t_half, 3),
                      "ms_two_engines_half_each_concurrent": round(t_pair, 3)}))


if __name__ == "__main__":
    main()
