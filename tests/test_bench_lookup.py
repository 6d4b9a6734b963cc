"""CPU: bench.py's counter lookup (VERDICT r4 Weak #4).  The roofline block quotes HBM traffic and MFMA-busy figures from a
committed rocprofv3 PMC profile; which layers a kernel symbol serves is the tuner's choice, so a figure is only used when
it describes the launches that ran: same symbol AND same layers, else re-assembled per layer under the same symbol, else
withheld with the reason."""
import json
import os

import bench

SYM_A = "void asr::conv3x3_winog<24, 24, true, 2, 8, 2, false>(asr::WinoGArgs)"
SYM_B = "void asr::conv3x3_winog<24, 24, true, 2, 4, 2, false>(asr::WinoGArgs)"
BY_SYMBOL = {SYM_A: {"hbm_bytes_per_launch": 123.8e6, "layers": ["conv4_v2"]},
             SYM_B: {"hbm_bytes_per_launch": 990e6, "layers": ["conv4_v1"]}}
BY_LAYER = {"conv4_v1": {"hbm_bytes_per_launch": 990e6, "symbol": SYM_B},
            "conv4_v2": {"hbm_bytes_per_launch": 123.8e6, "symbol": SYM_A},
            "conv2_v1": {"hbm_bytes_per_launch": 1900e6, "symbol": "other"}}


def test_same_symbol_and_layers_is_used_by_symbol():
    rec, how = bench._lookup_counters(BY_SYMBOL, BY_LAYER, SYM_A, ["conv4_v2"], ("hbm_bytes_per_launch",))
    assert rec == {"hbm_bytes_per_launch": 123.8e6} and how.startswith("by symbol")


def test_round4_case_is_withheld_not_misattributed():
    """round 4's driver box: one build served conv4 of BOTH towers, the profile had it on the spectrogram tower only - the
    by-symbol figure (123.8 MB) described no launch of that run; conv4_v1 ran another symbol in the profile, so the
    per-layer route does not apply either: nothing is reported, and the reason says what differs"""
    rec, how = bench._lookup_counters(BY_SYMBOL, BY_LAYER, SYM_A, ["conv4_v1", "conv4_v2"], ("hbm_bytes_per_launch",))
    assert rec is None and "conv4_v1" in how and "profile" in how


def test_per_layer_reassembly_needs_the_same_symbol_on_every_layer():
    by_layer = dict(BY_LAYER, conv4_v1={"hbm_bytes_per_launch": 950e6, "symbol": SYM_A})
    rec, how = bench._lookup_counters(BY_SYMBOL, by_layer, SYM_A, ["conv4_v1", "conv4_v2"], ("hbm_bytes_per_launch",))
    assert abs(rec["hbm_bytes_per_launch"] - (950e6 + 123.8e6) / 2) < 1 and "mean over the live layers" in how
    rec, how = bench._lookup_counters(BY_SYMBOL, BY_LAYER, "void asr::never_profiled()", ["conv2_v1"], ("hbm_bytes_per_launch",))
    assert rec is None and "did not run" in how
    rec, how = bench._lookup_counters(None, None, SYM_A, ["conv4_v2"], ("hbm_bytes_per_launch",))
    assert rec is None and "no committed" in how


def test_committed_profile_and_tune_cache_belong_together():
    """profiles/<round>_*_by_symbol.json name their layers and algorithmic bytes, carry the per-layer tables, and the tune
    cache of that run is committed next to them (bench.py runs those schedules by default)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rnd = bench.PROFILE_ROUND
    for kind in ("hbm_traffic", "mfma_busy"):
        with open(os.path.join(root, "profiles", "%s_%s_by_symbol.json" % (rnd, kind))) as fp:
            doc = json.load(fp)
        assert doc["layers"] and all("symbol" in v for v in doc["layers"].values())
        for sym, rec in doc["kernels"].items():
            assert rec["layers"], sym
            for lab in rec["layers"]:
                assert doc["layers"][lab]["symbol"] == sym
    cache = os.path.join(root, "profiles", "%s_tune_cache.txt" % rnd)
    assert os.path.getsize(cache) > 0
    tmp = bench._committed_tune_cache()
    assert tmp and open(tmp).read() == open(cache).read() and os.path.dirname(tmp) != os.path.dirname(cache)
    # the dominant symbols' traffic is within the accepted band of their algorithmic bytes
    with open(os.path.join(root, "profiles", "%s_hbm_traffic_by_symbol.json" % rnd)) as fp:
        doc = json.load(fp)
    big = [r for r in doc["kernels"].values() if (r.get("algorithmic_bytes_per_launch") or 0) > 2e8]
    assert big and all(0.8 <= r["hbm_bytes_per_launch"] / r["algorithmic_bytes_per_launch"] <= 2.0 for r in big)


def test_committed_tune_cache_was_written_by_this_build():
    """the lines of a tune cache carry a tag derived from asr_version() (source + flags hash); lines of another build are
    ignored by the library - bench.py would then time the schedules on its box and the counter lookup could be withheld.
    Any change to csrc/ therefore needs `bash tools/profile_round.sh <round>` again: this test says so at once."""
    import ctypes
    from audio_sheet_retrieval_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(_lib.LIB_PATH):
        import pytest
        pytest.skip("libasr_hip.so not built")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.asr_version.restype = ctypes.c_char_p
    h = 2166136261
    for ch in lib.asr_version():
        h = ((h ^ ch) * 16777619) & 0xffffffff
    tag = str(h & 0x7fffffff)
    lines = open(os.path.join(root, "profiles", "%s_tune_cache.txt" % bench.PROFILE_ROUND)).read().splitlines()
    assert lines and all(ln.split()[1] == tag for ln in lines), (tag, lines[0])
