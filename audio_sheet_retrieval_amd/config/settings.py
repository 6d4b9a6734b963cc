"""Paths (reference: audio_sheet_retrieval/config/settings.py:5-18, which
hard-codes the authors' home directories per hostname).  Here they come from the
environment, with the reference's layout below them."""
import os

# experiment root: <EXP_ROOT>/<model>/params_<split>_<config>.pkl (run_train.py:87-91)
EXP_ROOT = os.environ.get("ASR_EXP_ROOT", os.path.join(os.path.expanduser("~"), "experiments", "audio_sheet_retrieval"))
# MSMD data set root (config/settings.py:6); only needed for --data mutopia
DATA_ROOT_MSMD = os.environ.get("ASR_DATA_ROOT_MSMD", "/data/msmd_aug/")
