"""Audio-to-sheet alignment on top of the embedding space: distance matrix + DTW on the GPU.

Mirror of audio_sheet_retrieval/utils/alignment.py:112-186 (align_baseline, align_pydtw, compute_alignment,
estimate_alignment_error) with utils/dtw_by_dist.py:dtw_by_dist behind it.  The cosine distance matrix, the accumulated
cost (anti-diagonal wavefront) and the traceback run in the library (asr_dtw_dev, float64, bit-exact with the
reference's NumPy arithmetic); the O(n) path post-processing and the interpolation stay on the host as in the reference.
"""
from __future__ import print_function

import numpy as np
from scipy.interpolate import interp1d


def dtw_by_dist_codes(engine, img_codes, spec_codes):
    """dtw_by_dist(cdist(img_codes, spec_codes, "cosine")) (utils/dtw_by_dist.py:5-34) -> (min_dist, dists, path):
    the reference transposes a wide matrix and swaps the returned path when it did NOT transpose (:13-15, :30-31)."""
    n_r, n_c = len(img_codes), len(spec_codes)
    if n_c > n_r:                                        # transposed = True: rows = spec codes
        md, d, p, q = engine.dtw(spec_codes, img_codes)
        return md, d.T, (p, q)
    md, d, p, q = engine.dtw(img_codes, spec_codes)
    return md, d, (q, p)


def align_baseline(dists):
    """straight line from the first to the last sheet position, one value per audio excerpt (:112-116)"""
    return np.linspace(0, dists.shape[0] - 1, dists.shape[1])


def align_pydtw(engine, img_codes, spec_codes):
    """DTW alignment (:119-140): the sheet position of the first path entry of every audio excerpt"""
    _, dists, path = dtw_by_dist_codes(engine, img_codes, spec_codes)
    first = [int(np.flatnonzero(path[0] == col)[0]) for col in range(dists.shape[1])]
    return np.asarray(path[1])[first], dists


def compute_alignment(engine, img_codes, spec_codes, sheet_idxs, spec_idxs, align_by):
    """Audio frame -> sheet x-coordinate mapping (:143-177).  Returns (mapping dict, dict of intermediate results
    under the reference's keys)."""
    if align_by == "baseline":
        dists = engine.dtw(img_codes, spec_codes)[1]           # only the distance matrix is used here
        positions = align_baseline(dists)
    elif align_by == "pydtw":
        positions, dists = align_pydtw(engine, img_codes, spec_codes)
    else:
        raise ValueError("align_by must be 'baseline' or 'pydtw'")
    positions = np.round(positions).astype(np.int64)
    coords = sheet_idxs[positions]
    # excerpts whose frame index did not advance carry no new information for the interpolation (:163)
    advancing = np.diff(np.concatenate((spec_idxs[:1] - 1, spec_idxs))) > 0
    frames = np.arange(spec_idxs[0], spec_idxs[-1] + 1)
    on_sheet = interp1d(spec_idxs[advancing], coords[advancing])(frames)
    details = {"dists": dists, "aligned_sheet_idxs": positions, "aligned_sheet_coords": coords,
               "i_inter": frames, "a2s_alignment": on_sheet, "spec_idxs": spec_idxs}
    return dict(zip(frames, on_sheet)), details


def estimate_alignment_error(true_coords, true_onsets, a2s_mapping):
    """pixel error per annotated onset; onsets outside the mapped frame range count as 0 (:180-190)"""
    errors = np.zeros(len(true_onsets))
    for n, onset in enumerate(true_onsets):
        if onset in a2s_mapping:
            errors[n] = true_coords[n] - a2s_mapping[int(onset)]
    return errors
