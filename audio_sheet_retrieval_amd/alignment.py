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
    """ Compute alignment baseline by interpolation (:112-116) """
    return np.linspace(start=0, stop=dists.shape[0] - 1, num=dists.shape[1])


def align_pydtw(engine, img_codes, spec_codes):
    """ DTW alignment (:119-140), "fix path" loop as in the reference """
    _, dists, path = dtw_by_dist_codes(engine, img_codes, spec_codes)
    align_sheet_idxs = []
    for i in range(dists.shape[1]):
        sheet_idx = np.nonzero(path[0] == i)[0][0]
        align_sheet_idxs.append(path[1][sheet_idx])
    return np.array(align_sheet_idxs), dists


def compute_alignment(engine, img_codes, spec_codes, sheet_idxs, spec_idxs, align_by):
    """ Evaluate Alignment (:143-177) """
    if align_by == 'baseline':
        dists = engine.dtw(img_codes, spec_codes)[1]           # only the distance matrix is used here
        aligned_sheet_idxs = align_baseline(dists)
    elif align_by == 'pydtw':
        aligned_sheet_idxs, dists = align_pydtw(engine, img_codes, spec_codes)
    else:
        raise ValueError("align_by must be 'baseline' or 'pydtw'")
    aligned_sheet_idxs = np.round(aligned_sheet_idxs).astype(np.int64)
    aligned_sheet_coords = sheet_idxs[aligned_sheet_idxs]
    filterd_idxs = np.diff(np.concatenate((spec_idxs[0:1] - 1, spec_idxs))) > 0
    f_inter = interp1d(spec_idxs[filterd_idxs], aligned_sheet_coords[filterd_idxs])
    i_inter = np.arange(spec_idxs[0], spec_idxs[-1] + 1, 1)
    a2s_alignment = f_inter(i_inter)
    a2s_mapping = dict(zip(i_inter, a2s_alignment))
    dtw_res = {"dists": dists, "aligned_sheet_idxs": aligned_sheet_idxs, "aligned_sheet_coords": aligned_sheet_coords,
               "i_inter": i_inter, "a2s_alignment": a2s_alignment, "spec_idxs": spec_idxs}
    return a2s_mapping, dtw_res


def estimate_alignment_error(true_coords, true_onsets, a2s_mapping):
    """ Compute alignment error measures (:180-190) """
    pxl_errors = np.zeros(len(true_onsets))
    for j, o in enumerate(true_onsets):
        if o in a2s_mapping:
            pxl_errors[j] = true_coords[j] - a2s_mapping[int(o)]
    return pxl_errors
