"""TEST INFRASTRUCTURE (see oracle/__init__.py: pinned by reference outputs in tests/golden/reference_golden.npz).

CPU restatement of the alignment path of the reference:
  dtw_by_dist      audio_sheet_retrieval/utils/dtw_by_dist.py:5-34 (+ _traceback :76-91)
  align_baseline   utils/alignment.py:112-116
  align_pydtw      utils/alignment.py:119-140 (the "fix path" loop, literally)
  compute_alignment utils/alignment.py:143-177
Distances: cdist(..., "cosine") = oracle.retrieval.cdist_cosine64 (bit-exact with SciPy).
"""
import numpy as np
from scipy.interpolate import interp1d

from . import retrieval as oret


def _accumulate(cost):
    """acc[i+1, j+1] = cost[i, j] + min(acc[i, j], acc[i, j+1], acc[i+1, j]) with an infinite border except the
    corner (dtw_by_dist.py:17-27); same float64 additions in the same order as the reference's in-place loop."""
    rows, cols = cost.shape
    acc = np.full((rows + 1, cols + 1), np.inf)
    acc[0, 0] = 0.0
    for i in range(rows):
        above, here = acc[i], acc[i + 1]
        for j in range(cols):
            here[j + 1] = cost[i, j] + min(above[j], above[j + 1], here[j])
    return acc


def _walk_back(acc):
    """Optimal path from the last cell to (0, 0); ties prefer the diagonal, then the row step, then the column
    step (np.argmin order in dtw_by_dist.py:76-91)."""
    i, j = acc.shape[0] - 2, acc.shape[1] - 2
    steps = [(i, j)]
    while i > 0 or j > 0:
        move = int(np.argmin((acc[i, j], acc[i, j + 1], acc[i + 1, j])))
        i -= move in (0, 1)
        j -= move in (0, 2)
        steps.append((i, j))
    steps.reverse()
    rows_idx, cols_idx = zip(*steps)
    return np.array(rows_idx), np.array(cols_idx)


def dtw_by_dist(dist):
    """-> (normalised cost, local cost matrix, accumulated cost matrix, path); a wide matrix is processed
    transposed, and the path is swapped when it was NOT transposed (dtw_by_dist.py:13-15, :30-31)."""
    wide = dist.shape[1] > dist.shape[0]
    cost = np.array(dist.T if wide else dist, dtype=np.float64)
    acc = _accumulate(cost)
    p, q = _walk_back(acc)
    total = acc[1:, 1:]
    return total[-1, -1] / (total.shape[0] + total.shape[1]), cost, total, ((p, q) if wide else (q, p))


def align_baseline(dists):
    """straight line from the first to the last sheet position, one value per audio excerpt"""
    return np.linspace(0, dists.shape[0] - 1, dists.shape[1])


def align_pydtw(dists):
    """first path entry of every audio excerpt -> its sheet position (utils/alignment.py:131-138)"""
    path = dtw_by_dist(dists)[3]
    first = [int(np.flatnonzero(path[0] == col)[0]) for col in range(dists.shape[1])]
    return np.asarray(path[1])[first]


def compute_alignment(img_codes, spec_codes, sheet_idxs, spec_idxs, align_by):
    dists = oret.cdist_cosine64(img_codes, spec_codes)
    aligned = align_baseline(dists) if align_by == 'baseline' else align_pydtw(dists)
    aligned = np.round(aligned).astype(np.int64)
    coords = sheet_idxs[aligned]
    filt = np.diff(np.concatenate((spec_idxs[0:1] - 1, spec_idxs))) > 0
    f_inter = interp1d(spec_idxs[filt], coords[filt])
    i_inter = np.arange(spec_idxs[0], spec_idxs[-1] + 1, 1)
    a2s = f_inter(i_inter)
    return dict(zip(i_inter, a2s)), dict(dists=dists, aligned_sheet_idxs=aligned, aligned_sheet_coords=coords,
                                         i_inter=i_inter, a2s_alignment=a2s, spec_idxs=spec_idxs)
