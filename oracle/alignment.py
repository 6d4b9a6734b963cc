"""TEST INFRASTRUCTURE (see oracle/__init__.py: parity unpinned by the reference).

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


def _traceback(D):
    i, j = np.array(D.shape) - 2
    p, q = [i], [j]
    while (i > 0) or (j > 0):
        tb = np.argmin((D[i, j], D[i, j + 1], D[i + 1, j]))
        if tb == 0:
            i -= 1
            j -= 1
        elif tb == 1:
            i -= 1
        else:
            j -= 1
        p.insert(0, i)
        q.insert(0, j)
    return np.array(p), np.array(q)


def dtw_by_dist(dist):
    transposed = False
    if dist.shape[1] > dist.shape[0]:
        dist = dist.T
        transposed = True
    r, c = dist.shape
    D0 = np.zeros((r + 1, c + 1))
    D0[0, 1:] = np.inf
    D0[1:, 0] = np.inf
    D0[1:, 1:] = dist
    D1 = D0[1:, 1:]
    C = D1.copy()
    for i in range(r):
        for j in range(c):
            D1[i, j] += min(D0[i, j], D0[i, j + 1], D0[i + 1, j])
    path = _traceback(D0)
    if not transposed:
        path = (path[1], path[0])
    return D1[-1, -1] / sum(D1.shape), C, D1, path


def align_baseline(dists):
    return np.linspace(start=0, stop=dists.shape[0] - 1, num=dists.shape[1])


def align_pydtw(dists):
    _, _, _, path = dtw_by_dist(dists)
    align_sheet_idxs = []
    for i in range(dists.shape[1]):
        sheet_idx = np.nonzero(path[0] == i)[0][0]
        align_sheet_idxs.append(path[1][sheet_idx])
    return np.array(align_sheet_idxs)


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
