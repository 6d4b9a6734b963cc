"""TEST INFRASTRUCTURE (see oracle/__init__.py).  Slicing, retrieval and vote are pinned by a run of the reference's own
server methods (tests/golden/reference_golden.npz, keys vote_*).

CPU restatement of the piece-identification vote of the reference's server
(audio_sheet_retrieval/audio_sheet_server.py):
  window_starts      :217-218 / :273-274  np.linspace(0, T - w, n_samples).astype(int)
  slice_windows      :221-223 / :276-281  excerpts of one long spectrogram / unrolled sheet
  retrieve_ids       :530-563             cdist(db, q, "cosine") -> argsort[:n_candidates] -> ids
  vote               :228-244             np.unique(return_counts) -> argsort(counts)[::-1][:top_k] -> normalised votes
Tie order: the reference's argsort (quicksort) leaves equal counts undefined; here a STABLE ascending sort is
reversed, i.e. equal votes list the larger piece id first - the device kernel does the same.
"""
import numpy as np

from . import retrieval as oret


def window_starts(T, win_w, n_samples=100):
    return np.linspace(start=0, stop=T - win_w, num=n_samples).astype(np.int64)


def slice_windows(src, r0, win_h, win_w, starts):
    out = np.zeros((len(starts), 1, win_h, win_w), dtype=np.float32)
    for i, idx in enumerate(starts):
        out[i, 0] = src[r0:r0 + win_h, idx:idx + win_w]
    return out


def retrieve_ids(db_codes, db_ids, query_codes, n_candidates):
    idx, _ = oret.topk(db_codes, query_codes, n_candidates)          # stable argsort of the float64 distances
    valid = idx >= 0
    return db_ids[idx[valid]], idx


def vote(all_piece_ids, top_k):
    unique, counts = np.unique(np.asarray(all_piece_ids, dtype=np.int64), return_counts=True)
    order = np.argsort(counts, kind="stable")[::-1][:top_k]
    pieces, c = unique[order], counts[order]
    votes = c.astype(np.float64) / c.sum() if c.size else c.astype(np.float64)
    return pieces.astype(np.int32), c.astype(np.int32), votes
