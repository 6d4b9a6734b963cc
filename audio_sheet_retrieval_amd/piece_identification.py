"""Piece identification on top of top-k retrieval - the vote of the reference's server
(audio_sheet_retrieval/audio_sheet_server.py:213-300 detect_score / detect_performance) and its persistent
embedding data base (:496-522 load/save_*_db_file: pickle of [codes, ids, id_to_name, snippets]).

Everything between the long input and the vote result stays on the GPU: window slicing, tower forward, top-k against
the resident data base, vote histogram and selection (csrc/piece_vote_kernels.hip).  SURVEY.md 8f row 1.
"""
from __future__ import annotations

import pickle

import numpy as np


class EmbeddingDB(object):
    """codes (N,32) float32, ids (N,) piece index per code, id_to_name {index: name}, optional snippets -
    the four objects the reference pickles (:498-510), plus the device-resident copies used for retrieval."""

    def __init__(self, engine, codes, ids, id_to_name, snippets=None):
        self.engine = engine
        self.codes = np.ascontiguousarray(codes, dtype=np.float32)
        self.ids = np.ascontiguousarray(ids, dtype=np.int32)
        if self.codes.ndim != 2 or self.codes.shape[1] != 32 or self.ids.shape != (self.codes.shape[0],):
            raise ValueError("codes must be (N,32) and ids (N,), got %r and %r" % (self.codes.shape, self.ids.shape))
        self.id_to_name = dict(id_to_name)
        self.snippets = snippets
        self.n_pieces = int(self.ids.max()) + 1 if self.ids.size else 1
        self._d_codes = engine.alloc(max(self.codes.nbytes, 4)).upload(self.codes)
        self._d_ids = engine.alloc(max(self.ids.nbytes, 4)).upload(self.ids)

    @classmethod
    def load(cls, engine, path):
        with open(path, "rb") as fp:
            try:
                codes, ids, id_to_name, snippets = pickle.load(fp)
            except UnicodeDecodeError:          # written by the Python-2 reference
                fp.seek(0)
                codes, ids, id_to_name, snippets = pickle.load(fp, encoding="latin1")
        return cls(engine, codes, ids, id_to_name, snippets)

    def save(self, path):
        with open(path, "wb") as fp:
            pickle.dump([self.codes, self.ids.astype(np.int64), self.id_to_name, self.snippets], fp, protocol=2)

    def __len__(self):
        return self.codes.shape[0]


def _detect(engine, db, long_input, view, win_shape, r0, top_k, n_candidates, n_samples):
    long_input = np.ascontiguousarray(long_input, dtype=np.float32)
    rows, T = long_input.shape
    win_h, win_w = win_shape
    if T < win_w:
        raise ValueError("input has %d columns, a window needs %d" % (T, win_w))
    starts = np.linspace(start=0, stop=T - win_w, num=n_samples).astype(np.int32)        # :217-218
    d_src = engine.alloc(long_input.nbytes).upload(long_input)
    d_win = engine.alloc(n_samples * win_h * win_w * 4)
    d_codes = engine.alloc(n_samples * 32 * 4)
    d_idx = engine.alloc(n_samples * n_candidates * 4)
    d_dist = engine.alloc(n_samples * n_candidates * 8)
    try:
        engine.slice_windows_dev(d_src.ptr, rows, T, r0, win_h, win_w, starts, d_win.ptr)
        if view == 2:
            if (engine.cfg.h2, engine.cfg.w2) != (win_h, win_w):
                engine.set_input_size(2, win_h, win_w)
            engine.embed_view2_dev(d_win.ptr, n_samples, d_codes.ptr)
        else:
            from . import _lib
            engine.embed_view1_dev(d_win.ptr, _lib.IN_F32_RAW, n_samples, d_codes.ptr)
        engine.topk_dev(db._d_codes.ptr, len(db), d_codes.ptr, n_samples, n_candidates, d_idx.ptr, d_dist.ptr)
        pieces, counts = engine.piece_vote_dev(d_idx.ptr, n_samples * n_candidates, db._d_ids.ptr, len(db),
                                               db.n_pieces, top_k)
    finally:
        for b in (d_src, d_win, d_codes, d_idx, d_dist):
            b.free()
    names = [db.id_to_name[int(p)] for p in pieces]
    votes = counts.astype(np.float64) / counts.sum() if counts.size else counts.astype(np.float64)
    return names, votes, pieces, counts


def detect_score(engine, sheet_db, spectrogram, top_k=1, n_candidates=1, n_samples=100, spec_shape=(92, 42)):
    """detect piece from audio (:213-251): `spectrogram` (bins, frames) float32 -> (piece names, normalised votes)."""
    names, votes, _, _ = _detect(engine, sheet_db, spectrogram, 2, spec_shape, 0, top_k, n_candidates, n_samples)
    return names, votes


def detect_performance(engine, audio_db, sheet, top_k=1, n_candidates=1, n_samples=100, sheet_shape=(160, 200)):
    """detect performance from an unrolled score strip (:253-300): `sheet` (rows, columns) with the reference's 0..255
    value range (model.prepare divides by 255 - folded into the first kernel); the central `sheet_shape[0]` rows are
    used (:269-271)."""
    sheet = np.asarray(sheet)
    r0 = sheet.shape[0] // 2 - sheet_shape[0] // 2
    names, votes, _, _ = _detect(engine, audio_db, sheet, 1, sheet_shape, r0, top_k, n_candidates, n_samples)
    return names, votes
