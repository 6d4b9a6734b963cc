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
        # the server loads its data base once and queries it per frame (:496-522, :530-563): the rows' float64 norms,
        # reciprocal norms and the unit-length copy the filter reads are computed here, once (asr_db_create) - a call
        # used to spend an extra pass over the whole pool on them
        self._handle = engine.db_create(self._d_codes.ptr, self.codes.shape[0], dim=32)
        self._scratch = {}                  # device buffers of the query path, grown on demand, kept between calls

    def scratch(self, name, nbytes):
        """a device buffer of at least nbytes that lives as long as the data base (detect_* run per frame)"""
        buf = self._scratch.get(name)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = self._scratch[name] = self.engine.alloc(max(int(nbytes), 4))
        return buf

    def topk_dev(self, q_ptr, n_q, k, idx_ptr, dist_ptr):
        self._handle.topk_dev(q_ptr, n_q, k, idx_ptr, dist_ptr)

    def retrieve(self, queries, k):
        """_retrieve_*_ids_for_* (:530-563) for host codes: (idx (Q,k) int32, dist (Q,k) float64)"""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        n = q.shape[0]
        dq = self.scratch("q", q.nbytes).upload(q)
        di, dd = self.scratch("idx", n * k * 4), self.scratch("dist", n * k * 8)
        self.topk_dev(dq.ptr, n, k, di.ptr, dd.ptr)
        return di.download((n, k), np.int32), dd.download((n, k), np.float64)

    def close(self):
        if getattr(self, "_handle", None) is not None:
            self._handle.close()
            self._handle = None
            if getattr(self.engine, "ctx", None):      # (a closed engine took every device buffer with it)
                for b in list(self._scratch.values()) + [self._d_codes, self._d_ids]:
                    b.free()
            self._scratch = {}

    # persistent device buffers: `with EmbeddingDB(...) as db:` or close(); dropping the object releases them too
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            if getattr(self.engine, "ctx", None):      # after the engine closed, its buffers are gone with the context
                self.close()
        except Exception:
            pass

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
    # buffers that live with the data base: the reference's server answers one request after the other, and five
    # hipMalloc / hipFree pairs per request cost more than the retrieval itself
    d_src = db.scratch("src", long_input.nbytes).upload(long_input)
    d_win = db.scratch("win", n_samples * win_h * win_w * 4)
    d_codes = db.scratch("codes", n_samples * 32 * 4)
    d_idx = db.scratch("idx", n_samples * n_candidates * 4)
    d_dist = db.scratch("dist", n_samples * n_candidates * 8)
    engine.slice_windows_dev(d_src.ptr, rows, T, r0, win_h, win_w, starts, d_win.ptr)
    if view == 2:
        if (engine.cfg.h2, engine.cfg.w2) != (win_h, win_w):
            engine.set_input_size(2, win_h, win_w)
        engine.embed_view2_dev(d_win.ptr, n_samples, d_codes.ptr)
    else:
        from . import _lib
        engine.embed_view1_dev(d_win.ptr, _lib.IN_F32_RAW, n_samples, d_codes.ptr)
    db.topk_dev(d_codes.ptr, n_samples, n_candidates, d_idx.ptr, d_dist.ptr)
    pieces, counts = engine.piece_vote_dev(d_idx.ptr, n_samples * n_candidates, db._d_ids.ptr, len(db),
                                           db.n_pieces, top_k)
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
