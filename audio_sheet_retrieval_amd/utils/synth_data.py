"""Synthetic MSMD-shaped data pools (stand-in for utils/data_pools.py + the absent
`msmd` data set).

The reference's AudioScoreRetrievalPool.__getitem__ yields
    [sheet (n,1,160,200) float32 holding 0..255, spec (n,1,92,42) float32 >= 0]
(audio_sheet_retrieval/utils/data_pools.py:203-228; geometry from
exp_configs/*.yaml:1-4).  This module produces arrays of exactly that contract
from a counter-based generator (splitmix64 of seed/stream/global index), so the
same sample index yields bit-identical data on every host, rank and shard
without shipping files.

Pair structure: sample i owns a 64-bit key that places K_NOTES "notes"; each
note is a dark 6x8 blob in the sheet at (row, col) and a 3x3 energy bump in the
spectrogram at (bin, frame) derived from the same draw - enough cross-modal
signal for a briefly trained model to beat chance.
"""
from __future__ import annotations

import numpy as np

SHEET_CONTEXT = 200     # exp_configs/mutopia_full_aug.yaml:1
SYSTEM_HEIGHT = 160     # :2
SPEC_CONTEXT = 42       # :3
SPEC_BINS = 92          # :4
K_NOTES = 8

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_STREAM_SHEET = np.uint64(0x1000000000000000)
_STREAM_SPEC = np.uint64(0x2000000000000000)
_STREAM_NOTE = np.uint64(0x3000000000000000)
_STREAM_WEIGHT = np.uint64(0x4000000000000000)
_STAFF_ROWS = np.array([50, 55, 60, 65, 70, 100, 105, 110, 115, 120])


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _draw(seed, stream, sample_idx, n_per_sample):
    """u64 draws, shape (len(sample_idx), n_per_sample); counter =
    hash(seed, stream, sample) + element index."""
    sample_idx = np.asarray(sample_idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = splitmix64(splitmix64(np.uint64(seed) ^ stream) ^ sample_idx)
        ctr = base[:, None] + np.arange(n_per_sample, dtype=np.uint64)[None, :]
    return splitmix64(ctr)


def _unit(u64):
    """top 24 bits -> float32 in [0,1)."""
    return (u64 >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def synth_pairs(indices, seed=23):
    """(sheet uint8 (n,1,160,200), spec float32 (n,1,92,42)) for the given
    global sample indices."""
    idx = np.asarray(indices, dtype=np.int64).ravel()
    n = idx.shape[0]
    h, w, hb, wf = SYSTEM_HEIGHT, SHEET_CONTEXT, SPEC_BINS, SPEC_CONTEXT
    # --- sheet: mostly white, sparse dark speckle, staff lines
    u = _draw(seed, _STREAM_SHEET, idx, h * w)
    dark = (u & np.uint64(0xFF)) >= np.uint64(230)                   # P ~ 0.1
    val = ((u >> np.uint64(8)) & np.uint64(0x7F)).astype(np.uint8)   # 0..127
    sheet = np.where(dark, val, np.uint8(255)).astype(np.uint8).reshape(n, h, w)
    sheet[:, _STAFF_ROWS, :] = 0
    # --- spec: long-tailed non-negative noise
    us = _unit(_draw(seed, _STREAM_SPEC, idx, hb * wf))
    spec = (np.float32(3.0) * us * us).reshape(n, hb, wf).astype(np.float32)
    # --- notes shared by both views
    un = _draw(seed, _STREAM_NOTE, idx, K_NOTES)
    rows = (un % np.uint64(h - 6)).astype(np.int64)
    cols = ((un >> np.uint64(16)) % np.uint64(w - 8)).astype(np.int64)
    bins = (rows * (hb - 3)) // (h - 6)        # pitch axis: sheet row <-> spec bin
    frames = (cols * (wf - 3)) // (w - 8)      # time axis: sheet col <-> spec frame
    ar = np.arange(n)
    for k in range(K_NOTES):
        for dy in range(6):
            for dx in range(8):
                sheet[ar, rows[:, k] + dy, cols[:, k] + dx] = 0
        for dy in range(3):
            for dx in range(3):
                spec[ar, bins[:, k] + dy, frames[:, k] + dx] += np.float32(2.0)
    return sheet[:, None, :, :], spec[:, None, :, :]


class SyntheticRetrievalPool(object):
    """Drop-in for AudioScoreRetrievalPool (utils/data_pools.py:35-228): `.shape`,
    `__getitem__(int | slice | ndarray) -> [sheet f32 0..255, spec f32]`,
    `reset_batch_generator()` (reshuffles when shuffle=True)."""

    def __init__(self, n_samples, seed=23, shuffle=False, first_index=0):
        self.n_samples = int(n_samples)
        self.seed = int(seed)
        self.shuffle = shuffle
        self.first_index = int(first_index)
        self.shape = [self.n_samples]
        self.sheet_dim = [SYSTEM_HEIGHT, SHEET_CONTEXT]
        self.spec_dim = [SPEC_BINS, SPEC_CONTEXT]
        self._epoch = 0
        self.train_entities = np.arange(self.n_samples, dtype=np.int64)
        self.reset_batch_generator()

    def reset_batch_generator(self, indices=None):
        """data_pools.py:86-125: rebuild (and shuffle) the entity list."""
        self.train_entities = np.arange(self.n_samples, dtype=np.int64)
        if self.shuffle:
            # deterministic permutation per reshuffle: sort by a keyed hash
            keys = splitmix64(self.train_entities.astype(np.uint64)
                              ^ splitmix64(np.uint64(self.seed + 7919 * (self._epoch + 1))))
            self.train_entities = self.train_entities[np.argsort(keys, kind="stable")]
            self._epoch += 1

    def _resolve(self, key):
        if isinstance(key, (int, np.integer)):
            key = slice(int(key), int(key) + 1)
        return self.train_entities[key] + self.first_index

    def __getitem__(self, key):
        sheet_u8, spec = synth_pairs(self._resolve(key), self.seed)
        return [sheet_u8.astype(np.float32), spec]

    def get_u8(self, key):
        """Same samples with the sheet kept as uint8 (what the servers pass,
        audio_sheet_server.py:331,472) - 4x smaller host->device transfers."""
        return list(synth_pairs(self._resolve(key), self.seed))


def load_synthetic_retrieval(n_train=10000, n_valid=1000, n_test=2000, seed=23):
    """Stand-in for utils/mutopia_data.load_audio_score_retrieval (:47-98):
    dict(train=, valid=, test=, train_tag=) of pools with disjoint index ranges."""
    return dict(
        train=SyntheticRetrievalPool(n_train, seed, shuffle=True, first_index=0),
        valid=SyntheticRetrievalPool(n_valid, seed, shuffle=False, first_index=1 << 32),
        test=SyntheticRetrievalPool(n_test, seed, shuffle=False, first_index=2 << 32),
        train_tag="")


def synth_params(shapes, seed=1, trained_like=False):
    """Flat parameter list in the reference's pickle order
    drawn from the counter generator: W ~ HeUniform (A.1); BN beta 0, gamma 1,
    mean 0, inv_std 1 (Lasagne defaults) - or, with trained_like=True, mildly
    perturbed BN statistics and a random orthogonal-ish CCA projection so that
    every term of the deterministic path is exercised."""
    out = []
    n_tower = 90
    for i, shp in enumerate(shapes):
        cnt = int(np.prod(shp))
        u = _unit(_draw(seed, _STREAM_WEIGHT, [i], cnt))[0].reshape(shp)
        if i < n_tower:
            kind = i % 5
            if kind == 0:
                lim = np.float32(np.sqrt(3.0 / (shp[1] * shp[2] * shp[3])))
                a = (u * 2 - 1) * lim
            elif not trained_like:
                a = np.zeros(shp) if kind in (1, 3) else np.ones(shp)
            elif kind == 1:
                a = (u - 0.5) * 0.4                  # beta
            elif kind == 2:
                a = 0.6 + 0.8 * u                    # gamma
            elif kind == 3:
                a = (u - 0.5) * 0.2                  # running mean
            else:
                a = 0.7 + 1.5 * u                    # running inv_std
        else:
            j = i - n_tower
            if not trained_like:
                a = np.zeros(shp)
            elif j in (0, 1):                        # U, V: well-conditioned, not symmetric
                a = (u - 0.5) * 0.6 + np.eye(shp[0]) * 1.5
            elif j in (2, 3):                        # mean1, mean2
                a = (u - 0.5) * 0.1
            else:                                    # S12, S11, S22 (not used in eval)
                a = (u - 0.5) * 0.01 + (np.eye(shp[0]) if j > 4 else 0)
        out.append(np.ascontiguousarray(a, dtype=np.float32))
    return out
