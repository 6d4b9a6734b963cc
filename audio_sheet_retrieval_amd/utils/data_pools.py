"""AudioScoreRetrievalPool with the batch assembly on the GPU.

Mirror of audio_sheet_retrieval/utils/data_pools.py:36-228: the unrolled score images and the spectrograms of all pieces
stay resident on the device; `pool[key]` draws the augmentation random numbers on the host in the reference's order
(NumPy global RNG: sheet_scaling -> system_translation per image, then onset_translation -> spec_padding per
excerpt, sample after sample), reduces every sample to nine numbers and lets one gather kernel per view cut, rescale
(nearest neighbour) and pad the windows (csrc/piece_vote_kernels.hip: gather_windows_kernel).  `pool.get_device(key)`
returns device buffers that feed asr_train_step_dev / asr_embed_view*_dev directly (sheets un-normalised like the
reference's pool: model.prepare / ASR_IN_F32_RAW divides by 255); `pool[key]` downloads them (reference behaviour).

MSMD loading (msmd package, absent) is out of scope: the pool takes the arrays the reference's loader would pass.
"""
from __future__ import print_function

import numpy as np

SHEET_CONTEXT = 200
SYSTEM_HEIGHT = 160
SPEC_CONTEXT = 42
SPEC_BINS = 92

NO_AUGMENT = {"system_translation": 0, "sheet_scaling": [1.00, 1.00], "onset_translation": 0, "spec_padding": 0,
              "interpolate": -1, "synths": ["ElectricPiano"], "tempo_range": [1.00, 1.00]}
AUGMENT = dict(NO_AUGMENT)


class AudioScoreRetrievalPool(object):

    def __init__(self, engine, images, specs, o2c_maps, spec_context=SPEC_CONTEXT, sheet_context=SHEET_CONTEXT,
                 staff_height=SYSTEM_HEIGHT, data_augmentation=None, shuffle=True):
        self.engine = engine
        self.images = [np.ascontiguousarray(im, dtype=np.float32) for im in images]
        self.specs = [[np.ascontiguousarray(s, dtype=np.float32) for s in sp] for sp in specs]
        self.o2c_maps = o2c_maps
        self.spec_context, self.sheet_context, self.staff_height = spec_context, sheet_context, staff_height
        self.data_augmentation = dict(NO_AUGMENT) if data_augmentation is None else data_augmentation
        self.shuffle = shuffle
        self.sheet_dim = [self.staff_height, self.sheet_context]
        self.spec_dim = [self.specs[0][0].shape[0], self.spec_context]
        if self.data_augmentation['interpolate'] > 0:
            self.interpolate()
        self.prepare_train_entities()
        if self.shuffle:
            self.reset_batch_generator()
        # ---- resident pool: all strips in one float32 buffer per view
        self._img_off, self._spec_off = [], []
        off = 0
        for im in self.images:
            self._img_off.append(off)
            off += im.size
        self._img_floats = off
        off = 0
        for sp in self.specs:
            offs = []
            for s in sp:
                offs.append(off)
                off += s.size
            self._spec_off.append(offs)
        self._spec_floats = off
        self._d_img = engine.alloc(max(4, self._img_floats * 4)).upload(
            np.concatenate([im.ravel() for im in self.images]))
        self._d_spec = engine.alloc(max(4, self._spec_floats * 4)).upload(
            np.concatenate([s.ravel() for sp in self.specs for s in sp]))

    def interpolate(self):
        """Densify the onset -> x-coordinate maps to one entry every `interpolate` frames (:61-82): linear
        interpolation between the annotated onsets, truncated to integers like the reference's astype."""
        from scipy.interpolate import interp1d
        step = self.data_augmentation['interpolate']
        for per_piece in self.o2c_maps:
            for k, o2c in enumerate(per_piece):
                frames = np.arange(o2c[0, 0], o2c[-1, 0] + 1, step)
                xs = interp1d(o2c[:, 0], o2c[:, 1])(frames)
                per_piece[k] = np.stack((frames, xs), axis=1).astype(np.int64)

    def prepare_train_entities(self):
        """All (piece, performance, onset) triples whose excerpt and snippet windows lie inside their arrays
        (:84-117).  The snippet's right edge is tested as `spectrogram window start + sheet_context`, which is what
        the reference's line 109 computes."""
        half_spec, half_sheet = self.spec_context // 2, self.sheet_context // 2
        found = []
        for piece, sheet in enumerate(self.images):
            for perf, spec in enumerate(self.specs[piece]):
                o2c = np.asarray(self.o2c_maps[piece][perf])
                if len(o2c) == 0:
                    continue
                frame0 = o2c[:, 0] - half_spec
                keep = ((frame0 >= 0) & (frame0 + self.spec_context < spec.shape[1]) &
                        (o2c[:, 1] - half_sheet >= 0) & (frame0 + self.sheet_context < sheet.shape[1]))
                found.extend((piece, perf, int(n)) for n in np.flatnonzero(keep))
        self.train_entities = np.asarray(found, dtype=np.int64).reshape(-1, 3)
        self.shape = [self.train_entities.shape[0]]

    def reset_batch_generator(self):
        self.train_entities = self.train_entities[np.random.permutation(self.shape[0])]

    # ---- one sample -> nine numbers (random draws in the reference's order) -------------------------------------
    def _image_desc(self, i_sheet, i_spec, i_onset):
        sheet = self.images[i_sheet]
        Hs, Ws = sheet.shape
        target_coord = int(self.o2c_maps[i_sheet][i_spec][i_onset][1])
        c0 = max(0, target_coord - 2 * self.sheet_context)                      # :137-139
        c1 = min(c0 + 4 * self.sheet_context, Ws)
        c0 = max(0, c1 - 4 * self.sheet_context)
        Wc = c1 - c0
        new_w, new_h = Wc, Hs
        sx = sy = 1.0
        if self.data_augmentation['sheet_scaling']:                             # :142-147
            sc = self.data_augmentation['sheet_scaling']
            scale = (sc[1] - sc[0]) * np.random.random_sample() + sc[0]
            new_w, new_h = int(Wc * scale), int(Hs * scale)
            sx, sy = 1.0 / (float(new_w) / Wc), 1.0 / (float(new_h) / Hs)      # cv2 resizeNN: ifx = 1 / fx
        x = new_w // 2                                                           # :150-157
        x0 = max(x - self.sheet_context // 2, 0)
        x1 = int(min(x0 + self.sheet_context, new_w - 1))
        x0 = int(x1 - self.sheet_context)
        r0 = new_h // 2 - self.staff_height // 2                                 # :160-164
        if self.data_augmentation['system_translation']:
            t = self.data_augmentation['system_translation']
            r0 += np.random.randint(low=-t, high=t + 1)
        if r0 < 0 or r0 + self.staff_height > new_h or x0 < 0:
            raise ValueError("sheet window (rows %d..%d, cols %d..%d) leaves the %dx%d strip - the reference's slice "
                             "would not fill the batch either" % (r0, r0 + self.staff_height, x0, x1, new_h, new_w))
        return [self._img_off[i_sheet], Ws, r0, sy, Hs - 1, x0, sx, Wc - 1, c0]

    def _audio_desc(self, i_sheet, i_spec, i_onset):
        spec = self.specs[i_sheet][i_spec]
        bins, T = spec.shape
        sel_onset = int(self.o2c_maps[i_sheet][i_spec][i_onset][0])
        if self.data_augmentation['onset_translation']:                         # :182-184
            t = self.data_augmentation['onset_translation']
            sel_onset += np.random.randint(low=-t, high=t + 1)
        start = max(sel_onset - self.spec_context // 2, 0)                       # :187-191
        stop = min(start + self.spec_context, T - 1)
        start = stop - self.spec_context
        if start < 0:
            raise ValueError("spectrogram shorter than one excerpt")
        y0 = 0
        if self.data_augmentation['spec_padding']:                               # :195-199: edge padding, then a shift
            pad = self.data_augmentation['spec_padding']
            y0 = np.random.randint(0, pad) - pad
        return [self._spec_off[i_sheet][i_spec], T, y0, 1.0, bins - 1, 0, 1.0, T - 1 - start, start]

    def _descriptors(self, key):
        if key.__class__ == int:
            key = slice(key, key + 1)
        ents = self.train_entities[key]
        d1, d2 = [], []
        for (i_sheet, i_spec, i_onset) in ents:          # same call order as the reference's loop (:217-222)
            d1.append(self._image_desc(i_sheet, i_spec, i_onset))
            d2.append(self._audio_desc(i_sheet, i_spec, i_onset))
        return np.asarray(d1, np.float64).reshape(-1, 9), np.asarray(d2, np.float64).reshape(-1, 9)

    def get_device(self, key):
        """-> (sheet DeviceBuffer (n,1,160,200) float32 un-normalised, spec DeviceBuffer (n,1,bins,42), n)"""
        d1, d2 = self._descriptors(key)
        n = d1.shape[0]
        eng = self.engine
        b1 = eng.alloc(max(4, n * self.sheet_dim[0] * self.sheet_dim[1] * 4))
        b2 = eng.alloc(max(4, n * self.spec_dim[0] * self.spec_dim[1] * 4))
        eng.gather_windows_dev(self._d_img.ptr, self._img_floats, d1, self.sheet_dim[0], self.sheet_dim[1], b1.ptr)
        eng.gather_windows_dev(self._d_spec.ptr, self._spec_floats, d2, self.spec_dim[0], self.spec_dim[1], b2.ptr)
        return b1, b2, n

    def __getitem__(self, key):
        b1, b2, n = self.get_device(key)
        out = [b1.download((n, 1) + tuple(self.sheet_dim), np.float32),
               b2.download((n, 1) + tuple(self.spec_dim), np.float32)]
        b1.free()
        b2.free()
        return out
