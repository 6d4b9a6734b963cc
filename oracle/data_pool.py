"""TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned by a run of the reference's own pool class
(tests/golden/reference_golden.npz, keys pool/*) except for the cv2 rescaling, which stays unpinned.

CPU restatement of the training pool of the reference (audio_sheet_retrieval/utils/data_pools.py):
  interpolate            :61-82    densified onset -> coordinate maps
  prepare_train_entities :87-117   (including the reference's `c_stop = o_start + sheet_context`, :109)
  prepare_train_image    :126-170  crop around the target note, cv2.resize(INTER_NEAREST) scaling, vertical crop
  prepare_train_audio    :172-201  excerpt around the (translated) onset, edge padding
  __getitem__            :203-228
Third-party semantic, unverified offline (cv2 is absent here): cv2.resize(..., INTER_NEAREST) maps destination index
d to source index min(floor(d * (1 / (dst_size / src_size))), src_size - 1) per axis (OpenCV 3.1 resizeNN).
The NumPy global RNG is consumed in the reference's order, so seeding it reproduces a reference run.
"""
import numpy as np


def resize_nearest(img, new_w, new_h):
    h, w = img.shape
    ifx = 1.0 / (float(new_w) / w)
    ify = 1.0 / (float(new_h) / h)
    xs = np.minimum(np.floor(np.arange(new_w) * ifx).astype(np.int64), w - 1)
    ys = np.minimum(np.floor(np.arange(new_h) * ify).astype(np.int64), h - 1)
    return img[ys][:, xs]


def interpolate(o2c_maps, step):
    """new maps (lists of lists of (n, 2) int64 arrays); the input is left alone"""
    from scipy.interpolate import interp1d
    out = []
    for per_piece in o2c_maps:
        maps = []
        for o2c in per_piece:
            frames = np.arange(o2c[0, 0], o2c[-1, 0] + 1, step)
            xs = interp1d(o2c[:, 0], o2c[:, 1])(frames)
            maps.append(np.hstack((frames.reshape(-1, 1), xs.reshape(-1, 1))).astype(np.int64))
        out.append(maps)
    return out


def prepare_train_entities(images, specs, o2c_maps, spec_context, sheet_context):
    ents = []
    for i_sheet, sheet in enumerate(images):
        for i_spec, spec in enumerate(specs[i_sheet]):
            for i_onset in range(len(o2c_maps[i_sheet][i_spec])):
                onset = o2c_maps[i_sheet][i_spec][i_onset, 0]
                o_start = onset - spec_context // 2
                o_stop = o_start + spec_context
                coord = o2c_maps[i_sheet][i_spec][i_onset, 1]
                c_start = coord - sheet_context // 2
                c_stop = o_start + sheet_context                 # sic (:109)
                if o_start >= 0 and o_stop < spec.shape[1] and c_start >= 0 and c_stop < sheet.shape[1]:
                    ents.append((i_sheet, i_spec, i_onset))
    return np.asarray(ents, dtype=np.int64).reshape(-1, 3)


def prepare_train_image(images, o2c_maps, aug, i_sheet, i_spec, i_onset, sheet_context, staff_height):
    sheet = images[i_sheet]
    target_coord = int(o2c_maps[i_sheet][i_spec][i_onset][1])
    c0 = max(0, target_coord - 2 * sheet_context)
    c1 = min(c0 + 4 * sheet_context, sheet.shape[1])
    c0 = max(0, c1 - 4 * sheet_context)
    sheet = sheet[:, c0:c1]
    if aug['sheet_scaling']:
        sc = aug['sheet_scaling']
        scale = (sc[1] - sc[0]) * np.random.random_sample() + sc[0]
        new_size = (int(sheet.shape[1] * scale), int(sheet.shape[0] * scale))
        sheet = resize_nearest(sheet, new_size[0], new_size[1])
    x = sheet.shape[1] // 2
    x0 = np.max([x - sheet_context // 2, 0])
    x1 = x0 + sheet_context
    x1 = int(np.min([x1, sheet.shape[1] - 1]))
    x0 = int(x1 - sheet_context)
    r0 = sheet.shape[0] // 2 - staff_height // 2
    if aug['system_translation']:
        t = aug['system_translation']
        r0 += np.random.randint(low=-t, high=t + 1)
    r1 = r0 + staff_height
    return sheet[r0:r1, x0:x1]


def prepare_train_audio(specs, o2c_maps, aug, i_sheet, i_spec, i_onset, spec_context):
    spec = specs[i_sheet][i_spec]
    sel_onset = int(o2c_maps[i_sheet][i_spec][i_onset][0])
    if aug['onset_translation']:
        t = aug['onset_translation']
        sel_onset += np.random.randint(low=-t, high=t + 1)
    start = np.max([sel_onset - spec_context // 2, 0])
    stop = start + spec_context
    stop = np.min([stop, spec.shape[1] - 1])
    start = stop - spec_context
    excerpt = spec[:, start:stop]
    if aug['spec_padding']:
        pad = aug['spec_padding']
        excerpt = np.pad(excerpt, ((pad, pad), (0, 0)), mode='edge')
        s = np.random.randint(0, pad)
        excerpt = excerpt[s:s + spec.shape[0], :]
    return excerpt


def get_batch(images, specs, o2c_maps, aug, entities, spec_context=42, sheet_context=200, staff_height=160):
    sheet_batch = np.zeros((len(entities), 1, staff_height, sheet_context), dtype=np.float32)
    spec_batch = np.zeros((len(entities), 1, specs[0][0].shape[0], spec_context), dtype=np.float32)
    for i, (i_sheet, i_spec, i_onset) in enumerate(entities):
        sheet_batch[i, 0] = prepare_train_image(images, o2c_maps, aug, i_sheet, i_spec, i_onset, sheet_context, staff_height)
        spec_batch[i, 0] = prepare_train_audio(specs, o2c_maps, aug, i_sheet, i_spec, i_onset, spec_context)
    return [sheet_batch, spec_batch]
