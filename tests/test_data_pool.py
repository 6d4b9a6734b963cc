"""Training-pool batch assembly: oracle self-checks on CPU, device == oracle (bit-exact) on the GPU."""
import numpy as np
import pytest

FULL_AUG = dict(system_translation=5, sheet_scaling=[0.95, 1.05], onset_translation=1, spec_padding=0, interpolate=-1)
PAD_AUG = dict(system_translation=3, sheet_scaling=[0.9, 1.1], onset_translation=2, spec_padding=3, interpolate=-1)
NO_AUG = dict(system_translation=0, sheet_scaling=[1.0, 1.0], onset_translation=0, spec_padding=0, interpolate=-1)


def _fake_pieces(rng, n_pieces=3):
    images, specs, maps = [], [], []
    for p in range(n_pieces):
        W = int(rng.integers(1500, 2600))
        img = (rng.random((200, W)) * 255).astype(np.float32)
        T = int(rng.integers(500, 900))
        sp = [(3 * rng.random((92, T)) ** 2).astype(np.float32) for _ in range(1 + p % 2)]
        onsets = np.sort(rng.choice(np.arange(30, T - 30), size=40, replace=False))
        coords = np.linspace(450, W - 450, 40).astype(np.int64)
        images.append(img)
        specs.append(sp)
        maps.append([np.stack([onsets, coords], axis=1).astype(np.int64) for _ in sp])
    return images, specs, maps


def test_oracle_pool_no_augmentation_is_plain_slicing():
    from oracle import data_pool as op
    rng = np.random.default_rng(0)
    images, specs, maps = _fake_pieces(rng)
    ents = op.prepare_train_entities(images, specs, maps, 42, 200)
    assert len(ents) > 20
    sheet, spec = op.get_batch(images, specs, maps, NO_AUG, ents[:5])
    i_sheet, i_spec, i_onset = ents[0]
    onset, coord = maps[i_sheet][i_spec][i_onset]
    assert np.array_equal(spec[0, 0], specs[i_sheet][i_spec][:, onset - 21:onset + 21])
    # the 800-px crop is centred on the note, the 200-px window on the crop: columns coord-100 .. coord+100
    assert np.array_equal(sheet[0, 0], images[i_sheet][20:180, coord - 100:coord + 100])
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(op.resize_nearest(img, 8, 6), np.repeat(np.repeat(img, 2, axis=0), 2, axis=1))


@pytest.mark.gpu
@pytest.mark.parametrize("aug", [NO_AUG, FULL_AUG, PAD_AUG])
def test_device_pool_matches_oracle_bit_exact(aug):
    from audio_sheet_retrieval_amd import _lib
    from audio_sheet_retrieval_amd.utils.data_pools import AudioScoreRetrievalPool
    from oracle import data_pool as op
    rng = np.random.default_rng(1)
    images, specs, maps = _fake_pieces(rng)
    eng = _lib.Engine("mutopia_ccal_cont")
    np.random.seed(7)
    pool = AudioScoreRetrievalPool(eng, images, specs, maps, data_augmentation=dict(aug), shuffle=True)
    ents = pool.train_entities.copy()
    np.random.seed(11)
    got = pool[0:37]
    np.random.seed(11)
    ref = op.get_batch(images, specs, maps, aug, ents[0:37])
    assert got[0].shape == (37, 1, 160, 200) and got[1].shape == (37, 1, 92, 42)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # the device buffers feed the towers directly (un-normalised sheets -> ASR_IN_F32_RAW)
    np.random.seed(11)
    b1, b2, n = pool.get_device(slice(0, 8))
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    eng.set_params(synth_data.synth_params(param_shapes("mutopia_ccal_cont"), seed=1, trained_like=True))
    out = eng.alloc(8 * 32 * 4)
    eng.embed_view1_dev(b1.ptr, _lib.IN_F32_RAW, n, out.ptr)
    eng.sync()
    lv = out.download((8, 32), np.float32)
    assert np.abs(lv - eng.embed_view1(ref[0][:8], prepared=False)).max() <= 1e-6
    eng.close()
