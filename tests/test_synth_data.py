"""CPU: synthetic MSMD-shaped pools are deterministic, shard-consistent and
honour the AudioScoreRetrievalPool contract (utils/data_pools.py:203-228)."""
import numpy as np

from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes


def test_shapes_dtypes_ranges():
    s, z = synth_data.synth_pairs(np.arange(5))
    assert s.shape == (5, 1, 160, 200) and s.dtype == np.uint8
    assert z.shape == (5, 1, 92, 42) and z.dtype == np.float32 and z.min() >= 0
    assert (s == 255).mean() > 0.7          # mostly white sheet
    assert not np.array_equal(s[0], s[1])


def test_counter_based_sharding():
    a = synth_data.synth_pairs(np.arange(0, 12))
    b0 = synth_data.synth_pairs(np.arange(0, 6))
    b1 = synth_data.synth_pairs(np.arange(6, 12))
    assert np.array_equal(a[0], np.concatenate([b0[0], b1[0]]))
    assert np.array_equal(a[1], np.concatenate([b0[1], b1[1]]))
    c = synth_data.synth_pairs(np.array([7, 3]))
    assert np.array_equal(c[0][0], a[0][7]) and np.array_equal(c[1][1], a[1][3])
    assert not np.array_equal(synth_data.synth_pairs([0], seed=1)[0], synth_data.synth_pairs([0], seed=2)[0])


def test_pool_protocol():
    pool = synth_data.SyntheticRetrievalPool(50, seed=23, shuffle=False)
    assert pool.shape[0] == 50
    x, z = pool[0:7]
    assert x.dtype == np.float32 and x.shape == (7, 1, 160, 200) and x.max() <= 255 and x.max() > 1
    x1, _ = pool[3]
    assert x1.shape == (1, 1, 160, 200) and np.array_equal(x1[0], x[3])
    xi, _ = pool[np.array([6, 2])]
    assert np.array_equal(xi[0], x[6]) and np.array_equal(xi[1], x[2])
    u8, _ = pool.get_u8(slice(0, 7))
    assert u8.dtype == np.uint8 and np.array_equal(u8.astype(np.float32), x)
    sh = synth_data.SyntheticRetrievalPool(50, seed=23, shuffle=True)
    first = sh.train_entities.copy()
    sh.reset_batch_generator()
    assert sorted(first) == list(range(50)) and not np.array_equal(first, sh.train_entities)
    d = synth_data.load_synthetic_retrieval(20, 10, 10)
    assert set(d) == {"train", "valid", "test", "train_tag"}
    assert not np.array_equal(d["valid"][0:1][1], d["test"][0:1][1])


def test_synth_params_layout():
    for name in ("mutopia_ccal_cont", "mutopia_ccal_cont_rsz"):
        shapes = param_shapes(name)
        p = synth_data.synth_params(shapes, seed=1)
        assert [a.shape for a in p] == shapes and all(a.dtype == np.float32 for a in p)
        assert p[2].min() == 1.0 and np.abs(p[90]).max() == 0.0          # Lasagne defaults
        q = synth_data.synth_params(shapes, seed=1, trained_like=True)
        assert np.array_equal(p[0], q[0]) and np.abs(np.linalg.det(q[90].astype(np.float64))) > 1e-6
        assert np.array_equal(synth_data.synth_params(shapes, seed=1)[5], p[5])
