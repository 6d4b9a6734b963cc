"""Fake `compute` callables and a fake data pool for pinning utils/batch_iterators.py (reference :17-111, :163-221)
against the mirror in audio_sheet_retrieval_amd/utils/batch_iterators.py.  Used by make_reference_golden.py (which runs
the REFERENCE on them, build container only) and by tests/test_reference_golden.py (which runs the mirror on them and
compares with the stored outputs).  Nothing here comes from the reference."""
import numpy as np

#: (tag, inputs) for batch_compute1 / batch_compute2: ragged last chunk, exact multiple, fewer rows than one chunk
COMPUTE_CASES = [
    ("ragged_u8", dict(n=23, batch_size=10, dtype="uint8", prepare=True)),
    ("exact_f32", dict(n=20, batch_size=10, dtype="float32", prepare=False)),
    ("short_f32", dict(n=3, batch_size=8, dtype="float32", prepare=True)),
    ("single_rows", dict(n=5, batch_size=1, dtype="uint8", prepare=False)),
]

#: (tag, iterator set-up): sub-epoch windows, a batch that crosses the window end, wrap-around fill, reshuffle point
ITERATOR_CASES = [
    ("windows", dict(n_pool=53, batch_size=10, k_samples=20, shuffle=True, passes=5)),
    ("ragged_windows", dict(n_pool=53, batch_size=10, k_samples=25, shuffle=True, passes=5)),
    ("whole_pool", dict(n_pool=53, batch_size=10, k_samples=None, shuffle=True, passes=3)),
    ("no_shuffle", dict(n_pool=40, batch_size=16, k_samples=None, shuffle=False, passes=2)),
    ("k_larger_than_pool", dict(n_pool=12, batch_size=5, k_samples=30, shuffle=True, passes=2)),
]


def compute_inputs(n, batch_size, dtype, prepare):
    rng = np.random.RandomState(n * 31 + batch_size)
    X1 = rng.randint(0, 256, (n, 1, 4, 6)).astype(dtype)
    X2 = rng.randint(0, 9, (n, 1, 3, 2)).astype("float32")
    return X1, X2


def prepare_one(E):
    return E.astype(np.float32) / np.float32(255)


def prepare_two(x, z):
    return x * 2, z + 1


class RecordingCompute(object):
    """a 'compiled function': returns batch-shaped rows and remembers what it was called with"""

    def __init__(self):
        self.calls = []

    def one(self, E):
        self.calls.append((E.shape[0], float(E.sum(dtype=np.float64)), str(E.dtype)))
        return (E.reshape(E.shape[0], -1)[:, :5].astype(np.float32) * 2 + 1)

    def two(self, E1, E2):
        self.calls.append((E1.shape[0], float(E1.sum(dtype=np.float64)) + 1000.0 * float(E2.sum(dtype=np.float64)),
                           str(E1.dtype)))
        return np.concatenate([E1.reshape(E1.shape[0], -1)[:, :3].astype(np.float32),
                               E2.reshape(E2.shape[0], -1)[:, :2].astype(np.float32)], axis=1)

    def log(self):
        """(n_calls, 3): rows passed, checksum of the chunk, 1.0 if the chunk arrived as float32"""
        return np.array([[c[0], c[1], 1.0 if c[2] == "float32" else 0.0] for c in self.calls], np.float64).reshape(-1, 3)


class FakePool(object):
    """what the iterator needs of utils/data_pools.py:AudioScoreRetrievalPool: shape, __getitem__(slice),
    reset_batch_generator() (a reshuffle drawing from NumPy's global RNG, like the reference pool's)"""

    def __init__(self, n):
        self.shape = [n]
        self.order = np.arange(n)
        self.resets = 0

    def __getitem__(self, key):
        ids = self.order[key]
        x = ids.astype(np.float32).reshape(-1, 1, 1, 1) * np.ones((1, 1, 2, 3), np.float32)
        z = (1000 + ids).astype(np.float32).reshape(-1, 1, 1, 1) * np.ones((1, 1, 2, 2), np.float32)
        return x, z

    def reset_batch_generator(self):
        self.resets += 1
        self.order = np.random.permutation(self.shape[0])


def run_passes(iterator, pool, passes):
    """iterate `passes` sub-epochs; per batch the sample ids it held (recovered from the prepared arrays), per pass the
    epoch counter, n_batches, n_epochs and the number of reshuffles so far"""
    it = iterator(pool)
    ids1, ids2, sizes, per_pass = [], [], [], []
    for _ in range(passes):
        n = 0
        for xb, zb in it:
            ids1.append((xb[:, 0, 0, 0] / 2).astype(np.int64))
            ids2.append((zb[:, 0, 0, 0] - 1 - 1000).astype(np.int64))
            sizes.append(xb.shape[0])
            n += 1
        per_pass.append([n, it.epoch_counter, it.n_batches, it.n_epochs, pool.resets])
    return dict(ids1=np.concatenate(ids1), ids2=np.concatenate(ids2), sizes=np.array(sizes, np.int64),
                per_pass=np.array(per_pass, np.int64))
