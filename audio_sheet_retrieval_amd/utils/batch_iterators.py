"""Host-side batching helpers with the names and call signatures of the reference's utils/batch_iterators.py:
`batch_compute1/2` (chunked evaluation of a compiled callable, :17-111), the threaded prefetch generator (:114-157)
and `MultiviewPoolIteratorUnsupervised` (the training batch iterator over a data pool, :163-221).
"""
import queue
import threading

import numpy as np


def _padded(block, rows):
    """block zero-padded along axis 0 to `rows` rows (the reference pads the last chunk, :44-47)"""
    short = rows - block.shape[0]
    if short <= 0:
        return block
    return np.concatenate([block, np.zeros((short,) + block.shape[1:], dtype=block.dtype)], axis=0)


def _chunked(n_rows, batch_size, evaluate, verbose=False):
    """Evaluate `evaluate(lo, hi)` (returning at least hi-lo rows) over [0, n_rows) in steps of batch_size and
    stack the valid rows.  None for an empty input, like the reference (its result buffer is never created)."""
    out = None
    total = -(-n_rows // batch_size)
    for number, lo in enumerate(range(0, n_rows, batch_size), 1):
        hi = min(lo + batch_size, n_rows)
        if verbose:
            print("batch %d of %d" % (number, total), end="\r", flush=True)
        rows = evaluate(lo, hi)
        if out is None:
            out = np.zeros((n_rows,) + tuple(rows.shape[1:]), dtype=rows.dtype)
        out[lo:hi] = rows[:hi - lo]
    return out


def _fast_path(compute, prepare):
    """A function compiled by network.function() with no preparation or the model's own one: the library takes the
    whole unprepared array in one call and evaluates `prepare` in its first kernel.  Rows are independent in
    deterministic mode, so the reference's chunks and zero padding are not observable in the result."""
    from .. import network
    return isinstance(compute, network.CompiledFunction) and \
        (prepare is None or network.is_fused_prepare(compute.net, prepare))


def batch_compute1(X, compute, batch_size, verbose=False, prepare=None):
    """compute(prepare(chunk)) over X in chunks of batch_size; a short last chunk is zero-padded before the call
    and its padding rows are dropped afterwards."""
    if X.shape[0] and _fast_path(compute, prepare):
        if compute.view == 1 and prepare is None:
            return compute(X)
        return compute.embed_raw(X)

    def evaluate(lo, hi):
        block = _padded(X[lo:hi], batch_size)
        return compute(block if prepare is None else prepare(block))
    return _chunked(X.shape[0], batch_size, evaluate, verbose)


def batch_compute2(X1, X2, compute, batch_size, prepare1=None, prepare2=None):
    """Two-input form of batch_compute1.  `prepare2` is applied to the second input (the reference calls prepare1 on
    it, :98-99 - a slip none of its callers reach, they all leave prepare2 unset)."""
    if X1.shape[0] and prepare2 is None and _fast_path(compute, prepare1) and len(compute.views) == 2:
        if compute.view == 2:
            return compute.embed_raw(X2)
        return compute(X1, X2) if prepare1 is None else compute.embed_raw(X1)

    def evaluate(lo, hi):
        a, b = _padded(X1[lo:hi], batch_size), _padded(X2[lo:hi], batch_size)
        if prepare1 is not None:
            a = prepare1(a)
        if prepare2 is not None:
            b = prepare2(b)
        return compute(a, b)
    return _chunked(X1.shape[0], batch_size, evaluate)


_DONE = object()


def threaded_generator(generator, num_cached=10):
    """Run `generator` in a daemon thread, at most num_cached items ahead of the consumer."""
    buffer = queue.Queue(maxsize=num_cached)

    def fill():
        for item in generator:
            buffer.put(item)
        buffer.put(_DONE)

    threading.Thread(target=fill, daemon=True).start()
    while True:
        item = buffer.get()
        if item is _DONE:
            return
        yield item
        buffer.task_done()


def generator_from_iterator(iterator):
    yield from iterator


def threaded_generator_from_iterator(iterator, num_cached=10):
    return threaded_generator(generator_from_iterator(iterator), num_cached)


class MultiviewPoolIteratorUnsupervised(object):
    """Training batches from a two-view pool.

    One pass over the iterator is a "sub-epoch" of k_samples pairs (all of the pool when k_samples is None);
    successive passes walk through the pool window by window (:195-201).  A short last batch is filled up with
    pairs from the start of the pool (:204-210) so that every batch has batch_size rows, and after the last window
    the pool is reshuffled (:217-218).  `prepare(x, z)` is applied to every batch."""

    def __init__(self, batch_size, prepare=None, k_samples=None, shuffle=True):
        self.batch_size = batch_size
        self.prepare = prepare if prepare is not None else (lambda x, z: (x, z))
        self.shuffle = shuffle
        self.k_samples = k_samples
        self.epoch_counter = 0
        self.n_epochs = None

    def __call__(self, pool):
        self.pool = pool
        if self.k_samples is None:
            self.k_samples = pool.shape[0]
        self.n_batches = self.k_samples // self.batch_size
        self.n_epochs = max(1, pool.shape[0] // self.k_samples)
        return self

    def transform(self, xb, zb):
        return self.prepare(xb, zb)

    def __iter__(self):
        window = int(self.epoch_counter % self.n_epochs)
        first = window * self.k_samples
        for lo in range(0, self.k_samples, self.batch_size):
            xb, zb = self.pool[slice(first + lo, first + lo + self.batch_size)]
            missing = self.batch_size - xb.shape[0]
            if missing > 0:
                x_fill, z_fill = self.pool[0:missing]
                xb, zb = np.concatenate((xb, x_fill)), np.concatenate((zb, z_fill))
            yield self.transform(xb, zb)
        self.epoch_counter += 1
        if self.shuffle and window + 1 == self.n_epochs:
            self.pool.reset_batch_generator()
