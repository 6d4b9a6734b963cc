"""Batched inference helpers and pool iterators - Python-3 restatement of the
reference's audio_sheet_retrieval/utils/batch_iterators.py with the same names,
arguments and behaviour (host-side plumbing around the compiled callables)."""
from __future__ import annotations

import queue
import sys
import threading

import numpy as np


def batch_compute1(X, compute, batch_size, verbose=False, prepare=None):
    """Batch compute data (utils/batch_iterators.py:17-62): chunk X, zero-pad the
    last chunk to batch_size (:44-47), optional prepare (:49-50), keep the valid
    rows (:60)."""
    R = None
    n_samples = X.shape[0]
    in_shape = list(X.shape)[1:]
    n_batches = int(np.ceil(float(n_samples) / batch_size))
    for i_batch in range(n_batches):
        if verbose:
            print("Processing batch %d / %d" % (i_batch + 1, n_batches), end="\r")
            sys.stdout.flush()
        start_idx = i_batch * batch_size
        E = X[start_idx:start_idx + batch_size]
        n_missing = batch_size - E.shape[0]
        if n_missing > 0:
            E = np.vstack((E, np.zeros([n_missing] + in_shape, dtype=X.dtype)))
        if prepare is not None:
            E = prepare(E)
        r = compute(E)
        if R is None:
            R = np.zeros([n_samples] + list(r.shape[1:]), dtype=r.dtype)
        R[start_idx:start_idx + r.shape[0]] = r[0:batch_size - n_missing]
    return R


def batch_compute2(X1, X2, compute, batch_size, prepare1=None, prepare2=None):
    """Batch compute data for two-input callables (utils/batch_iterators.py:65-111).
    Note the reference applies `prepare1` to E2 when `prepare2` is given (:98-99,
    a latent typo no caller reaches - every caller passes prepare2=None or relies
    on it being a no-op); here prepare2 is applied to E2."""
    R = None
    n_samples = X1.shape[0]
    in_shape1 = list(X1.shape)[1:]
    in_shape2 = list(X2.shape)[1:]
    n_batches = int(np.ceil(float(n_samples) / batch_size))
    for i_batch in range(n_batches):
        start_idx = i_batch * batch_size
        E1, E2 = X1[start_idx:start_idx + batch_size], X2[start_idx:start_idx + batch_size]
        n_missing = batch_size - E1.shape[0]
        if n_missing > 0:
            E1 = np.vstack((E1, np.zeros([n_missing] + in_shape1, dtype=X1.dtype)))
            E2 = np.vstack((E2, np.zeros([n_missing] + in_shape2, dtype=X2.dtype)))
        if prepare1 is not None:
            E1 = prepare1(E1)
        if prepare2 is not None:
            E2 = prepare2(E2)
        r = compute(E1, E2)
        if R is None:
            R = np.zeros([n_samples] + list(r.shape[1:]), dtype=r.dtype)
        R[start_idx:start_idx + r.shape[0]] = r[0:batch_size - n_missing]
    return R


def threaded_generator(generator, num_cached=10):
    """Producer thread + bounded queue (utils/batch_iterators.py:114-141)."""
    q = queue.Queue(maxsize=num_cached)
    end_marker = object()

    def producer():
        for item in generator:
            q.put(item)
        q.put(end_marker)

    thread = threading.Thread(target=producer)
    thread.daemon = True
    thread.start()
    item = q.get()
    while item is not end_marker:
        yield item
        q.task_done()
        item = q.get()


def generator_from_iterator(iterator):
    """(:144-149)"""
    for x in iterator:
        yield x


def threaded_generator_from_iterator(iterator, num_cached=10):
    """(:152-157)"""
    return threaded_generator(generator_from_iterator(iterator), num_cached)


class MultiviewPoolIteratorUnsupervised(object):
    """Batch iterator for multiview data (utils/batch_iterators.py:163-221):
    sub-epochs of k_samples cycling through the pool (:195-201), wrap-around fill
    of a short last batch (:204-210), reshuffle after a full pass (:217-218)."""

    def __init__(self, batch_size, prepare=None, k_samples=None, shuffle=True):
        self.batch_size = batch_size
        if prepare is None:
            def prepare(x, y):
                return x, y
        self.prepare = prepare
        self.shuffle = shuffle
        self.k_samples = k_samples
        self.epoch_counter = 0
        self.n_epochs = None

    def __call__(self, pool):
        self.pool = pool
        if self.k_samples is None:
            self.k_samples = self.pool.shape[0]
        self.n_batches = self.k_samples // self.batch_size
        self.n_epochs = max(1, self.pool.shape[0] // self.k_samples)
        return self

    def __iter__(self):
        n_samples = self.k_samples
        bs = self.batch_size
        idx_epoch = np.mod(self.epoch_counter, self.n_epochs)
        for i in range((n_samples + bs - 1) // bs):
            i_start = i * bs + idx_epoch * self.k_samples
            i_stop = (i + 1) * bs + idx_epoch * self.k_samples
            xb, zb = self.pool[slice(i_start, i_stop)]
            if xb.shape[0] < self.batch_size:
                n_missing = self.batch_size - xb.shape[0]
                x_con, z_con = self.pool[0:n_missing]
                xb = np.concatenate((xb, x_con))
                zb = np.concatenate((zb, z_con))
            yield self.transform(xb, zb)
        self.epoch_counter += 1
        if self.shuffle and (idx_epoch + 1) == self.n_epochs:
            self.pool.reset_batch_generator()

    def transform(self, xb, zb):
        return self.prepare(xb, zb)
