"""Multi-GPU partitioning of the retrieval path.  The reference is
single-device; these helpers are new.

One process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  The pair list is sharded by contiguous index
ranges; embedding needs no communication (deterministic mode is row-independent,
utils/batch_iterators.py:90-93 relies on the same fact).  Ranking has ONE exchange
step: an all-gather of the 32-d candidate embeddings; every rank then ranks its
own queries against all candidates with `query_offset` = its first global index,
and the integer hit counters are all-reduced.  Integer results do not depend on
the number of ranks.

The rank / top-k arithmetic itself is passed in as a callable (Engine.rank /
Engine.topk on the GPU) so that the exchange logic can be exercised on CPU.
"""
from __future__ import annotations

import numpy as np


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of `n` items owned by `rank`: the first n % world ranks
    get one extra item."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class TorchComm(object):
    """Thin wrapper over an initialised torch.distributed process group working on
    host NumPy arrays (device-resident exchange lives in bench.py)."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device            # None: CPU tensors (gloo); "cuda": staged through the GPU (nccl)

    def _t(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device) if self.device else t

    def all_gather_rows(self, local):
        """Concatenate every rank's (n_r, ...) array along axis 0 (ragged n_r)."""
        local = np.ascontiguousarray(local)
        counts = [None] * self.world
        self.dist.all_gather_object(counts, int(local.shape[0]), group=self.group)
        nmax = max(counts)
        pad = np.zeros((nmax,) + local.shape[1:], dtype=local.dtype)
        pad[:local.shape[0]] = local
        outs = [self.torch.empty_like(self._t(pad)) for _ in range(self.world)]
        self.dist.all_gather(outs, self._t(pad), group=self.group)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)

    def all_reduce_sum(self, arr):
        t = self._t(np.asarray(arr))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


def sharded_eval_retrieval(rank_fn, lv1_local, lv2_local, comm):
    """eval_retrieval (utils/train_dcca_pool.py:28-82) over a pair list sharded
    across ranks (equal list sizes n1 == n2, pairs co-located).
    rank_fn(lv1, lv2_all, query_offset, n1_global) -> (ranks, dstar, ties).
    Returns (mean_rank, median_rank, mean_dist, hit_rates, map) - identical on
    every rank - plus this rank's integer ranks."""
    counts = comm.all_gather_rows(np.array([[lv1_local.shape[0]]], dtype=np.int64)).ravel()
    offset = int(counts[:comm.rank].sum())
    n_global = int(counts.sum())
    lv2_all = comm.all_gather_rows(lv2_local)                 # the one exchange step
    ranks, dstar, _ties = rank_fn(lv1_local, lv2_all, offset, n_global)
    hits = np.array([np.count_nonzero(ranks <= k) for k in (1, 5, 10, 25)], dtype=np.int64)
    hits = comm.all_reduce_sum(hits)
    # order statistics (median) need the ranks themselves: n int32 - tiny next to the embeddings
    all_ranks = comm.all_gather_rows(ranks.astype(np.int32))
    all_dstar = comm.all_gather_rows(dstar.astype(np.float64))
    hit_rates = dict(zip((1, 5, 10, 25), (int(h) for h in hits)))
    stats = (np.mean(all_ranks), np.median(all_ranks), float(np.mean(all_dstar)), hit_rates,
             float(np.mean(1.0 / all_ranks.astype(np.float64))))
    return stats, ranks


def merge_topk(idx_lists, dist_lists, k):
    """k-way merge of per-shard top-k lists: (distance, index) lexicographic."""
    cat_idx = np.concatenate(idx_lists, axis=1)
    cat_dist = np.concatenate(dist_lists, axis=1)
    cat_idx_key = np.where(cat_idx < 0, np.iinfo(np.int32).max, cat_idx)
    order = np.lexsort((cat_idx_key, cat_dist), axis=1)[:, :k]
    return np.take_along_axis(cat_idx, order, axis=1), np.take_along_axis(cat_dist, order, axis=1)


def sharded_topk(topk_fn, db_local, queries_local, k, comm):
    """Global top-k over a candidate pool sharded across ranks (BASELINE config 5;
    audio_sheet_server.py:530-563 on one device): all-gather the queries, each
    rank searches its shard with global indices, the per-shard lists are
    all-gathered and merged.  topk_fn(db, q, k, idx_offset) -> (idx, dist).
    Returns (idx, dist) for THIS rank's queries."""
    qcounts = comm.all_gather_rows(np.array([[queries_local.shape[0]]], dtype=np.int64)).ravel()
    dcounts = comm.all_gather_rows(np.array([[db_local.shape[0]]], dtype=np.int64)).ravel()
    q_all = comm.all_gather_rows(queries_local)
    db_offset = int(dcounts[:comm.rank].sum())
    idx, dist = topk_fn(db_local, q_all, k, db_offset)         # (Q_all, k) against my shard
    # exchange the short lists: Q_all*k*(4+8) bytes per rank
    idx_all = comm.all_gather_rows(idx[None])                  # (world, Q_all, k)
    dist_all = comm.all_gather_rows(dist[None])
    midx, mdist = merge_topk(list(idx_all), list(dist_all), k)
    q_lo = int(qcounts[:comm.rank].sum())
    return midx[q_lo:q_lo + queries_local.shape[0]], mdist[q_lo:q_lo + queries_local.shape[0]]


# --------------------------------------------------------------------------
# data-parallel training (SURVEY.md 8e "Training partitioning")
# --------------------------------------------------------------------------
def shard_batch(arrays, rank, world):
    """This rank's rows of a batch every rank drew identically (same iterator seed).  The library needs equal shard
    sizes, so a batch whose size is not a multiple of `world` loses its last len % world rows on every rank."""
    n = (int(arrays[0].shape[0]) // int(world)) * int(world)
    per = n // int(world)
    return [np.ascontiguousarray(a[rank * per:(rank + 1) * per]) for a in arrays]


def make_torch_transport(engine, comm):
    """(allreduce, allgather) host callbacks for Engine.comm_init_custom on top of a TorchComm (gloo or nccl): the
    device buffer is staged through host memory.  The native path is Engine.comm_init (RCCL inside the library);
    this one serves clusters where the process group is the only transport, and the CPU tests."""
    from . import _lib

    def allreduce(buf, count, dtype):
        dt = np.float64 if dtype == _lib.DTYPE_F64 else np.float32
        engine.raw_upload(buf, comm.all_reduce_sum(engine.raw_download(buf, (count,), dt)).astype(dt, copy=False))
        return 0

    def allgather(send, recv, nbytes):
        rows = comm.all_gather_rows(engine.raw_download(send, (1, nbytes), np.uint8))
        engine.raw_upload(recv, rows.reshape(-1))
        return 0
    return allreduce, allgather


def init_data_parallel(engine, rank=None, world=None, transport="rccl", comm=None, store=None):
    """Give `engine` a communicator (before train_begin).  transport "rccl": rank 0 draws the RCCL unique id and
    publishes it through `store` (a torch.distributed Store; default: the default process group's object
    broadcast); transport "torch": host callbacks over `comm` (TorchComm)."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank()
    if world is None:
        world = dist.get_world_size()
    if transport == "torch":
        comm = comm or TorchComm()
        engine.comm_init_custom(rank, world, *make_torch_transport(engine, comm))
        return
    if transport != "rccl":
        raise ValueError("transport must be 'rccl' or 'torch'")
    if store is not None:
        if rank == 0:
            store.set("asr_comm_id", engine.comm_unique_id())
        uid = bytes(store.get("asr_comm_id"))
    else:
        box = [engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    engine.comm_init(rank, world, uid)
