"""Multi-GPU partitioning of the retrieval path.  The reference is
single-device; these helpers are new.

One process per GPU.  The data plane is the library's own RCCL communicator
(asr_comm_init: all-gather / all-reduce enqueued on the context's stream, over
xGMI); the control plane that hands rank 0's communicator id to the other ranks
is either HubComm below (plain TCP on one node, no PyTorch anywhere) or an
existing torch.distributed process group (TorchComm: "gloo" in the CPU tests).
The pair list is sharded by contiguous index
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


class HubComm(object):
    """Control plane of a one-node job without PyTorch: rank 0 listens on an ephemeral port of MASTER_ADDR
    (127.0.0.1) and publishes it in a rendezvous file keyed by the launcher's pid and MASTER_PORT (all ranks of a
    job share both, under `torch.distributed.run` as well as under bench.py's / run_train's own spawner); every
    operation is "gather the ranks' byte strings at rank 0, send the list back".  Small host-side messages only
    (communicator id, barriers, timings, counters) - embeddings travel over RCCL inside the library.
    Same interface as TorchComm (rank, world, all_gather_rows, all_reduce_sum)."""

    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        import os
        import socket
        import tempfile
        import time
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self._peers, self._sock, self._path = [], None, None
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        key = key or os.environ.get("ASR_HUB_KEY") or "%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
        path = os.path.join(tempfile.gettempdir(), "asr_hub_%s" % key)
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, 0))
            srv.listen(self.world)
            srv.settimeout(timeout)
            with open(path + ".tmp", "w") as fp:
                fp.write("%s %d\n" % (addr, srv.getsockname()[1]))
            os.replace(path + ".tmp", path)               # appears atomically
            self._path = path
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                peers[int(self._recv(conn))] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    with open(path) as fp:
                        host, port = fp.read().split()
                    s = socket.create_connection((host, int(port)), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise RuntimeError("HubComm: rank 0 never published %s" % path)
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            self._send(s, str(self.rank).encode())
            self._sock = s

    @staticmethod
    def _send(sock, payload):
        import struct
        sock.sendall(struct.pack("<Q", len(payload)) + payload)

    @staticmethod
    def _recv(sock):
        import struct

        def exactly(n):
            buf = bytearray()
            while len(buf) < n:
                chunk = sock.recv(n - len(buf))
                if not chunk:
                    raise RuntimeError("HubComm: peer closed the connection")
                buf += chunk
            return bytes(buf)
        return exactly(struct.unpack("<Q", exactly(8))[0])

    def exchange(self, payload):
        """all-gather of one byte string per rank -> list in rank order"""
        import pickle
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv(c) for c in self._peers]
            blob = pickle.dumps(parts, protocol=4)
            for c in self._peers:
                self._send(c, blob)
            return parts
        self._send(self._sock, payload)
        return pickle.loads(self._recv(self._sock))

    def barrier(self):
        self.exchange(b"")

    def bcast_bytes(self, payload, src=0):
        return self.exchange(payload if self.rank == src else b"")[src]

    def all_gather_object(self, obj):
        import pickle
        return [pickle.loads(b) for b in self.exchange(pickle.dumps(obj, protocol=4))]

    def all_gather_rows(self, local):
        return np.concatenate(self.all_gather_object(np.ascontiguousarray(local)), axis=0)

    def all_reduce_sum(self, arr):
        parts = self.all_gather_object(np.asarray(arr))
        out = parts[0].copy()
        for p in parts[1:]:
            out = out + p
        return out

    def all_reduce_max(self, value):
        return max(self.all_gather_object(float(value)))

    def close(self):
        import os
        for c in self._peers + ([self._sock] if self._sock else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._sock = [], None
        if self._path:
            try:
                os.remove(self._path)
            except OSError:
                pass
            self._path = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
    """(allreduce, allgather) host callbacks for Engine.comm_init_custom on top of a TorchComm (gloo or nccl) or a
    HubComm: the device buffer is staged through host memory.  The native path is Engine.comm_init (RCCL inside the
    library); this one serves clusters where the process group is the only transport, several ranks sharing ONE GPU
    (RCCL refuses that), and the CPU tests."""
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
    """Give `engine` a communicator (before train_begin).
    transport "rccl": rank 0 draws the RCCL unique id and publishes it through `comm` (a HubComm - no PyTorch
    involved), through `store` (a torch.distributed Store) or, with neither, through the default torch process
    group's object broadcast; transport "host" (alias "torch"): host callbacks over `comm` (HubComm / TorchComm)."""
    if comm is not None and rank is None:
        rank, world = comm.rank, comm.world
    if transport in ("torch", "host"):
        if comm is None:
            comm = TorchComm()
            rank, world = (comm.rank if rank is None else rank), (comm.world if world is None else world)
        engine.comm_init_custom(rank, world, *make_torch_transport(engine, comm))
        return
    if transport != "rccl":
        raise ValueError("transport must be 'rccl' or 'host'")
    if isinstance(comm, HubComm):
        uid = comm.bcast_bytes(engine.comm_unique_id() if rank == 0 else b"", src=0)
    elif store is not None:
        if rank == 0:
            store.set("asr_comm_id", engine.comm_unique_id())
        uid = bytes(store.get("asr_comm_id"))
    else:
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank()
        if world is None:
            world = dist.get_world_size()
        box = [engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    engine.comm_init(rank, world, uid)


def broadcast_epoch(engine, epoch, root=0):
    """Data-parallel fit(): every rank evaluates the epoch's metrics on its own device, and float32 summation order
    may differ between ranks by one integer rank - enough to flip `map_va >= best` (utils/train_dcca_pool.py:391) on
    one rank only.  All ranks therefore continue with rank `root`'s numbers: the numeric entries of the epoch dict
    travel as one float64 vector through the communicator (zeros elsewhere + all-reduce sum)."""
    rank, world = engine.comm_info()
    if world <= 1:
        return epoch
    keys = sorted(k for k, v in epoch.items() if v is not None and k != "number")
    flat, shapes = [], []
    for k in keys:
        a = np.asarray(epoch[k], dtype=np.float64)
        shapes.append(a.shape)
        flat.append(a.ravel())
    vec = np.concatenate(flat) if flat else np.zeros(0)
    if rank != root:
        vec = np.zeros_like(vec)
    vec = engine.allreduce_host(vec)
    out, off = dict(epoch), 0
    for k, shp in zip(keys, shapes):
        n = int(np.prod(shp)) if shp else 1
        val = vec[off:off + n].reshape(shp)
        out[k] = val if shp else val[()]
        off += n
    return out
