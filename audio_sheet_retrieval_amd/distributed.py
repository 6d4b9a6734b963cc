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


class HubError(RuntimeError):
    """The control plane failed: a peer died, timed out or sent something that is not the protocol."""


def _pack(obj):
    """Wire form of the small host-side values the hub carries - no pickle: bytes, float, int, str, None, NumPy arrays
    and lists / tuples / dicts (str keys) of those, as a JSON header + raw array / byte blobs."""
    import json
    import struct
    blobs = []

    def enc(o):
        if o is None or isinstance(o, (bool, str)):
            return o
        if isinstance(o, (int, np.integer)):
            return {"$i": str(int(o))}
        if isinstance(o, (float, np.floating)):
            return {"$f": float(o).hex()}
        if isinstance(o, (bytes, bytearray, memoryview)):
            blobs.append(bytes(o))
            return {"$b": len(blobs) - 1}
        if isinstance(o, np.ndarray):
            if o.dtype.kind not in "biuf":
                raise TypeError("HubComm carries numeric arrays only, not dtype %s" % o.dtype)
            blobs.append(np.ascontiguousarray(o).tobytes())
            return {"$a": len(blobs) - 1, "dtype": o.dtype.str, "shape": list(o.shape)}
        if isinstance(o, (list, tuple)):
            return {"$l": [enc(x) for x in o]}
        if isinstance(o, dict):
            if not all(isinstance(k, str) for k in o):
                raise TypeError("HubComm carries dicts with str keys only")
            return {"$d": {k: enc(v) for k, v in o.items()}}
        raise TypeError("HubComm cannot carry %r" % type(o))
    head = json.dumps(enc(obj)).encode()
    out = [struct.pack("<II", len(head), len(blobs)), head]
    for b in blobs:
        out.append(struct.pack("<Q", len(b)))
        out.append(b)
    return b"".join(out)


def _unpack(buf):
    import json
    import struct
    if len(buf) < 8:
        raise HubError("HubComm: truncated message")
    n_head, n_blobs = struct.unpack_from("<II", buf, 0)
    off = 8
    if n_head > len(buf) - off or n_blobs > 1 << 16:
        raise HubError("HubComm: malformed message")
    head = json.loads(bytes(buf[off:off + n_head]).decode())
    off += n_head
    blobs = []
    for _ in range(n_blobs):
        if off + 8 > len(buf):
            raise HubError("HubComm: malformed message")
        (ln,) = struct.unpack_from("<Q", buf, off)
        off += 8
        if ln > len(buf) - off:
            raise HubError("HubComm: malformed message")
        blobs.append(bytes(buf[off:off + ln]))
        off += ln

    def dec(o):
        if o is None or isinstance(o, (bool, str)):
            return o
        if not isinstance(o, dict) or len(o) < 1:
            raise HubError("HubComm: malformed message")
        if "$i" in o:
            return int(o["$i"])
        if "$f" in o:
            return float.fromhex(o["$f"])
        if "$b" in o:
            return blobs[int(o["$b"])]
        if "$a" in o:
            dt = np.dtype(str(o["dtype"]))
            if dt.kind not in "biuf":
                raise HubError("HubComm: array dtype %s not accepted" % dt)
            shape = tuple(int(x) for x in o["shape"])
            raw = blobs[int(o["$a"])]
            if int(np.prod(shape, dtype=np.int64)) * dt.itemsize != len(raw):
                raise HubError("HubComm: array size does not match its shape")
            return np.frombuffer(raw, dtype=dt).reshape(shape).copy()
        if "$l" in o:
            return [dec(x) for x in o["$l"]]
        if "$d" in o:
            return {str(k): dec(v) for k, v in o["$d"].items()}
        raise HubError("HubComm: malformed message")
    return dec(head)


class HubComm(object):
    """Control plane of a one-node job without PyTorch: rank 0 listens on an ephemeral LOOPBACK port and publishes
    `port token` in a rendezvous file keyed by the launcher's pid and MASTER_PORT (all ranks of a job share both,
    under `torch.distributed.run` as well as under bench.py's / run_train's own spawner); every operation is "gather
    the ranks' byte strings at rank 0, send the list back".  Small host-side messages only (communicator id, barriers,
    timings, counters) - embeddings travel over RCCL inside the library.

    It is not a service: the file is created exclusively (O_EXCL | O_NOFOLLOW, mode 0600, must belong to this user),
    carries a random 128-bit token every peer has to present together with a rank in 1..world-1 that nobody else has
    claimed, the listener binds 127.0.0.1 unless ASR_HUB_BIND names another address, and nothing that arrives is ever
    unpickled (values travel as JSON + raw array bytes, `_pack` / `_unpack`).  A connection that does not present the
    token is dropped and does not count.  Any failure raises HubError within `timeout` seconds (default 60,
    ASR_HUB_TIMEOUT).  Same interface as TorchComm (rank, world, all_gather_rows, all_reduce_sum)."""

    MAX_MESSAGE = 1 << 30

    def __init__(self, rank=None, world=None, key=None, timeout=None):
        import os
        import socket
        import tempfile
        import time
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self._peers, self._sock, self._path = [], None, None
        if self.world == 1:
            return
        if not 0 <= self.rank < self.world:
            raise HubError("HubComm: rank %d outside 0..%d" % (self.rank, self.world - 1))
        if timeout is None:
            timeout = float(os.environ.get("ASR_HUB_TIMEOUT", "60"))
        self.timeout = timeout
        bind = os.environ.get("ASR_HUB_BIND", "127.0.0.1")
        key = key or os.environ.get("ASR_HUB_KEY") or "%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
        if not all(c.isalnum() or c in "_-." for c in key):
            raise HubError("HubComm: rendezvous key %r has characters outside [A-Za-z0-9_.-]" % key)
        path = os.path.join(tempfile.gettempdir(), "asr_hub_%s" % key)
        deadline = time.time() + timeout
        if self.rank == 0:
            import hmac
            import secrets
            token = secrets.token_hex(16)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.bind((bind, 0))
            srv.listen(self.world + 8)
            try:
                os.unlink(path)                             # a crashed job's file (a link is removed, never followed)
            except OSError:
                pass
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as fp:
                fp.write("%s %d %s\n" % (bind, srv.getsockname()[1], token))
            self._path = path
            peers = {}
            try:
                while len(peers) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise HubError("HubComm: only %d of %d ranks joined within %.0f s" % (len(peers) + 1, self.world,
                                                                                               timeout))
                    srv.settimeout(left)
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(min(5.0, timeout))
                    try:
                        hello = self._recv(conn, limit=256).split(b" ")
                        r = int(hello[1]) if len(hello) == 2 and hello[1].isdigit() else -1
                        good = len(hello) == 2 and hmac.compare_digest(hello[0], token.encode()) and \
                            1 <= r < self.world and r not in peers
                    except (HubError, OSError, ValueError):
                        good = False
                    if not good:                             # not one of ours: dropped, does not count
                        conn.close()
                        continue
                    conn.settimeout(timeout)
                    peers[r] = conn
            except BaseException:
                for c in peers.values():
                    c.close()
                srv.close()
                self.close()
                raise
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    host, port, token = self._read_rendezvous(path)
                    s = socket.create_connection((host, port), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise HubError("HubComm: rank 0 never published %s" % path)
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            self._send(s, token.encode() + b" " + str(self.rank).encode())
            self._sock = s

    @staticmethod
    def _read_rendezvous(path):
        """(host, port, token) from a file that is a regular file of this user nobody else can write"""
        import os
        import stat
        fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        try:
            st = os.fstat(fd)
            if not stat.S_ISREG(st.st_mode) or st.st_uid != os.geteuid() or st.st_mode & 0o077:
                raise HubError("HubComm: %s is not a private file of this user" % path)
            text = os.read(fd, 512).decode()
        finally:
            os.close(fd)
        host, port, token = text.split()
        return host, int(port), token

    @staticmethod
    def _send(sock, payload):
        import struct
        try:
            sock.sendall(struct.pack("<Q", len(payload)) + payload)
        except OSError as e:
            raise HubError("HubComm: send failed (%s) - a rank has died or stalled" % e)

    @classmethod
    def _recv(cls, sock, limit=None):
        import struct

        def exactly(n):
            buf = bytearray()
            while len(buf) < n:
                try:
                    chunk = sock.recv(min(n - len(buf), 1 << 20))
                except OSError as e:
                    raise HubError("HubComm: receive failed (%s) - a rank has died or stalled" % e)
                if not chunk:
                    raise HubError("HubComm: peer closed the connection")
                buf += chunk
            return bytes(buf)
        n = struct.unpack("<Q", exactly(8))[0]
        if n > (cls.MAX_MESSAGE if limit is None else limit):
            raise HubError("HubComm: message of %d bytes refused" % n)
        return exactly(n)

    def exchange(self, payload):
        """all-gather of one byte string per rank -> list in rank order"""
        if self.world == 1:
            return [bytes(payload)]
        if getattr(self, "_wd_thread", None) is not None or getattr(self, "_wd_used", False):
            raise HubError("HubComm: the hub is in watchdog mode (start_watchdog); no further exchanges")
        if self.rank == 0:
            parts = [bytes(payload)] + [self._recv(c) for c in self._peers]
            blob = _pack(parts)
            for c in self._peers:
                self._send(c, blob)
            return parts
        self._send(self._sock, bytes(payload))
        parts = _unpack(self._recv(self._sock))
        if not isinstance(parts, list) or len(parts) != self.world or not all(isinstance(b, bytes) for b in parts):
            raise HubError("HubComm: malformed reply from rank 0")
        return parts

    def barrier(self):
        self.exchange(b"")

    def bcast_bytes(self, payload, src=0):
        return self.exchange(payload if self.rank == src else b"")[src]

    def all_gather_object(self, obj):
        """one value per rank (numbers, strings, bytes, numeric arrays, lists / dicts of those - see _pack)"""
        return [_unpack(b) for b in self.exchange(_pack(obj))]

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

    # -- dead-peer watchdog ------------------------------------------------------------------------------------------
    # After the communicator id has travelled, a job on the RCCL transport has no use for the hub - except this one:
    # a process that dies closes its sockets, which is the only timely sign its peers get (an RCCL collective waits
    # for ever).  In watchdog mode a daemon thread watches the connections (rank 0: every peer's; the others: rank
    # 0's): end-of-file or a reset without the goodbye byte means the peer is gone and `on_dead(rank)` runs - by
    # default a message on stderr and os._exit(75), which in turn closes this rank's sockets, so the loss of ANY rank
    # takes the whole job down within a second or two (through rank 0).  The only byte ever sent in this mode is the
    # goodbye (stop_watchdog).  No exchange is possible afterwards: close the hub.
    def start_watchdog(self, on_dead=None):
        import os
        import sys
        import threading
        if self.world == 1 or getattr(self, "_wd_thread", None) is not None:
            return
        socks = dict((c, r + 1) for r, c in enumerate(self._peers)) if self.rank == 0 else {self._sock: 0}
        stop = threading.Event()

        def default_on_dead(peer):
            sys.stderr.write("[asr] rank %d: rank %d is gone - leaving (a collective would wait for it for ever)\n"
                             % (self.rank, peer))
            sys.stderr.flush()
            # os._exit skips atexit handlers and finally blocks: remove what this rank owns first (rank 0's rendezvous
            # file, the job's tune-cache directory); parameter files are written through a rename, so a dump in flight
            # leaves the previous file (utils.train_dcca_pool.atomic_pickle_dump)
            try:
                if self._path:
                    os.unlink(self._path)
            except OSError:
                pass
            try:
                _remove_tune_dirs()
            except Exception:
                pass
            os._exit(75)
        act = on_dead or default_on_dead

        def loop():
            import select
            live = dict(socks)
            while live and not stop.is_set():
                try:
                    ready, _, _ = select.select(list(live), [], [], 0.2)
                except (OSError, ValueError):
                    return                                   # sockets closed under us: the hub is being shut down
                for c in ready:
                    try:
                        data = c.recv(1)
                    except OSError:
                        data = b""
                    if stop.is_set():
                        return
                    peer = live.pop(c)
                    if data != b"Q":
                        act(peer)
                        return
        for c in socks:
            c.settimeout(None)
        self._wd_stop = stop
        self._wd_thread = threading.Thread(target=loop, name="asr-hub-watchdog", daemon=True)
        self._wd_thread.start()

    def stop_watchdog(self):
        """clean end of the job on this rank: the watchdog stops and the peers are told that this is not a death"""
        t = getattr(self, "_wd_thread", None)
        if t is None:
            return
        self._wd_stop.set()
        t.join(timeout=2.0)
        self._wd_thread, self._wd_used = None, True
        for c in self._peers + ([self._sock] if self._sock else []):
            try:
                c.sendall(b"Q")
            except OSError:
                pass

    def close(self):
        import os
        if getattr(self, "_wd_thread", None) is not None:
            self.stop_watchdog()
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


class EngineComm(object):
    """The library's own communicator (Engine.comm_init: RCCL over xGMI; comm_init_custom: host callbacks) behind the
    interface of HubComm / TorchComm - rank, world, all_gather_rows (ragged), all_reduce_sum - for small host arrays:
    they are staged through device buffers and travel over asr_comm_allgather_dev / asr_comm_allreduce_dev.  What
    run_eval / refine_cca use with --gpus N, so that no second transport is involved."""

    def __init__(self, engine):
        self.engine = engine
        self.rank, self.world = engine.comm_info()

    def all_gather_rows(self, local):
        local = np.ascontiguousarray(local)
        if self.world == 1:
            return local.copy()
        counts = self.engine.allgather_host(np.array([local.shape[0]], dtype=np.int64)).ravel()
        nmax = int(counts.max())
        pad = np.zeros((nmax,) + local.shape[1:], dtype=local.dtype)
        pad[:local.shape[0]] = local
        parts = self.engine.allgather_host(pad)                  # (world, nmax, ...)
        return np.concatenate([parts[r, :int(c)] for r, c in enumerate(counts)], axis=0)

    def all_reduce_sum(self, arr):
        a = np.asarray(arr)
        if self.world == 1:
            return a.copy()
        out = self.engine.allreduce_host(a.astype(np.float64))   # integers up to 2^53 travel exactly
        return np.rint(out).astype(a.dtype) if a.dtype.kind in "iu" else out.astype(a.dtype)


def sharded_eval_retrieval(rank_fn, lv1_local, lv2_local, comm, details=False):
    """eval_retrieval (utils/train_dcca_pool.py:28-82) over a pair list sharded
    across ranks (equal list sizes n1 == n2, pairs co-located; the shards may differ in size).
    rank_fn(lv1, lv2_all, query_offset, n1_global) -> (ranks, dstar, ties).
    Returns (mean_rank, median_rank, mean_dist, hit_rates, map) - identical on
    every rank and, the ranks being integers and d* exact float64, identical to the
    one-device call on the concatenated lists - plus this rank's integer ranks
    (details=True: plus the ranks and match distances of ALL queries, in pair order)."""
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
    if details:
        return stats, ranks, all_ranks, all_dstar
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
    """This rank's rows of a batch every rank drew identically (same iterator seed): the contiguous range
    shard_range(len, rank, world) - the first len % world ranks hold one row more, NO row is dropped (the reference's
    BATCH_SIZE = 100, models/mutopia_ccal_cont.py:26, stays 100 on 3 or 8 GPUs; the library is told the size of the
    whole batch with Engine.train_set_global_batch).  Every rank needs at least one row."""
    n = int(arrays[0].shape[0])
    if n < int(world):
        raise ValueError("a batch of %d rows cannot be sharded over %d ranks (every rank needs a row)" % (n, world))
    lo, hi = shard_range(n, rank, world)
    return [np.ascontiguousarray(a[lo:hi]) for a in arrays]


def make_torch_transport(engine, comm):
    """(allreduce, allgather) host callbacks for Engine.comm_init_custom on top of a TorchComm (gloo or nccl) or a
    HubComm: the device buffer is staged through host memory.  The native path is Engine.comm_init (RCCL inside the
    library); this one serves clusters where the process group is the only transport, several ranks sharing ONE GPU
    (RCCL refuses that), and the CPU tests."""
    from . import _lib

    def allreduce(buf, count, dtype):
        dt = np.float64 if dtype == _lib.DTYPE_F64 else np.int32 if dtype == _lib.DTYPE_I32 else np.float32
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
        share_tune_cache(comm)
        engine.comm_init_custom(rank, world, *make_torch_transport(engine, comm))
        return
    if transport != "rccl":
        raise ValueError("transport must be 'rccl' or 'host'")
    if isinstance(comm, HubComm):
        share_tune_cache(comm)
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


_TUNE_DIRS = []


def _remove_tune_dirs():
    import shutil
    while _TUNE_DIRS:
        shutil.rmtree(_TUNE_DIRS.pop(), ignore_errors=True)


def share_tune_cache(comm):
    """One ASR_TUNE_CACHE file per job: rank 0 publishes the path its environment names - or creates a private file -
    and every rank adopts it.  Together with tune_in_rank_order() all ranks run the schedules rank 0 timed - the
    same float32 summation order everywhere, which data-parallel fit() relies on to keep its replicas bit-identical
    between the gradient all-reduces.  The broadcast ALWAYS runs (a rank that returned early because its own
    environment had the variable would leave the hub's exchanges one step out of line); a directory created here is
    removed when the process exits.  Works over HubComm (bcast_bytes) and TorchComm (all_gather_rows of the path)."""
    import atexit
    import os
    import tempfile
    if getattr(comm, "world", 1) <= 1:
        return os.environ.get("ASR_TUNE_CACHE")
    path = b""
    if comm.rank == 0:
        path = os.environ.get("ASR_TUNE_CACHE", "").encode()
        if not path:
            d = tempfile.mkdtemp(prefix="asr_tune_")
            if not _TUNE_DIRS:
                atexit.register(_remove_tune_dirs)
            _TUNE_DIRS.append(d)
            path = os.path.join(d, "tune_cache.txt").encode()
    if hasattr(comm, "bcast_bytes"):
        path = comm.bcast_bytes(path, src=0)
    else:                                           # TorchComm: fixed-size rows, rank 0's row wins
        row = np.zeros((1, 4096), np.uint8)
        row[0, :len(path)] = np.frombuffer(path, np.uint8)
        path = bytes(comm.all_gather_rows(row)[0]).rstrip(b"\0")
    path = path.decode()
    os.environ["ASR_TUNE_CACHE"] = path
    return path


def tune_in_rank_order(engine, barrier, rank, trigger=None):
    """Rank 0 runs `trigger` (default: a one-sample embedding of each view, which makes a fresh context time its
    convolution schedules and append them to ASR_TUNE_CACHE) before the other ranks do: they then find every line in
    the cache.  `barrier(flag)` is the job's barrier AND carries rank 0's outcome: it returns the sum of the ranks'
    float flags (HubComm: hub_flag_barrier(hub); a communicator: engine_flag_barrier(engine)).  When rank 0's trigger
    raises, every rank leaves with an error instead of waiting in a collective that has no timeout."""
    def default_trigger():
        engine.embed_view1(np.zeros((1, 1, engine.net_h1, engine.net_w1), np.float32), prepared=True)
        engine.embed_view2(np.zeros((1, 1, engine.cfg.h2, engine.cfg.w2), np.float32))
    trigger = trigger or default_trigger
    if rank == 0:
        err = None
        try:
            trigger()
        except BaseException as e:          # the others must learn about it before this rank unwinds
            err = e
        barrier(0.0 if err is None else 1.0)
        if err is not None:
            raise err
    else:
        if barrier(0.0) > 0.0:
            raise RuntimeError("rank 0 failed while timing the kernel schedules; rank %d stops" % rank)
        trigger()


def hub_flag_barrier(hub):
    """barrier(flag) -> sum of the ranks' flags over the TCP hub"""
    return lambda flag=0.0: float(sum(hub.all_gather_object(float(flag))))


def engine_flag_barrier(engine):
    """barrier(flag) -> sum of the ranks' flags over the engine's communicator (an all-reduce of one double)"""
    return lambda flag=0.0: float(engine.allreduce_host(np.array([flag], dtype=np.float64))[0])


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
