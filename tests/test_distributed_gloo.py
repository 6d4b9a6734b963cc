"""CPU, world_size 2, gloo: the N>1 exchange logic (sharding, all-gather of
candidate embeddings, query offsets, hit all-reduce, sharded top-k merge) gives
the same integers as the single-process oracle.  The rank/top-k arithmetic is
supplied by the oracle here; on the GPU box bench.py plugs in Engine.rank."""
import os
import sys
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unit(rng, n, d=32):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _worker(rank, world, port, n, out_dir):
    import torch.distributed as dist
    from audio_sheet_retrieval_amd import distributed as D
    from oracle import retrieval as oret
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm()
        rng = np.random.default_rng(123)
        a, b = _unit(rng, n), _unit(rng, n)
        b = (b + 1.2 * a).astype(np.float32)
        lo, hi = D.shard_range(n, rank, world)

        def rank_fn(lv1, lv2_all, off, n_glob):
            d = oret.cdist_cosine64(lv1, lv2_all)
            k, h = oret.k_h(n_glob, lv2_all.shape[0])
            return oret.ranks_by_counting(d, k=k, h=h, query_offset=off)

        stats, ranks = D.sharded_eval_retrieval(rank_fn, a[lo:hi], b[lo:hi], comm)

        def topk_fn(db, q, k, off):
            idx, dist_ = oret.topk(db, q, k)
            return (idx + off).astype(np.int32), dist_

        db = _unit(rng, 301)
        dlo, dhi = D.shard_range(301, rank, world)
        qlo, qhi = D.shard_range(9, rank, world)
        q = _unit(rng, 9)
        tidx, tdist = D.sharded_topk(topk_fn, db[dlo:dhi], q[qlo:qhi], 25, comm)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), ranks=ranks, lo=lo, hi=hi, stats=np.array(
            [stats[0], stats[1], stats[2], stats[4]] + [stats[3][k] for k in (1, 5, 10, 25)]),
            tidx=tidx, tdist=tdist, qlo=qlo, qhi=qhi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64])
def test_sharded_eval_and_topk_match_single_process(tmp_path, n):
    from audio_sheet_retrieval_amd import distributed as D
    from oracle import retrieval as oret
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    a, b = _unit(rng, n), _unit(rng, n)
    b = (b + 1.2 * a).astype(np.float32)
    ref = oret.eval_retrieval(a, b)
    ranks_ref, _, _ = oret.ranks_by_counting(oret.cdist_cosine64(a, b))
    db = _unit(rng, 301)
    q = _unit(rng, 9)
    tidx_ref, tdist_ref = oret.topk(db, q, 25)
    got = np.zeros(n, np.int32)
    for r in range(world):
        z = np.load(tmp_path / ("r%d.npz" % r))
        assert (int(z["lo"]), int(z["hi"])) == D.shard_range(n, r, world)
        got[int(z["lo"]):int(z["hi"])] = z["ranks"]
        s = z["stats"]
        assert s[0] == ref[0] and s[1] == ref[1] and abs(s[2] - ref[2]) < 1e-15 and s[3] == ref[4]
        assert [int(v) for v in s[4:]] == [ref[3][k] for k in (1, 5, 10, 25)]
        assert np.array_equal(z["tidx"], tidx_ref[int(z["qlo"]):int(z["qhi"])])
        assert np.array_equal(z["tdist"], tdist_ref[int(z["qlo"]):int(z["qhi"])])
    assert np.array_equal(got, ranks_ref)


def test_shard_range_covers_everything():
    from audio_sheet_retrieval_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---- data-parallel training exchange (SURVEY 8e "Training partitioning") over gloo ------------------------------
class _HostEngine(object):
    """stands in for Engine in make_torch_transport: 'device' buffers are host arrays addressed by a handle"""

    def __init__(self):
        self.mem = {}

    def put(self, arr):
        h = 1000 + len(self.mem)
        self.mem[h] = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        return h

    def raw_download(self, ptr, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.mem[ptr][:n].view(dtype).reshape(shape).copy()

    def raw_upload(self, ptr, arr):
        b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        self.mem[ptr][:b.size] = b


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from audio_sheet_retrieval_amd import _lib, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm()
        eng = _HostEngine()
        allreduce, allgather = D.make_torch_transport(eng, comm)
        rng = np.random.default_rng(5)
        z = rng.standard_normal((48, 7, 12)).astype(np.float32)          # (batch, pixels, channels), same on all ranks
        H = rng.standard_normal((48, 32)).astype(np.float32)
        (zl, Hl) = D.shard_batch([z, H], rank, world)
        # BatchNorm statistics of the FULL batch from all-reduced float64 column sums (train_fwd_kernels.hip)
        sums = np.concatenate([zl.astype(np.float64).sum((0, 1)), (zl.astype(np.float64) ** 2).sum((0, 1))])
        hs = eng.put(sums)
        assert allreduce(hs, sums.size, _lib.DTYPE_F64) == 0
        tot = eng.raw_download(hs, (24,), np.float64)
        count = zl.shape[0] * zl.shape[1] * world
        mu, var = tot[:12] / count, tot[12:] / count - (tot[:12] / count) ** 2
        # tower outputs all-gathered in rank order
        hsend, hrecv = eng.put(Hl), eng.put(np.zeros_like(H))
        assert allgather(hsend, hrecv, Hl.nbytes) == 0
        Hall = eng.raw_download(hrecv, H.shape, np.float32)
        # parameter gradients: float32 sum over ranks
        g = (zl.sum((0, 1)) * (rank + 1)).astype(np.float32)
        hg = eng.put(g)
        assert allreduce(hg, g.size, _lib.DTYPE_F32) == 0
        gsum = eng.raw_download(hg, (12,), np.float32)
        np.savez(os.path.join(out_dir, "dp%d.npz" % rank), mu=mu, var=var, Hall=Hall, gsum=gsum, g=g)
    finally:
        dist.destroy_process_group()


def test_data_parallel_exchange_over_gloo(tmp_path):
    world = 2
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    z = rng.standard_normal((48, 7, 12)).astype(np.float32)
    H = rng.standard_normal((48, 32)).astype(np.float32)
    r = [np.load(os.path.join(str(tmp_path), "dp%d.npz" % k)) for k in range(world)]
    z64 = z.astype(np.float64)
    for k in range(world):
        assert np.allclose(r[k]["mu"], z64.mean((0, 1)), rtol=0, atol=1e-12)
        assert np.allclose(r[k]["var"], z64.var((0, 1)), rtol=0, atol=1e-12)
        assert np.array_equal(r[k]["Hall"], H)                       # rank order == batch order
        assert np.allclose(r[k]["gsum"], r[0]["g"] + r[1]["g"], rtol=1e-6)
    assert np.array_equal(r[0]["gsum"], r[1]["gsum"])                # every rank applies the same update


def test_shard_batch_keeps_every_row():
    """a batch that is not a multiple of the world size is NOT truncated: contiguous shards, the first ones a row longer
    (the reference's BATCH_SIZE = 100 on 3 or 8 GPUs, models/mutopia_ccal_cont.py:26)"""
    from audio_sheet_retrieval_amd import distributed as D
    x = np.arange(50).reshape(25, 2)
    a = D.shard_batch([x], 0, 2)[0]
    b = D.shard_batch([x], 1, 2)[0]
    assert a.shape == (13, 2) and b.shape == (12, 2) and np.array_equal(np.concatenate([a, b]), x)
    y = np.arange(100)
    for world in (3, 8):
        parts = [D.shard_batch([y], r, world)[0] for r in range(world)]
        assert np.array_equal(np.concatenate(parts), y)
        assert max(len(p) for p in parts) - min(len(p) for p in parts) == 1
        assert [len(p) for p in parts] == sorted((len(p) for p in parts), reverse=True)
        for r in range(world):
            lo, hi = D.shard_range(100, r, world)
            assert np.array_equal(parts[r], y[lo:hi])
    with pytest.raises(ValueError):
        D.shard_batch([np.arange(2)], 0, 3)


class _FakeCommEngine(object):
    """the slice of Engine that distributed.EngineComm uses (comm_info, allgather_host, allreduce_host), over threads"""
    import threading as _th
    _barrier, _slots = None, None

    def __init__(self, rank, world, barrier, slots):
        self.rank, self.world, self._barrier, self._slots = rank, world, barrier, slots

    def comm_info(self):
        return self.rank, self.world

    def allgather_host(self, arr):
        self._slots[self.rank] = np.ascontiguousarray(arr).copy()
        self._barrier.wait()
        out = np.stack([self._slots[r] for r in range(self.world)])
        self._barrier.wait()
        return out

    def allreduce_host(self, values):
        parts = self.allgather_host(np.asarray(values, dtype=np.float64))
        return parts.sum(axis=0)


def test_engine_comm_ragged_gather_and_sharded_eval_over_three_ranks():
    """distributed.EngineComm (what run_eval / refine_cca --gpus N use): ragged all_gather_rows = padded equal-size
    all-gather + compaction in rank order, integer all-reduce exact; sharded_eval_retrieval over it gives the
    single-process numbers for 3 ranks x 100 pairs (34 / 33 / 33)"""
    import threading
    from audio_sheet_retrieval_amd import distributed as D
    from oracle import retrieval as oret
    world, n = 3, 100
    rng = np.random.default_rng(8)
    a = rng.standard_normal((n, 32)).astype(np.float32)
    b = (a + 0.8 * rng.standard_normal((n, 32))).astype(np.float32)
    ref = oret.eval_retrieval(a, b)
    ref_ranks, ref_dstar, _ = oret.ranks_by_counting(oret.cdist_cosine64(a, b))
    barrier, slots, out, errs = threading.Barrier(world), [None] * world, [None] * world, []

    def body(r):
        try:
            comm = D.EngineComm(_FakeCommEngine(r, world, barrier, slots))
            lo, hi = D.shard_range(n, r, world)
            rows = comm.all_gather_rows(np.arange(lo, hi, dtype=np.int64)[:, None])
            assert np.array_equal(rows.ravel(), np.arange(n))
            assert comm.all_reduce_sum(np.array([r + 1, 10], dtype=np.int64)).tolist() == [6, 30]

            def rank_fn(q, c_all, off, n_glob):
                d = oret.cdist_cosine64(q, c_all)
                k, h = oret.k_h(n_glob, c_all.shape[0])
                return oret.ranks_by_counting(d, k=k, h=h, query_offset=off)
            out[r] = D.sharded_eval_retrieval(rank_fn, a[lo:hi], b[lo:hi], comm, details=True)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            barrier.abort()
    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    for r in range(world):
        stats, _, all_ranks, all_dstar = out[r]
        assert stats[0] == ref[0] and stats[1] == ref[1] and stats[2] == ref[2] and stats[3] == ref[3] and stats[4] == ref[4]
        assert np.array_equal(all_ranks, ref_ranks) and np.array_equal(all_dstar, ref_dstar)


def test_launch_environment_helpers(monkeypatch):
    from audio_sheet_retrieval_amd import launch
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASR_SAME_GPU", "ASR_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    assert launch.world_from_env() == (0, 0, 1) and launch.device_for(3) == 3
    monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "1"); monkeypatch.setenv("WORLD_SIZE", "8")
    assert launch.world_from_env() == (5, 1, 8)
    monkeypatch.setenv("ASR_SAME_GPU", "1")
    assert launch.device_for(1) == 0
    monkeypatch.delenv("ASR_SAME_GPU")
    monkeypatch.setenv("ASR_DEVICE", "6")
    assert launch.device_for(1) == 6
    assert launch.spawn_ranks([sys.executable, "-c", "import os, sys; sys.exit(0 if os.environ['WORLD_SIZE'] == '2' else 3)"], 2) == 0
