"""CPU, world_size 2, NO torch: the TCP hub that carries the control plane of bench.py / run_train.py
(distributed.HubComm: rendezvous file keyed by the launcher's pid, gather-at-rank-0 exchanges) and the sharded
retrieval logic on top of it give the same integers as the single-process oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
from audio_sheet_retrieval_amd import distributed as D
from oracle import retrieval as oret
assert "torch" not in sys.modules
hub = D.HubComm()
rank, world = hub.rank, hub.world
n = %(n)d
rng = np.random.default_rng(123)
def unit(m, d=32):
    x = rng.standard_normal((m, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
a, b = unit(n), unit(n)
b = (b + 1.2 * a).astype(np.float32)
lo, hi = D.shard_range(n, rank, world)
def rank_fn(lv1, lv2_all, off, n_glob):
    d = oret.cdist_cosine64(lv1, lv2_all)
    k, h = oret.k_h(n_glob, lv2_all.shape[0])
    return oret.ranks_by_counting(d, k=k, h=h, query_offset=off)
stats, ranks = D.sharded_eval_retrieval(rank_fn, a[lo:hi], b[lo:hi], hub)
# primitives
got = hub.bcast_bytes(b"id-from-rank-0" if rank == 0 else b"", src=0)
assert got == b"id-from-rank-0"
assert hub.all_reduce_max(1.0 + rank) == float(world)
assert hub.all_reduce_sum(np.array([1, 2])).tolist() == [world, 2 * world]
objs = hub.all_gather_object(dict(r=rank))
assert [o["r"] for o in objs] == list(range(world))
for _ in range(50):
    hub.barrier()
np.savez(os.path.join(%(out)r, "r%%d.npz" %% rank), ranks=ranks, lo=lo, hi=hi,
         stats=np.array([stats[0], stats[1], stats[2], stats[4]] + [stats[3][k] for k in (1, 5, 10, 25)]))
hub.close()
assert "torch" not in sys.modules
"""


@pytest.mark.parametrize("world,n", [(2, 101), (3, 64)])
def test_hub_sharded_eval_matches_single_process(tmp_path, world, n):
    from oracle import retrieval as oret
    code = WORKER % dict(root=ROOT, n=n, out=str(tmp_path))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29%03d" % (os.getpid() % 1000))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    rng = np.random.default_rng(123)

    def unit(m, d=32):
        x = rng.standard_normal((m, d)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    a, b = unit(n), unit(n)
    b = (b + 1.2 * a).astype(np.float32)
    ref = oret.eval_retrieval(a, b)
    ranks_ref, _, _ = oret.ranks_by_counting(oret.cdist_cosine64(a, b))
    got = np.zeros(n, np.int32)
    for r in range(world):
        z = np.load(tmp_path / ("r%d.npz" % r))
        got[int(z["lo"]):int(z["hi"])] = z["ranks"]
        st = z["stats"]
        assert st[0] == ref[0] and st[1] == ref[1] and st[3] == ref[4]
        assert [int(v) for v in st[4:]] == [ref[3][k] for k in (1, 5, 10, 25)]
    assert np.array_equal(got, ranks_ref)
    # the rendezvous file is gone after close()
    import tempfile
    assert not [f for f in os.listdir(tempfile.gettempdir()) if f.startswith("asr_hub_%d_" % os.getpid())]


def test_bench_parent_never_loads_the_gpu_library():
    """`python bench.py --gpus N` must start its ranks from a process that has not touched the GPU: the spawner path
    imports neither the ctypes binding nor torch (checked by running it with a stub child command)."""
    code = ("import sys, os; sys.path.insert(0, %r); sys.argv = ['bench.py', '--gpus', '2']\n"
            "import bench, subprocess\n"
            "calls = []\n"
            "class P:\n"
            "    def __init__(self, cmd, env=None):\n"
            "        calls.append((cmd, {k: env[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))\n"
            "    def wait(self): return 0\n"
            "bench.subprocess.Popen = P\n"
            "try:\n"
            "    bench.main()\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0\n"
            "assert len(calls) == 2 and [c[1]['RANK'] for c in calls] == ['0', '1']\n"
            "assert all(c[1]['WORLD_SIZE'] == '2' and c[1]['MASTER_ADDR'] == '127.0.0.1' for c in calls)\n"
            "assert calls[0][1]['MASTER_PORT'] == calls[1][1]['MASTER_PORT']\n"
            "assert 'torch' not in sys.modules and 'audio_sheet_retrieval_amd._lib' not in sys.modules\n"
            "print('OK')\n") % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.stdout.strip().endswith("OK"), out.stdout + out.stderr
