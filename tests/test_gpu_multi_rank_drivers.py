"""GPU, ONE device: the multi-GPU product surface - run_eval / refine_cca / run_train with --gpus N, bench.py
--workload train, unequal shards - driven exactly as a user drives it (one process per rank, the GPU-free parent
spawns them), with the ranks sharing this box's GPU: ASR_SAME_GPU=1 / ASR_BENCH_SAME_GPU=1 and `--comm host` (host
callbacks over the TCP hub; RCCL refuses two ranks on one device).  Everything but the transport is the code the
8-GPU job runs.  What must hold: N ranks print / write what one rank prints / writes.

Reference: run_eval.py:102-108,174; refine_cca.py:95-107; utils/train_dcca_pool.py:203-205 (all single-device)."""
import os
import pickle
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPLIT, CONFIG = "splits/all_split.yaml", "exp_configs/mutopia_full_aug.yaml"
TAG = "all_split_mutopia_full_aug"
MODEL = "mutopia_ccal_cont"


def _env(exp_root, **extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASR_TUNE_CACHE")}
    env.update(ASR_EXP_ROOT=str(exp_root), ASR_SAME_GPU="1", ASR_AUTOTUNE="0", PYTHONPATH=ROOT, ASR_HUB_TIMEOUT="120")
    env.update(extra)
    return env


def _dump_params(exp_root, sub=MODEL):
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    params = synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=True)
    d = exp_root / sub
    d.mkdir(exist_ok=True)
    with open(d / ("params_%s.pkl" % TAG), "wb") as fp:
        pickle.dump(params, fp, protocol=2)
    return params


def _module(mod, args, env, timeout=900):
    out = subprocess.run([sys.executable, "-m", "audio_sheet_retrieval_amd." + mod] + list(args), env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-5000:]
    return out


@pytest.mark.parametrize("gpus,n_test,extra", [(2, 120, []), (3, 100, ["--V2_to_V1", "--max_dim", "16"])])
def test_run_eval_on_n_ranks_prints_and_dumps_what_one_rank_does(tmp_path, gpus, n_test, extra):
    """the test pairs are sharded (3 ranks x 100 pairs: 34 / 33 / 33), candidates all-gathered over the library's
    communicator, hit counters all-reduced, ranks gathered: stdout and the dumped yaml are BYTE-identical to N = 1"""
    common = ["--model", "models/%s.py" % MODEL, "--data", "synthetic:50:50:150", "--train_split", SPLIT, "--config",
              CONFIG, "--n_test", str(n_test), "--dump_results"] + extra
    direction = "A2S" if "--V2_to_V1" in extra else "S2A"
    outs, dumps = [], []
    for n in (1, gpus):
        root = tmp_path / ("n%d" % n)
        root.mkdir()
        _dump_params(root)
        res = _module("run_eval", common + (["--gpus", str(n), "--comm", "host"] if n > 1 else []), _env(root))
        outs.append(res.stdout.replace(str(root), "<root>"))
        with open(root / MODEL / ("eval_%s_%s.yaml" % (TAG, direction)), "rb") as fp:
            dumps.append(fp.read())
    assert "Hit Rates" in outs[0] and "median rank" in outs[0]
    assert outs[0] == outs[1], "\n--- one rank ---\n%s\n--- %d ranks ---\n%s" % (outs[0], gpus, outs[1])
    assert dumps[0] == dumps[1]


def test_refine_cca_on_three_ranks_writes_the_same_file(tmp_path):
    """250 training pairs over 3 ranks (84 / 83 / 83): towers local, the tower outputs all-gathered, the same fit on
    every rank - the refined parameter file is bit-identical to the one-rank run's"""
    common = ["--model", "models/%s.py" % MODEL, "--data", "synthetic:250:50:50", "--train_split", SPLIT, "--config",
              CONFIG, "--n_train", "250"]
    files = []
    for n in (1, 3):
        root = tmp_path / ("n%d" % n)
        root.mkdir()
        params = _dump_params(root)
        _module("refine_cca", common + (["--gpus", "3", "--comm", "host"] if n > 1 else []), _env(root))
        with open(root / (MODEL + "_est_UV") / ("params_%s.pkl" % TAG), "rb") as fp:
            files.append(pickle.load(fp))
    for i in range(97):
        assert np.array_equal(files[0][i], files[1][i]), i
    assert not np.array_equal(files[0][90], params[90])            # U was re-estimated


def _bench(env, *args):
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_train_workload_two_and_three_ranks_equal_one():
    """`bench.py --workload train` (BASELINE configs[2] split N ways): the loss of the first update from the seeded
    parameters does not depend on N (1e-5), the replicas end with identical parameters, and the line says what one
    update costs in collectives.  3 ranks x batch 64 = 22 / 21 / 21 rows."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ASR_TUNE_CACHE")}
    env["ASR_AUTOTUNE"] = "0"       # the model's schedule picks in every process: the loss bar compares the exchange, not
    #                                 F(2x2) against F(4x4) rounding between separately tuned runs
    common = ("--workload", "train", "--train-batch", "64", "--steps", "2", "--warmup", "1", "--repeats", "2")
    one = _bench(env, *common)
    assert one["n_gpus"] == 1 and one["collectives_per_update"] is None and one["scaling"] == "strong"
    assert one["config"]["rows_per_gpu"] == [64] and one["value"] > 0 and 0 < one["roofline"]["frac"] < 1
    for n, rows in ((2, [32, 32]), (3, [22, 21, 21])):
        rec = _bench(dict(env, ASR_BENCH_SAME_GPU="1"), "--gpus", str(n), "--comm", "host", *common)
        assert rec["n_gpus"] == n and rec["config"]["rows_per_gpu"] == rows and rec["comm"]["rccl_ranks"] == n
        assert abs(rec["loss_first_update"] - one["loss_first_update"]) <= 1e-5, (rec["loss_first_update"],
                                                                                   one["loss_first_update"])
        assert rec["replicas_equal"] is True
        c = rec["collectives_per_update"]
        # 9 forward + 9 backward all-reduces (the two towers' BatchNorm sums travel together), the gradients; two all-gathers
        assert c["allreduce_calls"] == 19 and c["allgather_calls"] == 2
        assert c["allgather_bytes_per_rank"] == 2 * max(rows) * 128
        assert rec["torch_imported"] is False


def test_batch_100_on_three_ranks_trains_on_all_100_rows(monkeypatch):
    """models/mutopia_ccal_cont.py:26 BATCH_SIZE = 100 on 3 ranks: 34 / 33 / 33 rows, nothing dropped or padded.  Two
    updates through the host mirror's iter_funcs['train'] on every rank (threads, one context each, host-callback
    exchange) against the single-context updates on the whole batch: loss 1e-5, parameters within a fraction of the
    update, replicas identical."""
    import threading
    from audio_sheet_retrieval_amd import _lib
    from tests.test_gpu_data_parallel import HostExchange, _problem, _engine
    monkeypatch.setenv("ASR_AUTOTUNE", "0")      # the model's schedule picks in all four contexts (see test_gpu_data_parallel)
    B, world, hw1, hw2 = 100, 3, (48, 64), (32, 24)
    params, x1, x2 = _problem(B, hw1, hw2)
    ref = _engine(params, hw1, hw2)
    ref.train_begin(B)
    ref_losses = [ref.train_step(x1, x2, lr=0.002)]
    ref_H = ref.debug_train_tensor("H", view=1, batch=B).reshape(B, 32)
    trainable = [i for i in range(90) if i % 5 <= 2]
    ref_grads = dict((i, ref.debug_train_tensor("grad", index=i)) for i in trainable)
    ref_losses.append(ref.train_step(x1, x2, lr=0.002))
    ref_params = ref.get_params()
    ref.close()

    ex = HostExchange(world)
    out, errs = [None] * world, []

    def body(r):
        try:
            from audio_sheet_retrieval_amd import distributed as D
            eng = _engine(params, hw1, hw2)
            ar, ag = ex.bind(r, eng)
            eng.comm_init_custom(r, world, ar, ag)
            lo, hi = D.shard_range(B, r, world)
            eng.train_begin(34)
            with pytest.raises(_lib.AsrError):              # equal-shard default: 34 rows on rank 0 only is a mismatch
                eng.train_set_global_batch(2)               # fewer rows than ranks
            eng.train_set_global_batch(B)
            a, b = D.shard_batch([x1, x2], r, world)
            assert a.shape[0] == hi - lo == (34 if r == 0 else 33)
            losses = [eng.train_step(a, b, lr=0.002)]
            H = eng.debug_train_tensor("H", view=1, batch=hi - lo).reshape(hi - lo, 32)
            grads = dict((i, eng.debug_train_tensor("grad", index=i)) for i in trainable)
            losses.append(eng.train_step(a, b, lr=0.002))
            p = eng.get_params()
            eng.train_end()
            eng.comm_destroy()
            eng.close()
            out[r] = (losses, p, H, grads)
        except BaseException as e:          # noqa: BLE001
            errs.append(e)
            ex.barrier.abort()
    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errs:
        raise errs[0]
    Hcat = np.concatenate([out[r][2] for r in range(world)])
    # tower outputs of the first step: BatchNorm used the statistics of all 100 rows on every rank
    assert np.abs(Hcat - ref_H).max() <= 1e-5 * max(1.0, float(np.abs(ref_H).max()))
    for r in range(world):
        losses, p, _, grads = out[r]
        for step_no, ((l, c), (rl, rc)) in enumerate(zip(losses, ref_losses)):
            assert abs(l - rl) <= 1e-5, (r, l, rl)
            # canonical correlations: 1e-4 from the same parameters; the second update starts from parameters that
            # already differ by Adam's noise-sized steps (measured 1.0e-4 there)
            assert np.abs(c - rc).max() <= (1e-4 if step_no == 0 else 3e-4)
        # the all-reduced gradient of the first update IS the single-context gradient (float32 summation order aside)
        # (block 9's beta has a zero gradient in exact arithmetic - the CCALayer subtracts the batch mean - so its
        # tensor is float32 noise: errors are scaled by the tensor's maximum, floored at 1e-3 of the largest gradient)
        gmax = max(float(np.abs(g).max()) for g in ref_grads.values())
        for i in trainable:
            err = float(np.abs(grads[i] - ref_grads[i]).max()) / max(1e-3 * gmax, float(np.abs(ref_grads[i]).max()))
            assert err <= 2e-4, (r, i, err)
        # the parameters after two Adam updates: Adam normalises every element's step to ~lr, so an element whose
        # gradient is float32 noise around zero may move by lr in either direction - a handful of elements, bounded
        # by the two updates themselves; everything else within a fraction of the update
        n_off = n_all = 0
        for i in range(90):
            d = np.abs(p[i] - ref_params[i])
            tol = 3e-4 * max(1.0, float(np.abs(ref_params[i]).max()))
            n_off += int(np.count_nonzero(d > tol))
            n_all += d.size
            assert d.max() <= 2.2 * 2 * 0.002, (r, i, float(d.max()))
        assert n_off <= 1e-3 * n_all, (r, n_off, n_all)
        for i in range(97):
            assert np.array_equal(p[i], out[0][1][i]), i


def test_run_train_on_two_ranks_and_a_killed_rank_ends_the_job(tmp_path):
    """run_train --gpus 3 (BATCH_SIZE 100 = 34 + 33 + 33 rows) trains an epoch and writes rank 0's files; a rank killed
    in the middle of training ends the whole job with a non-zero code in well under 15 s"""
    import psutil
    common = ["--model", "models/%s.py" % MODEL, "--data", "synthetic:300:100:100", "--train_split", SPLIT,
              "--config", CONFIG, "--comm", "host"]
    root = tmp_path / "ok"
    root.mkdir()
    res = _module("run_train", common + ["--gpus", "3", "--max_epochs", "1"], _env(root))
    assert "Epoch 1 of 1" in res.stdout and res.stdout.count("Epoch 1 of 1") == 3      # every rank follows rank 0
    params = pickle.load(open(root / MODEL / ("params_%s.pkl" % TAG), "rb"))
    assert len(params) == 97 and all(np.isfinite(p).all() for p in params)

    root = tmp_path / "killed"
    root.mkdir()
    log = open(tmp_path / "killed.log", "w")
    proc = subprocess.Popen([sys.executable, "-u", "-m", "audio_sheet_retrieval_amd.run_train"] + common +
                            ["--gpus", "2", "--max_epochs", "50"], env=_env(root), cwd=ROOT,
                            stdout=log, stderr=subprocess.STDOUT)
    try:
        victim, deadline = None, time.time() + 300
        while victim is None and time.time() < deadline:
            assert proc.poll() is None, open(tmp_path / "killed.log").read()[-3000:]
            text = open(tmp_path / "killed.log").read()
            if "ups:" in text:                                    # updates are running
                for c in psutil.Process(proc.pid).children():
                    try:
                        if c.environ().get("RANK") == "1":
                            victim = c
                    except psutil.Error:
                        pass
            time.sleep(0.2)
        assert victim is not None, "training never started"
        t0 = time.time()
        victim.kill()
        code = proc.wait(timeout=60)
        took = time.time() - t0
    finally:
        if proc.poll() is None:
            proc.kill()
        log.close()
    assert code != 0 and took < 15, (code, took)
