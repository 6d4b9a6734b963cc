"""One process per GPU: the spawner and the join step shared by run_train / run_eval / refine_cca / bench.py
(the reference is single-process, utils/train_dcca_pool.py:203-205 and run_eval.py:102-108,174 run on
one device).  Nothing here imports PyTorch and the spawning process never touches a GPU.

    spawn_ranks(cmd, n)          N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment),
                                 polled: when one exits non-zero the others are terminated and its code is returned
                                 within seconds - a dead rank never leaves the rest sitting in a collective.
    join(engine, ...)            in a rank: control-plane hub, the library's communicator (RCCL, or host callbacks
                                 over the hub for several ranks on ONE GPU), one tune cache for the job, rank 0 times
                                 the kernel schedules first; with RCCL the hub stays open as a dead-peer watchdog.
"""
from __future__ import annotations

import os
import subprocess
import sys
import time


def world_from_env():
    """(rank, local_rank, world) a launcher put into the environment (defaults: a single process)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def device_for(local_rank):
    """ASR_SAME_GPU=1 (tests: several ranks on a one-GPU box, needs the host transport) puts every rank on device 0;
    ASR_DEVICE overrides for a rank somebody else launched (spawn_ranks removes it from its children's environment and
    an inherited value is ignored there: one exported ASR_DEVICE would otherwise stack all ranks on one GPU)"""
    if os.environ.get("ASR_SAME_GPU", "0") == "1":
        return 0
    if os.environ.get("ASR_SPAWNED") == "1":
        return int(local_rank)
    return int(os.environ.get("ASR_DEVICE", str(local_rank)))


def spawn_ranks(cmd, n, extra_env=None, on_failure=None, grace=5.0):
    """Start `cmd` (argv list) n times, one rank each, and wait.  Returns 0 when every rank exits 0; otherwise the
    first failing rank's exit code (1 for a signal) after terminating the others (SIGTERM, SIGKILL after `grace`
    seconds).  Fresh children only - this process has not initialised a GPU and never replaces itself.
    on_failure(rank, code): called once before returning (bench.py prints its error line there)."""
    # No port is probed here (binding port 0, closing the socket and handing the number on is a race: somebody else may
    # own it a millisecond later).  The job's ranks rendezvous through a file named by a random per-job key
    # (ASR_HUB_KEY); rank 0's hub binds port 0 ITSELF and publishes what it got there (distributed.HubComm).
    # MASTER_PORT stays in the environment for launchers' sake and is derived from the key, never bound by us.
    import secrets
    key = "%d_%s" % (os.getpid(), secrets.token_hex(8))
    port = 20000 + int(key[-4:], 16) % 20000
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ASR_SPAWNED="1", ASR_HUB_KEY=key)
        # a single-GPU user's exported ASR_DEVICE must not put every rank of a multi-GPU job on that one device
        # (device_for honours it only for externally launched ranks)
        env.pop("ASR_DEVICE", None)
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(cmd), env=env))
    failed = None
    try:
        while failed is None:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = bad[0]
            elif all(c == 0 for c in codes):
                return 0
            else:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    if on_failure is not None:
        on_failure(*failed)
    else:
        sys.stderr.write("rank %d exited with code %s; the other ranks were terminated\n" % failed)
    return failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1


def join(engine, transport="rccl", tune=True, watchdog=True):
    """This process is rank RANK of WORLD_SIZE: give `engine` the job's communicator.  Returns the HubComm (rank,
    world on it).  transport "rccl": the hub carries the communicator id, then - with `watchdog` - stays open and
    ends this process when a peer's connection drops (RCCL collectives have no timeout: without it a rank that dies
    in epoch 3 leaves the others hung in an all-reduce).  transport "host": the hub is the data plane itself
    (host callbacks; a dead peer is a HubError in the next collective)."""
    from . import distributed
    hub = distributed.HubComm()
    distributed.init_data_parallel(engine, transport=transport, comm=hub)      # also: one tune cache for the job
    if tune:
        distributed.tune_in_rank_order(engine, distributed.hub_flag_barrier(hub), hub.rank)
    if transport == "rccl" and watchdog and hub.world > 1:
        hub.start_watchdog()
    return hub


def leave(hub):
    """end of the job on this rank: clean goodbye to the watchdogs of the peers, close the hub"""
    if hub is None:
        return
    try:
        if getattr(hub, "_wd_thread", None) is not None:
            hub.stop_watchdog()
        elif hub.world > 1:
            hub.barrier()
    finally:
        hub.close()
