#!/usr/bin/env python
"""bench.py - snippet-pairs/sec embedded+ranked (32-d CCA) on N MI355X GPUs.

One "step" = the hot path of BASELINE.json configs[1] over one batch:
  1000 synthetic (sheet uint8 1x160x200, spectrogram f32 1x92x42) pairs per GPU,
  already resident in HBM  ->  both towers (mutopia_ccal_cont, deterministic)
  -> CCA projection -> L2 norm  ->  [N>1: all-gather of the candidate
  embeddings over RCCL]  ->  float64 all-pairs cosine ranking of this GPU's
  1000 queries against all N*1000 candidates (integer ranks).
Weak scaling: per-GPU work is fixed; value = N*1000*K / max-over-ranks time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline":     dominant kernel, algorithmic FLOP / measured HIP-event time
                  vs the dense fp32-MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md)
  "cpu_baseline": the CPU oracle (oracle/, a NumPy/C port of the reference path)
                  timed on a bounded sample on this host's cores (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRS_PER_GPU = 1000
MODEL = os.environ.get("ASR_BENCH_MODEL", "mutopia_ccal_cont")     # the headline workload; _rsz for side measurements
FLOP_PER_PAIR = 552594048 if MODEL.endswith("_rsz") else 425302464   # BASELINE.md section 2 (conv MACs x 2, both towers)
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md, dense fp32 matrix
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="samples per internal launch (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the informational single-stream pass")
    ap.add_argument("--comm", choices=["torch", "native"], default="torch",
                    help="N>1 exchange: torch.distributed all-gather (RCCL) or the library's own RCCL communicator")
    ap.add_argument("--cpu-pairs", type=int, default=2000, help="sample size of the CPU baseline leg")
    return ap.parse_args()


CPU_THREADS = 16      # the oracle's OpenMP/BLAS loops scale to ~16 threads on the EPYC host, then degrade


def _cpu_baseline_worker(n_pairs, seed):
    """Oracle (CPU port of the reference path) on a bounded sample: embed n_pairs in chunks of 100 like
    run_eval.py:107, float64 cdist + per-row argsort like utils/train_dcca_pool.py:28-82."""
    from audio_sheet_retrieval_amd.utils import synth_data
    from oracle import network as onet, retrieval as oret
    sheet, spec = synth_data.synth_pairs(np.arange(n_pairs), seed=seed)
    params = synth_data.synth_params(onet.param_shapes(MODEL), seed=1, trained_like=True)
    warm = onet.prepare(sheet[:8], MODEL)
    onet.compute_output(warm, spec[:8], params)               # thread pools, page faults
    t0 = time.perf_counter()
    lv1, lv2 = [], []
    for s in range(0, n_pairs, 100):
        x = onet.prepare(sheet[s:s + 100], MODEL)
        a, b = onet.compute_output(x, spec[s:s + 100], params)
        lv1.append(a)
        lv2.append(b)
    lv1, lv2 = np.vstack(lv1), np.vstack(lv2)
    oret.eval_retrieval(lv1, lv2)
    return time.perf_counter() - t0


def cpu_baseline(n_pairs, seed):
    """Runs the oracle in a child process pinned to CPU_THREADS OpenMP/BLAS threads (thread counts are fixed at
    library load) and reports pairs/s of the bounded sample."""
    import subprocess
    threads = max(1, min(CPU_THREADS, len(os.sched_getaffinity(0))))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads),
               MKL_NUM_THREADS=str(threads))
    code = "import bench; print('CPU_BASELINE_S', bench._cpu_baseline_worker(%d, %d))" % (n_pairs, seed)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    dt = None
    for line in out.stdout.splitlines():
        if line.startswith("CPU_BASELINE_S"):
            dt = float(line.split()[1])
    if dt is None:
        raise RuntimeError("cpu baseline failed: " + out.stderr[-2000:])
    return dict(value=n_pairs / dt, unit="pairs/s", cores=threads, kind="port",
                sample="%d pairs embedded (chunks of 100) + %dx%d float64 cdist/argsort ranking, %.1f s of CPU work"
                       % (n_pairs, n_pairs, n_pairs, dt))


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d"
                             % (args.gpus, args.gpus))
        args.gpus = world

    from audio_sheet_retrieval_amd import _lib
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes

    dist = None
    torch = None
    # ASR_BENCH_FORCE_DIST=1 under `torch.distributed.run --nproc-per-node 1` exercises the RCCL hand-off on one GPU
    use_dist = world > 1 or os.environ.get("ASR_BENCH_FORCE_DIST", "0") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n = args.pairs
    if "ASR_TUNE_CACHE" not in os.environ:          # the isolated pass below re-uses the tuner's choices
        import tempfile
        os.environ["ASR_TUNE_CACHE"] = os.path.join(tempfile.mkdtemp(prefix="asr_bench_"), "tune_rank%d.txt" % rank)
    eng = _lib.Engine(MODEL, device=local_rank, max_chunk=args.chunk)
    native = use_dist and args.comm == "native"
    if native:
        # the library's own RCCL communicator (asr_comm_init): the all-gather is enqueued on the library's stream,
        # no hand-off to torch inside the step; torch.distributed only carries the id and the timing barrier
        from audio_sheet_retrieval_amd import distributed as D
        if world == 1:
            os.environ["ASR_COMM_FORCE"] = "1"
        D.init_data_parallel(eng, rank, world, transport="rccl")
    eng.set_params(synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=True))

    # ---- synthetic shard of this rank, resident in HBM before the timed region
    first = rank * n
    sheet_u8, spec = synth_data.synth_pairs(np.arange(first, first + n), seed=23)
    d_sheet = eng.alloc(sheet_u8.nbytes).upload(sheet_u8)
    d_spec = eng.alloc(spec.nbytes).upload(spec)
    d_lv1 = eng.alloc(n * 32 * 4)
    d_ranks = eng.alloc(n * 4)
    d_dstar = eng.alloc(n * 8)
    d_ties = eng.alloc(n * 4)
    if native:
        d_lv2 = eng.alloc(n * 32 * 4)
        d_all = eng.alloc(world * n * 32 * 4)
        lv2_ptr, all_ptr = d_lv2.ptr, d_all.ptr
    elif use_dist:
        t_lv2 = torch.empty((n, 32), dtype=torch.float32, device="cuda")
        t_all = torch.empty((world * n, 32), dtype=torch.float32, device="cuda")
        lv2_ptr, all_ptr = t_lv2.data_ptr(), t_all.data_ptr()
    else:
        d_lv2 = eng.alloc(n * 32 * 4)
        lv2_ptr = all_ptr = d_lv2.ptr

    def step():
        eng.embed_view1_dev(d_sheet.ptr, _lib.IN_U8_RAW, n, d_lv1.ptr)
        eng.embed_view2_dev(d_spec.ptr, n, lv2_ptr)
        if native:
            eng.rank_sharded_dev(d_lv1.ptr, lv2_ptr, n, all_ptr, d_ranks.ptr, d_dstar.ptr, d_ties.ptr)
            return
        if use_dist:
            eng.sync()                                   # lib stream -> torch stream hand-off
            dist.all_gather_into_tensor(t_all, t_lv2)    # RCCL over xGMI
            torch.cuda.synchronize()
        eng.rank_dev(d_lv1.ptr, n, all_ptr, world * n, d_ranks.ptr, d_dstar.ptr, d_ties.ptr,
                     query_offset=rank * n, n1_global=world * n)

    def fence():
        eng.sync()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_reset()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    if use_dist:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.profile_enable(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dt = float(t.item())

    ranks = d_ranks.download((n,), np.int32)
    ties = d_ties.download((n,), np.int32)
    hits = np.array([np.count_nonzero(ranks <= k) for k in (1, 5)], dtype=np.int64)
    if use_dist:
        th = torch.from_numpy(hits).cuda()
        dist.all_reduce(th)
        hits = th.cpu().numpy()
    prof = eng.profile()

    # ---- only with ASR_TWO_STREAMS=1 (towers overlapping on two streams, which stretches every kernel's own duration):
    # the same kernels WITHOUT the other tower sharing the GPU, a few untimed steps, reported next to the timed figures.
    # By default the library runs both towers on one stream and the timed region already shows stand-alone durations.
    iso = None
    if rank == 0 and world == 1 and not use_dist and not args.no_isolated and os.environ.get("ASR_TWO_STREAMS") == "1":
        os.environ["ASR_TWO_STREAMS"] = "0"
        eng2 = _lib.Engine(MODEL, device=local_rank, max_chunk=args.chunk)
        os.environ["ASR_TWO_STREAMS"] = "1"
        eng2.set_params(synth_data.synth_params(param_shapes(MODEL), seed=1, trained_like=True))
        e_lv1, e_lv2 = eng2.alloc(n * 128), eng2.alloc(n * 128)
        e_sheet = eng2.alloc(sheet_u8.nbytes).upload(sheet_u8)
        e_spec = eng2.alloc(spec.nbytes).upload(spec)
        for it in range(6):
            if it == 1:
                eng2.sync(); eng2.profile_reset(); eng2.profile_enable(True)
            eng2.embed_view1_dev(e_sheet.ptr, _lib.IN_U8_RAW, n, e_lv1.ptr)
            eng2.embed_view2_dev(e_spec.ptr, n, e_lv2.ptr)
        eng2.sync(); eng2.profile_enable(False)
        iso = {}
        for p in eng2.profile():
            if p["launches"] > 0:
                a = iso.setdefault(p["symbol"] or p["name"], dict(ms=0.0, launches=0, flops=0.0))
                a["ms"] += p["total_ms"]; a["launches"] += p["launches"]; a["flops"] += p["flops"] * p["launches"]
        eng2.close()

    if rank == 0:
        total_pairs = world * n * args.steps
        value = total_pairs / dt
        recs = [p for p in prof if p["launches"] > 0]
        # aggregate per kernel SYMBOL, exactly like `rocprofv3 --kernel-trace --stats` does: one template
        # instantiation serves the same block of both towers (e.g. conv2 of view 1 and of view 2)
        by_sym = {}
        for p in recs:
            a = by_sym.setdefault(p["symbol"] or p["name"], dict(ms=0.0, launches=0, flops=0.0, labels=[]))
            a["ms"] += p["total_ms"]
            a["launches"] += p["launches"]
            a["flops"] += p["flops"] * p["launches"]
            a["labels"].append(p["name"])
        dom_sym, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
        avg_s = dom["ms"] / dom["launches"] * 1e-3
        achieved = dom["flops"] / dom["launches"] / avg_s / 1e12
        conv_ms = sum(p["total_ms"] for p in recs if p["name"].startswith("conv") or p["name"].startswith("tail"))
        conv_fl = sum(p["flops"] * p["launches"] for p in recs
                      if p["name"].startswith("conv") or p["name"].startswith("tail"))
        traffic, traffic_symbol = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic_by_symbol.json")
        if os.path.exists(tpath) and n == PAIRS_PER_GPU and (eng.cfg.max_chunk or 1000) == 1000:
            # PMC counters cannot be read from inside this process: the value is the committed rocprofv3
            # measurement of this same command (tools/pmc_traffic.sh), bytes per launch of the dominant symbol
            with open(tpath) as fp:
                table = json.load(fp)["kernels"]
            traffic = table.get(dom_sym, {}).get("hbm_bytes_per_launch")
            if traffic is None:
                # the tuner may have picked another tiling of the same block (same kernel, same C_in / C_out / pool):
                # its HBM traffic differs by the halo share only - report that measurement and say which symbol it is
                import re
                fam = re.match(r"(void asr::\w+<\d+, \d+, \w+),", dom_sym)
                if fam:
                    for sym, rec in table.items():
                        if sym.startswith(fam.group(1) + ",") and rec.get("hbm_bytes_per_launch"):
                            traffic, traffic_symbol = rec["hbm_bytes_per_launch"], sym
                            break
        out = {
            "metric": "snippet-pairs/sec embedded+ranked (32-d CCA)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: twin-CNN fwd (%s) + 32-d CCA embed + all-pairs cosine ranking, "
                                   "%d pairs per GPU, %d candidates" % (MODEL, n, world * n),
                       "pairs_per_gpu": n, "candidates": world * n, "chunk": eng.cfg.max_chunk or 1000,
                       "partitioning": "pairs sharded by rank; all-gather of candidate embeddings"
                       if world > 1 else "single GPU"},
            "recall_at_1": float(hits[0]) / (world * n), "recall_at_5": float(hits[1]) / (world * n),
            "rank_ties": int(ties.sum()),
            "roofline": {"bound": "mfma", "kernel": dom_sym, "layers": dom["labels"], "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic, **({"traffic_symbol": traffic_symbol} if traffic_symbol else {}),
                         "avg_launch_ms": avg_s * 1e3, "launches": dom["launches"],
                         "all_conv_tflops": conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None,
                         "flop_per_launch": dom["flops"] / dom["launches"],
                         "whole_step_tflops": n * FLOP_PER_PAIR / (dt / args.steps) / 1e12,
                         "gpu_time_share": dom["ms"] / sum(p["total_ms"] for p in recs)},
            "roofline_isolated": None if not iso or dom_sym not in iso else {
                "note": "same kernel symbol, single stream (no tower overlap), 5 untimed steps after the timed region",
                "achieved": iso[dom_sym]["flops"] / (iso[dom_sym]["ms"] * 1e-3) / 1e12,
                "frac": iso[dom_sym]["flops"] / (iso[dom_sym]["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "avg_launch_ms": iso[dom_sym]["ms"] / iso[dom_sym]["launches"], "launches": iso[dom_sym]["launches"]},
            "kernels": {p["name"]: round(p["total_ms"] / p["launches"], 4) for p in recs},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_pairs, seed=23)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
