#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/refresh_profiles.sh (gpurun_out/) into the committed summaries:
    profiles/<tag>_kernel_stats.csv          per-kernel launches / total / average duration (--kernel-trace --stats)
    profiles/<tag>_hbm_traffic_by_symbol.json  HBM bytes per launch of every kernel symbol (PMC, separate passes;
                                             FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md "HBM")
    profiles/<tag>_secondary_bench.jsonl, profiles/<tag>_bench_line.json
usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _one(pattern):
    hits = glob.glob(os.path.join(OUT, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def counter_mean_per_launch(directory, counter):
    rows = list(csv.DictReader(open(_one(os.path.join(directory, "**", "*counter_collection.csv")))))
    per_dispatch = collections.defaultdict(float)
    name = {}
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for d, v in per_dispatch.items():
        tot[name[d]] += v
        cnt[name[d]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    traffic_only = "--traffic-only" in sys.argv       # on the GPU box, before the final bench line is taken
    prof = os.path.join(ROOT, "profiles")
    # 1. kernel stats
    if not traffic_only:
        stats = _one(os.path.join("prof_stats", "**", "*kernel_stats.csv"))
        shutil.copy(stats, os.path.join(prof, tag + "_kernel_stats.csv"))
    # 2. HBM traffic
    rd, n_rd = counter_mean_per_launch("traffic_rd", "FETCH_SIZE")
    wr, _ = counter_mean_per_launch("traffic_wr", "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        if not k.startswith("void asr::") and not k.startswith("asr::"):
            continue
        r = rd.get(k, 0.0) * 1024.0 * 2.0          # KiB -> bytes, x2: gfx950 FETCH_SIZE counts 128-B requests as 64 B
        w = wr.get(k, 0.0) * 1024.0
        kernels[k] = dict(launches_profiled=int(n_rd.get(k, 0)), hbm_read_bytes_per_launch=r,
                          hbm_write_bytes_per_launch=w, hbm_bytes_per_launch=r + w)
    src = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_traffic.sh = the default bench.py "
           "command with ASR_SINGLE_STREAM=1; FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section); mean over the "
           "launches of each kernel symbol (both towers, autotuner launches on the same problem size included)")
    with open(os.path.join(prof, tag + "_hbm_traffic_by_symbol.json"), "w") as fp:
        json.dump(dict(source=src, kernels=kernels), fp, indent=1)
    if traffic_only:
        return
    # 3./4. bench lines
    shutil.copy(os.path.join(OUT, "secondary.jsonl"), os.path.join(prof, tag + "_secondary_bench.jsonl"))
    line = open(os.path.join(OUT, "bench_line.json")).read().strip().splitlines()[-1]
    json.loads(line)
    with open(os.path.join(prof, tag + "_bench_line.json"), "w") as fp:
        fp.write(line + "\n")
    print("wrote profiles/%s_*" % tag)


if __name__ == "__main__":
    main()
