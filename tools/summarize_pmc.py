#!/usr/bin/env python
"""Join rocprofv3's per-dispatch PMC counters (tools/profile_round.sh, gpurun_out/pmc_*/) with the library's launch
log (ASR_LAUNCH_LOG: label, algorithmic FLOP / bytes, kernel symbol of every timed launch, in launch order) and write
the committed summaries:

    profiles/<tag>_kernel_stats.csv             rocprofv3 --kernel-trace --stats of the default bench command
    profiles/<tag>_pmc_by_layer.csv             per layer label (conv4_v1 ...): launches, HBM read / written bytes per
                                                launch, MFMA-pipe busy fraction, VALU / wait shares, clock
    profiles/<tag>_hbm_traffic_by_symbol.json   per kernel SYMBOL (what bench.py's roofline aggregates by, like
    profiles/<tag>_mfma_busy_by_symbol.json     rocprofv3 --stats): the same quantities over all launches of the symbol

Corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE (KiB) is doubled on gfx950 (128-B requests tallied
at 64 B); FETCH_SIZE and WRITE_SIZE come from separate passes; SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per
v_mfma_f32_16x16x4_f32), summed over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
usage: python tools/summarize_pmc.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
N_SIMD = 1024


def _glob_one(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True))
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


def dispatches(directory):
    """[(dispatch id, kernel name, {counter: value})] in dispatch order"""
    rows = list(csv.DictReader(open(_glob_one(os.path.join(directory, "**", "*counter_collection.csv")))))
    per = collections.OrderedDict()
    for r in rows:
        d = int(r["Dispatch_Id"])
        rec = per.setdefault(d, [r["Kernel_Name"], collections.defaultdict(float)])
        rec[1][r["Counter_Name"]] += float(r["Counter_Value"])
    return [(d, per[d][0], dict(per[d][1])) for d in sorted(per)]


def launch_log(path):
    out = []
    for line in open(path):
        parts = line.rstrip("\n").split("\t")
        if len(parts) == 4:
            out.append(dict(label=parts[0], flops=float(parts[1]), bytes=float(parts[2]), symbol=parts[3]))
    return out


def attribute(directory, log_path):
    """walk the launch log and the dispatch list together: every logged launch with a kernel symbol is matched with
    the next dispatch of exactly that kernel -> [(label, symbol, counters, flops, bytes)]"""
    disp = dispatches(directory)
    log = [e for e in launch_log(log_path) if e["symbol"]]
    # the log only holds the TIMED launches (profiling on): align from the END of the dispatch list
    out, j = [], len(disp) - 1
    for e in reversed(log):
        while j >= 0 and disp[j][1] != e["symbol"]:
            j -= 1
        if j < 0:
            raise SystemExit("launch log and dispatch list do not line up at %s" % e["label"])
        out.append((e["label"], e["symbol"], disp[j][2], e["flops"], e["bytes"]))
        j -= 1
    return out[::-1]


def mean_by(records, key_index, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for rec in records:
        if counter in rec[2]:
            tot[rec[key_index]] += rec[2][counter]
            cnt[rec[key_index]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    prof = os.path.join(ROOT, "profiles")
    shutil.copy(_glob_one(os.path.join("prof_stats", "**", "*kernel_stats.csv")),
                os.path.join(prof, tag + "_kernel_stats.csv"))
    rd = attribute("pmc_rd", os.path.join(OUT, "launch_rd.log"))
    wr = attribute("pmc_wr", os.path.join(OUT, "launch_wr.log"))
    sq = attribute("pmc_sq", os.path.join(OUT, "launch_sq.log"))
    try:
        lds = attribute("pmc_lds", os.path.join(OUT, "launch_lds.log"))
    except SystemExit:
        lds = []
    sym_of, alg = {}, {}
    layers_of, alg_sym = collections.defaultdict(set), collections.defaultdict(list)
    for label, symbol, _, flops, nbytes in sq:
        sym_of[label] = symbol
        alg[label] = (flops, nbytes)
        layers_of[symbol].add(label)
        alg_sym[symbol].append(nbytes)
    tables = {}
    for key_index, name in ((0, "layer"), (1, "symbol")):
        t = {}
        r_mean, r_cnt = mean_by(rd, key_index, "FETCH_SIZE")
        w_mean, _ = mean_by(wr, key_index, "WRITE_SIZE")
        mf, _ = mean_by(sq, key_index, "SQ_VALU_MFMA_BUSY_CYCLES")
        mops, _ = mean_by(sq, key_index, "SQ_INSTS_VALU_MFMA_MOPS_F32")
        gui, _ = mean_by(sq, key_index, "GRBM_GUI_ACTIVE")
        wc, _ = mean_by(sq, key_index, "SQ_WAVE_CYCLES")
        wait, _ = mean_by(sq, key_index, "SQ_WAIT_ANY")
        winst, _ = mean_by(sq, key_index, "SQ_WAIT_INST_ANY")
        act, _ = mean_by(sq, key_index, "SQ_ACTIVE_INST_ANY")
        valu, _ = mean_by(lds, key_index, "SQ_ACTIVE_INST_VALU")
        conf, _ = mean_by(lds, key_index, "SQ_LDS_BANK_CONFLICT")
        lact, _ = mean_by(lds, key_index, "SQ_LDS_IDX_ACTIVE")
        for k in sorted(set(r_mean) | set(mf)):
            rb = r_mean.get(k, 0.0) * 1024.0 * 2.0        # KiB -> bytes, x2 (gfx950 FETCH_SIZE)
            wb = w_mean.get(k, 0.0) * 1024.0
            cyc = gui.get(k, 0.0) / 8.0                   # kernel duration in shader cycles
            t[k] = dict(launches_profiled=int(r_cnt.get(k, 0)),
                        hbm_read_bytes_per_launch=rb, hbm_write_bytes_per_launch=wb, hbm_bytes_per_launch=rb + wb,
                        mfma_busy=(mf[k] / (N_SIMD * cyc)) if k in mf and cyc > 0 else None,
                        mfma_busy_cycles_per_launch=mf.get(k), mfma_mops_f32_per_launch=mops.get(k),
                        duration_cycles=cyc or None,
                        wave_cycles_share=dict(wait_any=wait.get(k, 0) / wc[k], wait_inst_any=winst.get(k, 0) / wc[k],
                                               active_inst_any=act.get(k, 0) / wc[k]) if wc.get(k) else None,
                        active_inst_valu_per_wave_cycle=(valu[k] / wc[k]) if k in valu and wc.get(k) else None,
                        lds_bank_conflict_share=(conf[k] / lact[k]) if k in conf and lact.get(k) else None)
        tables[name] = t
    src = ("rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group, tools/profile_round.sh) of "
           "`bench.py --steps 4 --warmup 1`, timed launches only, attributed through ASR_LAUNCH_LOG; FETCH_SIZE doubled "
           "per MI355X_MICROARCH.md (HBM); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8)")
    # Which layers a symbol served is the tuner's choice, box by box: a by-symbol figure only describes a run whose
    # (symbol -> layers) map is the same.  Every record therefore names its layers and its algorithmic bytes, the same
    # quantities are kept per LAYER as well, and the tune cache of this run is committed next to them
    # (profiles/<tag>_tune_cache.txt - bench.py runs those schedules by default and refuses a lookup that does not fit).
    def extra(k):
        return dict(layers=sorted(layers_of.get(k, ())),
                    algorithmic_bytes_per_launch=(sum(alg_sym[k]) / len(alg_sym[k])) if alg_sym.get(k) else None)
    traffic_fields = ("launches_profiled", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch")
    busy_fields = ("mfma_busy", "mfma_busy_cycles_per_launch", "mfma_mops_f32_per_launch", "duration_cycles",
                   "wave_cycles_share", "active_inst_valu_per_wave_cycle", "lds_bank_conflict_share")
    by_layer_t = {k: dict({f: v[f] for f in traffic_fields}, symbol=sym_of.get(k), algorithmic_bytes_per_launch=alg.get(k, (0, None))[1])
                  for k, v in tables["layer"].items()}
    by_layer_b = {k: dict({f: v[f] for f in busy_fields}, symbol=sym_of.get(k)) for k, v in tables["layer"].items()}
    with open(os.path.join(prof, tag + "_hbm_traffic_by_symbol.json"), "w") as fp:
        json.dump(dict(source=src, kernels={k: dict({f: v[f] for f in traffic_fields}, **extra(k))
                                            for k, v in tables["symbol"].items()}, layers=by_layer_t), fp, indent=1)
    with open(os.path.join(prof, tag + "_mfma_busy_by_symbol.json"), "w") as fp:
        json.dump(dict(source=src, kernels={k: dict({f: v[f] for f in busy_fields}, **extra(k))
                                            for k, v in tables["symbol"].items()}, layers=by_layer_b), fp, indent=1)
    cache = os.path.join(OUT, "tune_cache.txt")
    if os.path.exists(cache):
        shutil.copy(cache, os.path.join(prof, tag + "_tune_cache.txt"))
    with open(os.path.join(prof, tag + "_pmc_by_layer.csv"), "w") as fp:
        w = csv.writer(fp)
        w.writerow(["layer", "symbol", "algorithmic_flop", "algorithmic_bytes", "hbm_read_MB", "hbm_write_MB",
                    "hbm_over_algorithmic", "mfma_busy", "duration_cycles", "wait_any", "wait_inst_any",
                    "active_inst_any", "lds_conflict_share"])
        for k, v in tables["layer"].items():
            fl, by = alg.get(k, (0.0, 0.0))
            sh = v["wave_cycles_share"] or {}
            w.writerow([k, sym_of.get(k, ""), "%.0f" % fl, "%.0f" % by, "%.2f" % (v["hbm_read_bytes_per_launch"] / 1e6),
                        "%.2f" % (v["hbm_write_bytes_per_launch"] / 1e6),
                        "%.3f" % (v["hbm_bytes_per_launch"] / by) if by else "",
                        "" if v["mfma_busy"] is None else "%.3f" % v["mfma_busy"],
                        "" if not v["duration_cycles"] else "%.0f" % v["duration_cycles"],
                        "%.3f" % sh.get("wait_any", 0), "%.3f" % sh.get("wait_inst_any", 0),
                        "%.3f" % sh.get("active_inst_any", 0),
                        "" if v["lds_bank_conflict_share"] is None else "%.3f" % v["lds_bank_conflict_share"]])
    print("wrote profiles/%s_{kernel_stats.csv,pmc_by_layer.csv,hbm_traffic_by_symbol.json,mfma_busy_by_symbol.json}" % tag)


if __name__ == "__main__":
    main()
