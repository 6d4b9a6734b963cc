# Run ON THE GPU BOX after tools/profile_round.sh <tag>: the lines that are not part of the default bench command
#   pool2m (configs[4]) fused and as round 3's two passes, the data-parallel training line (one rank through RCCL),
#   per-kernel stats of the retrieval shapes, the training step by kernel, the weight-gradient LDS counters
TAG=${1:-r05}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
P=$R/gpurun_out/profiles_out; mkdir -p $P
cd $R
python3 bench.py --workload pool2m 2> /dev/null | grep '^{' | tail -1 > $P/${TAG}_pool2m_bench_line.json
ASR_POOL2M_SEPARATE=1 python3 bench.py --workload pool2m 2> /dev/null | grep '^{' | tail -1 > $P/${TAG}_pool2m_two_pass_bench_line.json
python3 bench.py --workload train 2> /dev/null | grep '^{' | tail -1 > $P/${TAG}_train_workload_bench_line.json
ASR_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --workload train 2> /dev/null | grep '^{' | tail -1 > $P/${TAG}_train_workload_rccl_world1_bench_line.json
cd /tmp
rm -rf $R/gpurun_out/prof_topk4; mkdir -p $R/gpurun_out/prof_topk4
for cfg in "2000000 1 db" "2000000 64 db" "250000 1024 db" "2097152 4096 fused" "2000000 64 stateless"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_topk4 -o t_$1_$2_$3 -- python3 $R/tools/ab_topk.py $1 $2 25 $3 10 > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_topk4 -o pmc_scan_rd -- python3 $R/tools/ab_topk.py 2000000 1 25 db 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_topk4 -o pmc_f64_rd -- python3 $R/tools/ab_topk.py 2000000 64 25 db 5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_topk4 -o pmc_fused -- python3 $R/tools/ab_topk.py 2097152 4096 25 fused 5 > /dev/null 2>&1
cd $R
python3 - > $P/${TAG}_topk_kernel_stats.txt <<'PY'
import csv, glob, os, collections
for f in sorted(glob.glob("gpurun_out/prof_topk4/t_*kernel_stats.csv")):
    print(os.path.basename(f).replace("_kernel_stats.csv", "").replace("t_", "shape (pool, queries, mode): "))
    for r in list(csv.DictReader(open(f)))[:8]:
        print("   %-84s calls %5s  avg %10.1f us  %6s %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open("gpurun_out/prof_topk4/pmc_fused_counter_collection.csv")):
    k = r["Kernel_Name"]
    if "topk_filter" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
# HBM reads per launch of the pool-streaming kernels (FETCH_SIZE is in KiB and is doubled on gfx950, MI355X_MICROARCH.md)
durs = {}
for f in glob.glob("gpurun_out/prof_topk4/t_2000000_*_db_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        durs[(os.path.basename(f), r["Name"])] = float(r["AverageNs"]) / 1e3
for tag, name, statf in (("pmc_scan_rd", "topk_scan_kernel", "t_2000000_1_db_kernel_stats.csv"),
                         ("pmc_f64_rd", "topk_filter_kernel", "t_2000000_64_db_kernel_stats.csv")):
    tot, cnt = 0.0, 0
    path = "gpurun_out/prof_topk4/%s_counter_collection.csv" % tag
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if name in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); cnt += 1
    us = [v for (f, k), v in durs.items() if f == statf and name in k]
    if cnt and us:
        mb = tot / cnt * 1024.0 * 2.0 / 1e6
        print("%s: HBM read %.1f MB per launch (algorithmic 256.0 MB: 2 M rows x 128 B), %.1f us per launch -> %.2f TB/s (%.2f of the 8 TB/s peak)"
              % (name, mb, us[0], mb / us[0], mb / us[0] / 8.0))
print("SQ counters of the filter kernels, fused 4096 x 2^21 call:")
for k, c in acc.items():
    d = n[k] or 1
    print("   %-70s launches %3d cycles/launch %9.0f mfma_busy %.3f lds_conflict_share %.3f" % (
        k[:70], d, c["GRBM_GUI_ACTIVE"] / d, c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024 * c["GRBM_GUI_ACTIVE"] / 8, 1),
        c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)))
PY
bash tools/train_kernel_table.sh > $P/${TAG}_train_kernel_table.txt 2>&1
bash tools/train_timeline.sh > $P/${TAG}_train_timeline.txt 2>&1
python3 tools/train_batch_sizes.py 64 100 512 > $P/${TAG}_train_batch_sizes.txt 2>&1
ASR_BENCH_MODEL=mutopia_ccal_cont_rsz python3 bench.py --no-cpu-baseline --no-secondary --no-dropin 2> /dev/null | grep '^{' | tail -1 > $P/${TAG}_rsz_bench_line.json
bash tools/pmc_train.sh wgrad > $P/${TAG}_train_wgrad_pmc.txt 2>&1
ls -la $P | tail -12
