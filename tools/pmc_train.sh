# Run ON THE GPU BOX: SQ counters of the training step's kernels, summed per kernel symbol (tools/pmc_train.sh [filter])
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/pmc_train
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_train -o p -- python3 $R/tools/bench_secondary.py train > /dev/null 2>&1
python3 - "$1" <<'PY'
import csv, os, sys, collections
R = os.environ["GRAFT_REPO_ROOT"]
flt = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "wgrad"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(R + "/gpurun_out/pmc_train/p_counter_collection.csv")):
    k = r["Kernel_Name"]
    if flt not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, c in acc.items():
    d = n[k] or 1
    gui = c["GRBM_GUI_ACTIVE"] / d
    print("%-60s launches %3d  cycles/launch %9.0f  mfma_busy %.3f  valu_insts/wave %7.0f lds_insts/wave %6.0f  bank_conflict/idx_active %.3f  waves %d" % (
        k[:60], d, gui, c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024 * c["GRBM_GUI_ACTIVE"] / 8 / 1, 1) , c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1),
        c["SQ_INSTS_LDS"] / max(c["SQ_WAVES"], 1), c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_WAVES"] / d))
PY
