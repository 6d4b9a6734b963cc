export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
export ASR_SINGLE_STREAM=1 ASR_TUNE_CACHE=$R/gpurun_out/tune_cache_pmc.txt
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk 500 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/pmcw1 -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk 500 > $R/gpurun_out/pmcw1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmcw2 -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk 500 > $R/gpurun_out/pmcw2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for d in ("pmcw1", "pmcw2"):
    files = glob.glob(R + "/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "wino" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(d, k[:70])
        print("    ", {c: round(x / 1e6, 2) for c, x in sorted(v.items())})
PY
