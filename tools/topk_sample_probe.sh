export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for S in 16384 32768 65536 131072; do
  rm -rf $R/gpurun_out/prof_tk; 
  ASR_TOPK_SAMPLE=$S rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tk -o t -- python3 $R/tools/ab_topk.py 2000000 64 25 db 20 > $R/gpurun_out/tk_$S.log 2>&1
  tail -1 $R/gpurun_out/tk_$S.log
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_tk/**/t_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:7]:
    print("   %-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
