export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for cfg in "2000000 1" "250000 1"; do
  set -- $cfg
  rm -rf $R/gpurun_out/prof_scan
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scan -o t -- python3 $R/tools/ab_topk.py $1 $2 25 db 20 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_scan/**/t_kernel_stats.csv", recursive=True)[0]
print("$cfg")
for r in list(csv.DictReader(open(f)))[:5]:
    print("   %-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
