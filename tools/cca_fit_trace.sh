# Run ON THE GPU BOX: start / duration of the kernels of one device-resident CCA fit on 25 000 samples (configs[3])
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_cca -o t -- python3 $R/tools/bench_secondary.py cca > /dev/null 2>&1
python3 - <<PY
import csv,os,re
R=os.environ["GRAFT_REPO_ROOT"]
tr=[r for r in csv.DictReader(open(R+"/gpurun_out/prof_cca/t_kernel_trace.csv"))]
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last complete fit: from the last cca_stats-like first kernel; print the last 12 kernels
def nm(r): return re.sub(r"\(.*","",r["Kernel_Name"]).replace("void asr::","").replace("asr::","")[:50]
last=tr[-14:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.1f %8.1f  %s"%((s-t0)/1e3,(e-s)/1e3,nm(r)))
PY
