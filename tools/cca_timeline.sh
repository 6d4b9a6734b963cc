# Run ON THE GPU BOX: start / duration of every kernel around the CCALayer stage of one training step (launch gaps)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_train -o t -- python3 $R/tools/bench_secondary.py train > /dev/null 2>&1
python3 - <<PY
import csv,os
R=os.environ["GRAFT_REPO_ROOT"]
tr=[r for r in csv.DictReader(open(R+"/gpurun_out/prof_train/t_kernel_trace.csv"))]
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
ad=[i for i,r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
a,b=ad[-2],ad[-1]
step=tr[a+1:b+1]
t0=int(step[0]["Start_Timestamp"])
idx=[i for i,r in enumerate(step) if any(k in r["Kernel_Name"] for k in ("cca_train_kernel","ct_","loss_rows","loss_cols"))]
lo,hi=idx[0],idx[-1]
print("step span %.1f us, kernels %d" % ((int(step[-1]["End_Timestamp"])-t0)/1e3, len(step)))
for i in range(max(0,lo-4), min(len(step),hi+7)):
    r=step[i]; s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%8.1f %8.1f  q%s  %s" % ((s-t0)/1e3, (e-s)/1e3, r["Queue_Id"], r["Kernel_Name"][:50]))
PY
