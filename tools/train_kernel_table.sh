# Run ON THE GPU BOX: one batch-512 training step by kernel name (launches, total us), the time no kernel runs, and the
# small kernels (< 12 us) summed - what the step would gain from fewer, larger launches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trainK -o t -- python3 $R/tools/bench_secondary.py train > /dev/null 2>&1
python3 - <<PY
import csv,os,re
R=os.environ["GRAFT_REPO_ROOT"]
tr=[r for r in csv.DictReader(open(R+"/gpurun_out/prof_trainK/t_kernel_trace.csv"))]
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
ad=[i for i,r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
a,b=ad[-2],ad[-1]
step=tr[a+1:b+1]
t0=int(step[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in step)
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in step)
busy=0; cur_s,cur_e=ev[0]
for s,e in ev[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
agg={}
small=0; nsmall=0
for r in step:
    n=re.sub(r"\(.*","",r["Kernel_Name"]); n=n.replace("void asr::","").replace("asr::","")
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    c=agg.setdefault(n,[0,0.0]); c[0]+=1; c[1]+=d
    if d<12: small+=d; nsmall+=1
print("step span %.1f us, %d kernels, sum of durations %.1f us, some kernel running %.1f us (idle %.1f us); %d kernels under 12 us: %.1f us"
      % ((t1-t0)/1e3, len(step), sum(v[1] for v in agg.values()), busy/1e3, (t1-t0-busy)/1e3, nsmall, small))
for n,(c,d) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    print("%4d %9.1f us  %s" % (c,d,n[:110]))
PY
