# Run ON THE GPU BOX: per-launch bandwidth of the BatchNorm kernels of the batch-512 step, towers on one stream
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
ASR_TRAIN_ONE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_train1 -o t -- python3 $R/tools/bench_secondary.py train > /dev/null 2>&1
python3 - <<PY
import csv,os
R=os.environ["GRAFT_REPO_ROOT"]
tr=[r for r in csv.DictReader(open(R+"/gpurun_out/prof_train1/t_kernel_trace.csv"))]
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
ad=[i for i,r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
a,b=ad[-2],ad[-1]
B=512
L=[(160,200,12,0),(160,200,12,1),(80,100,24,0),(80,100,24,1),(40,50,48,0),(40,50,48,1),(20,25,48,0),(20,25,48,1)]
L2=[(92,42,12,0),(92,42,12,1),(46,21,24,0),(46,21,24,1),(23,10,48,0),(23,10,48,1),(11,5,48,0),(11,5,48,1)]
cnt={}; tot={}
for r in tr[a:b]:
    n=r["Kernel_Name"]; d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    for key in ("bn_bwd_reduce","bn_bwd_apply","bn_apply_elu_pool"):
        if key in n:
            k=cnt.get(key,0); cnt[key]=k+1
            t=0 if k<8 else 1
            bi=k%8 if key=="bn_apply_elu_pool" else 7-(k%8)
            H,W,C,p=(L if t==0 else L2)[bi]
            S=B*H*W*C*4
            byts=S*((2 if key=="bn_bwd_apply" else 1)+(0.25 if p else 1))
            tot[key]=tot.get(key,0)+d
            if t==0: print("%-18s v%d b%d %8.1f us %7.2f TB/s" % (key,t+1,bi+1,d,byts/d/1e6))
print({k:round(v) for k,v in tot.items()})
PY
