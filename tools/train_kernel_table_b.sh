# Run ON THE GPU BOX: tools/train_kernel_table.sh at another batch size ($1, default 100: the reference's BATCH_SIZE)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B=${1:-100}
cat > /tmp/tb_one.py <<PY
import sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
import bench_secondary as bs
from audio_sheet_retrieval_amd import _lib
from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
eng = _lib.Engine(bs.MODEL)
eng.set_params(synth_data.synth_params(param_shapes(bs.MODEL), seed=1, trained_like=False))
print(bs.measure_train(eng, B=$B)["ms_per_step"])
PY
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trainK -o t -- python3 /tmp/tb_one.py > /dev/null 2>&1
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
print("step span %.1f us, %d kernels, some kernel running %.1f us (idle %.1f us)" % ((t1-t0)/1e3, len(step), busy/1e3, (t1-t0-busy)/1e3))
def nm(r): return re.sub(r"\(.*","",r["Kernel_Name"]).replace("void asr::","").replace("asr::","")[:46]
for r in sorted(step,key=lambda r:int(r["Start_Timestamp"])):
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    print("%9.1f %8.1f q%s %s"%((s-t0)/1e3,(e-s)/1e3,r["Queue_Id"],nm(r)))
PY
