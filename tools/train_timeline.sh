# Run ON THE GPU BOX: one batch-512 training update as a timeline per HIP stream (queue): which stream paces the step?
#   - per queue: first start, last end, busy time, kernel count
#   - the kernels in time order with queue id (for the step between the last two adam_kernel launches)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/prof_trainT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_trainT -o t -- python3 $R/tools/bench_secondary.py train > /dev/null 2>&1
python3 - <<PY
import csv, os, re, glob, collections
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_trainT/**/t_kernel_trace.csv", recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(tr) if "adam_kernel" in r["Kernel_Name"]]
a, b = ad[-2], ad[-1]
step = tr[a + 1:b + 1]
t0 = int(tr[a]["End_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void asr::", "").replace("asr::", "")
    return n[:44]
qs = collections.OrderedDict()
for r in step:
    q = r.get("Queue_Id", "?")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    d = qs.setdefault(q, dict(first=s, last=e, busy=0.0, n=0))
    d["last"] = max(d["last"], e); d["busy"] += e - s; d["n"] += 1
print("step %.1f us, %d kernels" % ((int(step[-1]["End_Timestamp"]) - t0) / 1e3, len(step)))
for q, d in qs.items():
    print("queue %s: %4d kernels, first start %8.1f, last end %8.1f, busy %8.1f us" % (q, d["n"], d["first"], d["last"], d["busy"]))
print("--- kernels >= 40 us (start, end, dur, queue, name)")
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s >= 40:
        print("%8.1f %8.1f %7.1f  q%s  %s" % (s, e, e - s, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
PY
