# Run ON THE GPU BOX: idle time between the kernels of one headline step (would fewer / graph-captured launches pay?)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rm -rf $R/gpurun_out/prof_gaps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gaps -o t -- python3 $R/bench.py --steps 10 --repeats 2 --no-cpu-baseline --no-host-leg --no-isolated --no-secondary --no-dropin > /dev/null 2>&1
python3 - <<PY
import csv, os, glob
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_gaps/**/t_kernel_trace.csv", recursive=True)[0]
tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ranks = [i for i, r in enumerate(tr) if "rank_kernel" in r["Kernel_Name"]]
# the last 8 steps: from one rank_kernel's end to the next one's end
for a, b in list(zip(ranks[:-1], ranks[1:]))[-8:]:
    step = tr[a + 1:b + 1]
    t0 = int(tr[a]["End_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    gaps = [int(step[i]["Start_Timestamp"]) - int(step[i - 1]["End_Timestamp"]) for i in range(1, len(step))]
    first = int(step[0]["Start_Timestamp"]) - t0
    print("step %.1f us: %d kernels busy %.1f us, gaps between kernels %.1f us (max %.1f), before the first kernel %.1f us"
          % ((t1 - t0) / 1e3, len(step), busy / 1e3, sum(gaps) / 1e3, max(gaps) / 1e3, first / 1e3))
PY
