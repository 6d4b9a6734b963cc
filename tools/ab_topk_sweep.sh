# scratch: bf16 x 3 filter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_code_db.py tests/test_gpu_rank_parity.py -q -m gpu -x 2>&1 | tail -3
python tools/ab_topk.py 2000000 64 25 db
python tools/ab_topk.py 2000000 1 25 db
python tools/ab_topk.py 250000 1024 25 db
python tools/ab_topk.py 2097152 4096 25 fused 5
python tools/ab_topk.py 2000000 64 25 stateless
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_seed -o seed -- python tools/ab_topk.py 2000000 64 25 db > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_seed/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print('%-90s calls %5s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
