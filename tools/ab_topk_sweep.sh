cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_code_db.py tests/test_gpu_rank_parity.py -x -q -m gpu 2>&1 | tail -3
python tools/ab_topk.py 2097152 4096 25 fused 5
python tools/ab_topk.py 2000000 64 25 db
ASR_TOPK_CHUNKS=4 python tools/ab_topk.py 2000000 64 25 db
ASR_TOPK_CHUNKS=8 python tools/ab_topk.py 2000000 64 25 db
python tools/ab_topk.py 250000 1024 25 db
ASR_TOPK_QG=2 python tools/ab_topk.py 250000 1024 25 db
ASR_TOPK_SEED=0 python tools/ab_topk.py 250000 1024 25 db
mkdir -p gpurun_out/prof_topk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_topk -o t64 -- python tools/ab_topk.py 2000000 64 25 db 20 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_topk -o t1024 -- python tools/ab_topk.py 250000 1024 25 db 20 > /dev/null 2>&1
python - <<'PY'
import csv,glob
for f in sorted(glob.glob('gpurun_out/prof_topk/*kernel_stats.csv', recursive=True)):
    print(f)
    for r in list(csv.DictReader(open(f)))[:7]:
        print('   %-80s %6s %12s %6s' % (r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage']))
PY
