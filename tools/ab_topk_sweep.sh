cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_rank_parity.py tests/test_gpu_code_db.py -x -q -m gpu 2>&1 | tail -2
for q in 1 16 32 64 100; do python tools/ab_topk.py 2000000 $q 25 db; done
