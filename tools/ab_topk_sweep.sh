# Run ON THE GPU BOX: the retrieval shapes of DESIGN.md section 4 ("Round 4: a resident code data base") in one go -
# parity tests first, then tools/ab_topk.py per shape (ASR_TOPK_* / ASR_RANK_* switches are taken from the environment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_code_db.py tests/test_gpu_rank_parity.py -q -m gpu -x 2>&1 | tail -3
python tools/ab_topk.py 2000000 64 25 db
python tools/ab_topk.py 2000000 1 25 db
python tools/ab_topk.py 250000 1024 25 db
python tools/ab_topk.py 2097152 4096 25 fused 5
python tools/ab_topk.py 2097152 4096 25 rank 5
python tools/ab_topk.py 2000000 64 25 stateless
