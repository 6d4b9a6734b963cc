cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/ab_topk.py 2000000 64 25 db
python tools/ab_topk.py 2000000 1 25 db
python tools/ab_topk.py 250000 1024 25 db
for w in 1024 2048 4096; do ASR_TOPK_WGS=$w ASR_TOPK_SLICES=4096 python tools/ab_topk.py 2000000 64 25 db; done
