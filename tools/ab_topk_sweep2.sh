# scratch: trace + ablations (1: no trigger) for 1024 x 250k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 4 5; do
echo "== ablation build $a"
ASR_LIB_PATH=$GRAFT_REPO_ROOT/audio_sheet_retrieval_amd/libasr_hip_abl$a.so ASR_TF_TRACE_WGS=1024 python tools/ab_topk.py 250000 1024 25 db 2>&1 | grep -E "trace of|prologue|main loop|final|write-out|lifetime|median 0\.|rounds|start times"
done
