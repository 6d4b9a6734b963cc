# Run ON THE GPU BOX: the batch-512 update with the tower gates (ASR_TRAIN_GATE_FWD / ASR_TRAIN_GATE_BWD = lead, see
# csrc/asr_api_train.hip: train_gate) against the ungated default; one shared tune cache so that every run uses the same
# schedules.  Usage: bash tools/train_gate_ab.sh ["F:B F:B ..."]   (x = unset)
cd $GRAFT_REPO_ROOT
export ASR_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/gate_tune_cache.txt
rm -f $ASR_TUNE_CACHE
python tools/train_batch_sizes.py 512 > /dev/null 2>&1        # schedules timed once
COMBOS=${1:-"x:x 0:x 1:x 2:x x:0 x:1 x:2 x:3 0:0 1:1 1:2 2:2"}
for i in 1 2; do
for c in $COMBOS; do
    f=${c%%:*}; b=${c##*:}
    ( [ "$f" != x ] && export ASR_TRAIN_GATE_FWD=$f; [ "$b" != x ] && export ASR_TRAIN_GATE_BWD=$b
      printf "fwd=%s bwd=%s  " $f $b; python tools/train_batch_sizes.py 512 2>&1 | tail -1 )
done
done
