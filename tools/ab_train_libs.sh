# Run ON THE GPU BOX: the batch-512 update, default library against variant builds (tools/build_variant.sh), alternating,
# three rounds, one shared tune cache per library.   usage: bash tools/ab_train_libs.sh "<name1> <name2> ..."
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  printf "default : "; ASR_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/abt_default.txt python tools/train_batch_sizes.py 512 2>&1 | tail -1
  for n in $1; do
    printf "%-8s: " $n; ASR_TUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/abt_$n.txt ASR_LIB_PATH=$GRAFT_REPO_ROOT/audio_sheet_retrieval_amd/libasr_hip_$n.so python tools/train_batch_sizes.py 512 2>&1 | tail -1
  done
done
