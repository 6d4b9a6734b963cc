# Run ON THE GPU BOX: tools/ab_topk.py on a few shapes, the default library against variant builds
# (tools/build_variant.sh), alternating, twice.   usage: bash tools/ab_topk_libs.sh "<name1> <name2> ..." ["shape;shape"]
cd $GRAFT_REPO_ROOT
NAMES=${1:-""}
SHAPES=${2:-"2000000 64 25 db;250000 1024 25 db;2097152 512 25 db;2097152 4096 25 fused 5"}
IFS=';' read -ra SH <<< "$SHAPES"
for i in 1 2; do
  for s in "${SH[@]}"; do
    printf "%-28s default : " "$s"; python tools/ab_topk.py $s 2>&1 | tail -1
    for n in $NAMES; do
      printf "%-28s %-8s: " "$s" $n; ASR_LIB_PATH=$GRAFT_REPO_ROOT/audio_sheet_retrieval_amd/libasr_hip_$n.so python tools/ab_topk.py $s 2>&1 | tail -1
    done
  done
done
