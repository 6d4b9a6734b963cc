# Run ON THE GPU BOX: rebuild with extra hipcc flags and run a command, for each flag set:  bash tools/ab_build.sh "<cmd>" "<flags1>" "<flags2>" ...
export ASR_ALLOW_STALE_LIB=1
R=$GRAFT_REPO_ROOT; cd $R
# whatever ends this script (also an interrupt) puts the default build back; a left-over experiment build would be
# refused by the loader anyway: the flags are part of the library's source hash
trap 'env -u ASR_EXTRA_HIPCC_FLAGS python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1' EXIT
CMD="$1"; shift
for f in "$@"; do
  touch audio_sheet_retrieval_amd/csrc/*.hip
  ASR_EXTRA_HIPCC_FLAGS="$f" python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
  echo "== $f"; bash -c "$CMD"
done
