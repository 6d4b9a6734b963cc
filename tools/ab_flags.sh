# Run ON THE GPU BOX: A/B of compile-time variants of conv_wino_kernels.hip (correct results in every variant).
#   usage: bash tools/ab_flags.sh "<flags A>" "<flags B>" ...     (an empty string = the default build)
export ASR_ALLOW_STALE_LIB=1
R=$GRAFT_REPO_ROOT; cd $R
# whatever ends this script (also an interrupt) puts the default build back; a left-over experiment build would be
# refused by the loader anyway: the flags are part of the library's source hash
trap 'env -u ASR_EXTRA_HIPCC_FLAGS python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1' EXIT
for f in "$@"; do
  touch audio_sheet_retrieval_amd/csrc/conv_wino_kernels.hip
  ASR_EXTRA_HIPCC_FLAGS="$f" python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
  for rep in 1 2; do
  python3 bench.py --steps 10 --warmup 2 --repeats 3 --batches 4 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('[%s] %.0f pairs/s ' % ('$f', d['value']), ' '.join('%s=%.3f'%(n[:5],k[n]) for n in ['conv2_v1','conv3_v1','conv4_v1','conv5_v1','conv6_v1','conv7_v1','conv8_v1']))"
  done
done
