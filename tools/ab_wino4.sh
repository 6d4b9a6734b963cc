# Run ON THE GPU BOX: A/B of compile-time variants of conv3x3_wino4s:  bash tools/ab_wino4.sh "-DASR_WINO4_BURST=1" "-DASR_WINO4_BURST=3"
export ASR_TUNE_ONLY=wino4 ASR_ALLOW_STALE_LIB=1
R=$GRAFT_REPO_ROOT; cd $R
# whatever ends this script (also an interrupt) puts the default build back; a left-over experiment build would be
# refused by the loader anyway: the flags are part of the library's source hash
trap 'env -u ASR_EXTRA_HIPCC_FLAGS python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1' EXIT
for f in "$@"; do
  touch audio_sheet_retrieval_amd/csrc/conv_wino4_kernels.hip
  ASR_EXTRA_HIPCC_FLAGS="$f" python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
  python3 bench.py --steps 6 --warmup 2 --repeats 2 --batches 2 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('%-40s' % '$f', ' '.join('%s=%.3f'%(n[:5],k[n]) for n in ['conv4_v1','conv5_v1','conv6_v1','conv7_v1','conv8_v1']))"
done
