export ASR_TUNE_ONLY=wino ASR_ALLOW_STALE_LIB=1
R=$GRAFT_REPO_ROOT; cd $R
# whatever ends this script (also an interrupt) puts the default build back; a left-over experiment build would be
# refused by the loader anyway: the flags are part of the library's source hash
trap 'env -u ASR_EXTRA_HIPCC_FLAGS python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1' EXIT
for a in 0 1 32 64 97 113; do
  touch audio_sheet_retrieval_amd/csrc/conv_wino_kernels.hip
  ASR_EXTRA_HIPCC_FLAGS="-DASR_WINOG_ABL=$a" python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
  python3 bench.py --steps 6 --warmup 2 --repeats 2 --batches 2 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('abl %4d' % $a, ' '.join('%s=%.3f'%(n[:8],k[n]) for n in ['conv2_v1','conv3_v1']))"
done
