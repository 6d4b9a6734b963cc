# Run ON THE GPU BOX: timing experiments on the global-A Winograd kernels (wrong results unless the bits say otherwise,
# -DASR_WINOG_ABL bits, see conv_wino_kernels.hip).  Rebuilds the one object file per variant; prints per-layer kernel
# times of the bench and the tuner's winog timings of conv4.
#   usage: bash tools/ablate_winog.sh "0 2 4 8 128 256 14 398"
export ASR_TUNE_ONLY=winog ASR_ALLOW_STALE_LIB=1 ASR_DEBUG=1
R=$GRAFT_REPO_ROOT; cd $R
# whatever ends this script (also an interrupt) puts the default build back; a left-over experiment build would be
# refused by the loader anyway: the flags are part of the library's source hash
trap 'env -u ASR_EXTRA_HIPCC_FLAGS python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1' EXIT
for a in ${1:-0 2 4 8 128 256}; do
  touch audio_sheet_retrieval_amd/csrc/conv_wino_kernels.hip
  ASR_EXTRA_HIPCC_FLAGS="-DASR_WINOG_ABL=$a" python3 -m audio_sheet_retrieval_amd.build > /dev/null 2>&1
  for rep in 1 2; do
  python3 bench.py --steps 6 --warmup 2 --repeats 3 --batches 2 --no-cpu-baseline --no-host-leg --no-secondary --no-dropin 2>/tmp/abl.err | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('abl %4d' % $a, ' '.join('%s=%.3f'%(n[:5],k[n]) for n in ['conv2_v1','conv3_v1','conv4_v1','conv5_v1','conv6_v1','conv7_v1','conv8_v1']))"
  grep "tune v1 conv4 wino#50" /tmp/abl.err | sed 's/.*wino#/      #/' | tr '\n' ';'; echo
  done
done
