export ASR_SINGLE_STREAM=1 ASR_CONV_V1=1
for a in 0 1 2 4 3 6 5; do
  ASR_ABLATE=$a python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-dropin 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('ablate $a', ' '.join('%s=%.3f'%(n,k[n]) for n in ['conv2_v1','conv3_v1','conv4_v1','conv5_v1','conv6_v1','conv7_v1','conv8_v1']))"
done
