#!/bin/bash
# scratch: the headline and its per-layer table against the samples per internal launch (does a chunk whose block-1
# output fits the 256 MB memory-side cache make blocks 1-2 faster?)
for c in 64 125 250 500 1000; do
  python bench.py --chunk $c --steps 10 --repeats 3 --no-cpu-baseline --no-host-leg --no-isolated --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print('chunk $c: %.0f pairs/s  %.3f ms/step  ' % (d['value'], d['ms_per_step']) + ' '.join('%s=%.3f' % (n.replace('_v1',''), k[n]) for n in k if n.endswith('_v1')))
"
done > gpurun_out/chunk_sweep.log 2>&1
cat gpurun_out/chunk_sweep.log
