#!/bin/bash
# scratch: the headline with the towers on one stream (default) and on two
for two in 0 1 0 1; do
  ASR_TWO_STREAMS=$two python bench.py --steps 20 --repeats 5 --no-cpu-baseline --no-host-leg --no-isolated --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('two_streams=$two: %.0f pairs/s  %.4f ms/step (min %.4f max %.4f)' % (d['value'], d['ms_per_step'], d['repeats']['min_ms_per_step'], d['repeats']['max_ms_per_step']))
"
done > gpurun_out/two_streams.log 2>&1
cat gpurun_out/two_streams.log
