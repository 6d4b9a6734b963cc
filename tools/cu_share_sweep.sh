#!/bin/bash
# Run ON THE GPU BOX: the headline with the towers on two streams and the chip divided between them (ASR_TOWER2_CUS)
cd $GRAFT_REPO_ROOT
digest='import sys, json
d = json.loads(sys.stdin.read())
print("%s: %.0f pairs/s  %.4f ms/step (min %.4f max %.4f)" % (sys.argv[1], d["value"], d["ms_per_step"], d["repeats"]["min_ms_per_step"], d["repeats"]["max_ms_per_step"]))'
common="--steps 20 --repeats 5 --no-cpu-baseline --no-host-leg --no-isolated --no-secondary --no-dropin"
python bench.py $common 2>/dev/null | grep '^{' | tail -1 | python -c "$digest" "one stream"
for c in 0 16 24 32 40 48 64; do
  ASR_TWO_STREAMS=1 ASR_TOWER2_CUS=$c python bench.py $common 2>/dev/null | grep '^{' | tail -1 | python -c "$digest" "two streams, tower 2 on $c CUs"
done
python bench.py $common 2>/dev/null | grep '^{' | tail -1 | python -c "$digest" "one stream"
