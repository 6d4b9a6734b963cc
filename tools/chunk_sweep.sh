#!/bin/bash
# Run ON THE GPU BOX: the headline against (a) the samples per internal launch and (b) one / two tower streams
#   (DESIGN.md "Round 4, the tower kernels": 64 / 125 / 250 / 500 / 1000 -> 169 / 215 / 253 / 274 / 289 k pairs/s;
#   two streams 279-287 k against 289 k)
cd $GRAFT_REPO_ROOT
digest='import sys, json
d = json.loads(sys.stdin.read()); k = d["kernels"]
print("%s: %.0f pairs/s  %.4f ms/step  " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s=%.3f" % (n.replace("_v1", ""), k[n]) for n in k if n.endswith("_v1")))'
common="--steps 10 --repeats 3 --no-cpu-baseline --no-host-leg --no-isolated --no-secondary --no-dropin"
for c in 64 125 250 500 1000; do
  python bench.py --chunk $c $common 2>/dev/null | grep '^{' | tail -1 | python -c "$digest" "chunk $c"
done
for two in 0 1 0 1; do
  ASR_TWO_STREAMS=$two python bench.py $common 2>/dev/null | grep '^{' | tail -1 | python -c "$digest" "two_streams=$two"
done
