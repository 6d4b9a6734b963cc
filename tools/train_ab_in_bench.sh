# Run ON THE GPU BOX: the batch-512 update as bench.py measures it (secondary leg, after the headline), default against single
# switches - which of round 5's defaults costs what in THAT harness
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.3f ms' % ('$*', d['secondary']['configs[2]_train_step_b512']['ms_per_step']))"; }
for i in 1 2; do
run X=default
run ASR_POOL_TIES=first
run ASR_TRAIN_FUSED_REDUCE=0
run ASR_TRAIN_WGRAD_STREAM2=1
run ASR_POOL_TIES=first ASR_TRAIN_FUSED_REDUCE=0 ASR_TRAIN_WGRAD_STREAM2=1
done
