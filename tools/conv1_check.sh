cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_embed_parity.py tests/test_gpu_bench_sizes.py -q -m gpu -x -k "block1 or quad or bench_launch or embeddings_match" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-secondary --no-dropin --no-host-leg 2>/dev/null | grep "^{" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step']); k=r['kernels']; print({a:k[a] for a in k if 'conv1' in a})"
