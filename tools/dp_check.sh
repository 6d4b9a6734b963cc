# Run ON THE GPU BOX: the data-parallel tests and the training workload with / without a communicator
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_data_parallel.py tests/test_gpu_multi_rank_drivers.py tests/test_gpu_train_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
ASR_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --workload train 2>/dev/null | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('one rank through RCCL', r['ms_per_step'], r['collectives_per_update'])"
python bench.py --workload train 2>/dev/null | grep "^{" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('no communicator', r['ms_per_step'])"
