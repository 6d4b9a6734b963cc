#!/bin/bash
# scratch: the two-rank training test, three times as is and three times with the forward F(4x4) builds out of the tuner
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python -m pytest "tests/test_gpu_data_parallel.py::test_two_rank_training_step_equals_single_context_step" -q -m gpu 2>&1 | grep -E "^E  .*assert|passed|failed" | cut -c1-300
done
for i in 1 2 3; do
  ASR_TRAIN_WINO4=5 python -m pytest "tests/test_gpu_data_parallel.py::test_two_rank_training_step_equals_single_context_step" -q -m gpu 2>&1 | grep -E "^E  .*assert|passed|failed" | cut -c1-300
done
