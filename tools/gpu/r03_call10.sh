#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03i; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_sequence_window_gpu.py tests/test_optimistic_gpu.py tests/test_group_local_fold_gpu.py tests/test_group_gpu.py tests/test_account_gpu.py tests/test_partials_gpu.py -q -m gpu -k "not bench" > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -30 $OUT/pytest.txt
