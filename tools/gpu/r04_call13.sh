#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_local_fold_gpu.py -k "large_batches or sequence_window" -x -q -m gpu > gpurun_out/r04h/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04h/pytest.txt
tail -30 gpurun_out/r04h/pytest.txt
