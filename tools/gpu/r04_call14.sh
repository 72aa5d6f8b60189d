#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_full_size_gpu.py -k "configs4_at_size" -x -q -m gpu > gpurun_out/r04h/pytest2.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04h/pytest2.txt
tail -30 gpurun_out/r04h/pytest2.txt
