#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_gpu.py -x -q -m gpu > gpurun_out/r04f/pytest2.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04f/pytest2.txt
tail -8 gpurun_out/r04f/pytest2.txt
