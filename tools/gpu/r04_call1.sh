#!/bin/bash
# round 4, GPU call 1: kernel-dedup local fold (sub-flow tables + join), ring -> staging -> device, then the whole GPU suite
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_local_fold_gpu.py tests/test_ring_to_device_gpu.py tests/test_c_driver.py -x -q -m gpu > gpurun_out/r04a/pytest_new.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04a/pytest_new.txt
tail -25 gpurun_out/r04a/pytest_new.txt
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_dedup_local_fold_gpu.py --deselect tests/test_ring_to_device_gpu.py --deselect tests/test_c_driver.py > gpurun_out/r04a/pytest_all.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04a/pytest_all.txt
tail -15 gpurun_out/r04a/pytest_all.txt
