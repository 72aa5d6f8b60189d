#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
timeout 300 python tools/small_table_bench.py 5000 > gpurun_out/r04d/small_table.txt 2>&1; cat gpurun_out/r04d/small_table.txt | grep -v amdgpu
timeout 600 python -m pytest tests/test_full_size_gpu.py -k configs3 "tests/test_dedup_local_fold_gpu.py::test_bench_gpus_2_dedup_runs_the_common_stream_rehearsed_on_one_gpu" -x -q -m gpu > gpurun_out/r04d/pytest2.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04d/pytest2.txt
tail -12 gpurun_out/r04d/pytest2.txt
