#!/bin/bash
# round 3, GPU call 1: new partials path + changed group code, then the default bench line with the extra legs
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_partials_gpu.py tests/test_group_local_fold_gpu.py tests/test_group_gpu.py -x -q -m gpu > gpurun_out/r03a/pytest_new.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r03a/pytest_new.txt
tail -15 gpurun_out/r03a/pytest_new.txt
timeout 300 python bench.py > gpurun_out/r03a/bench_n1.json 2> gpurun_out/r03a/bench_n1.err
echo "bench rc $?"; tail -c 3000 gpurun_out/r03a/bench_n1.json; tail -5 gpurun_out/r03a/bench_n1.err
timeout 300 python bench.py --gpus 2 --same-device --backend gloo --records 20000000 --flows 1250000 --steps 3 > gpurun_out/r03a/bench_n2_rehearsal.json 2> gpurun_out/r03a/bench_n2_rehearsal.err
echo "bench2 rc $?"; tail -c 2500 gpurun_out/r03a/bench_n2_rehearsal.json; tail -5 gpurun_out/r03a/bench_n2_rehearsal.err
timeout 200 python bench.py --group-devices 0,0,0,0 --group-local-fold --records 50000000 --steps 3 > gpurun_out/r03a/bench_group4_local.json 2>&1
tail -c 1500 gpurun_out/r03a/bench_group4_local.json
timeout 200 python bench.py --group-devices 0,0,0,0 --group-local-fold --group-threads --records 50000000 --steps 3 > gpurun_out/r03a/bench_group4_local_threads.json 2>&1
tail -c 1500 gpurun_out/r03a/bench_group4_local_threads.json
