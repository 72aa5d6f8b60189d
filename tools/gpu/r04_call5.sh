#!/bin/bash
# round 4, GPU call 5: dedup rehearsal through bench.py, ring tests (bulk drain), the default bench line with the new legs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dedup_local_fold_gpu.py -k "bench" tests/test_ring_to_device_gpu.py tests/test_c_driver.py -x -q -m gpu > gpurun_out/r04c/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04c/pytest.txt
tail -8 gpurun_out/r04c/pytest.txt
timeout 400 python bench.py > gpurun_out/r04c/bench_n1.json 2> gpurun_out/r04c/bench_n1.err
echo "bench rc $?"; tail -c 6000 gpurun_out/r04c/bench_n1.json; tail -5 gpurun_out/r04c/bench_n1.err
timeout 300 python bench.py --gpus 2 --dedup --hot-permille 900 --no-sketches --same-device --backend gloo --records 20000000 --flows 500000 --steps 3 > gpurun_out/r04c/bench_n2_dedup_rehearsal.json 2> gpurun_out/r04c/bench_n2_dedup_rehearsal.err
echo "bench2 rc $?"; tail -c 2500 gpurun_out/r04c/bench_n2_dedup_rehearsal.json; tail -5 gpurun_out/r04c/bench_n2_dedup_rehearsal.err
