#!/bin/bash
# round 4, GPU call 6: u8 HLL registers, configs[3] at size on one GPU, the --dedup rehearsal, the bench line with the new legs
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sketch_rollup_gpu.py tests/test_group_gpu.py tests/test_group_local_fold_gpu.py tests/test_full_size_gpu.py tests/test_partials_gpu.py tests/test_c_driver.py tests/test_ring_to_device_gpu.py "tests/test_dedup_local_fold_gpu.py::test_bench_gpus_2_dedup_runs_the_common_stream_rehearsed_on_one_gpu" "tests/test_dedup_local_fold_gpu.py::test_group_local_fold_equals_one_dedup_table" -x -q -m gpu --durations=8 > gpurun_out/r04d/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04d/pytest.txt
tail -22 gpurun_out/r04d/pytest.txt
timeout 500 python bench.py > gpurun_out/r04d/bench_n1.json 2> gpurun_out/r04d/bench_n1.err
echo "bench rc $?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04d/bench_n1.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'])
print(json.dumps(j['cpu_baseline'])[:900])
for k,v in j['extra'].items():
    print(k, json.dumps({a:b for a,b in v.items() if a!='what'})[:700])
PY
tail -3 gpurun_out/r04d/bench_n1.err
