#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_device_path_gpu.py tests/test_parity_gpu.py "tests/test_full_size_gpu.py::test_uniform_singleton_heavy_stream_bit_exact_80m" "tests/test_full_size_gpu.py::test_configs1_bench_stream_bit_exact_100m" tests/test_optimistic_gpu.py -x -q -m gpu > gpurun_out/r04g/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04g/pytest.txt
tail -6 gpurun_out/r04g/pytest.txt
for a in "" "--flows 10000000 --max-entries 16777216" "--records 125000000 --flows 10000000 --max-entries 16777216 --sketches" "--records 125000000 --flows 2500000 --max-entries 4194304 --sketches"; do
  timeout 300 python bench.py $a --steps 4 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$a', j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"
done
