#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dedup_gpu.py tests/test_optimistic_gpu.py tests/test_dedup_local_fold_gpu.py tests/test_group_gpu.py "tests/test_full_size_gpu.py::test_configs4_hot_flow_dedup_bit_exact_100m" -x -q -m gpu > gpurun_out/r04f/pytest.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04f/pytest.txt
tail -15 gpurun_out/r04f/pytest.txt
for a in "--dedup" "--dedup --hot-permille 900" "--dedup --records 125000000 --flows 10000000 --max-entries 16777216"; do
  timeout 300 python bench.py $a --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$a', j['value'], j['ms_per_step'], j['roofline']['launch_ms'])"
done
