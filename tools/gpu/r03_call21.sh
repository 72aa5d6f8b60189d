#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03t; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_group_gpu.py tests/test_group_local_fold_gpu.py tests/test_sequence_window_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -5 $OUT/pytest.txt
for mode in routed routed_threads; do
  extra=""; [ $mode = routed_threads ] && extra="--group-threads"
  timeout -k 5 200 python bench.py --group-devices 0,0,0,0 --records 50000000 --flows 1000000 $extra --steps 3 --warmup 1 > $OUT/bench_group_$mode.json 2> $OUT/bench_group_$mode.err
  echo "bench $mode rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_group_$mode.json')); print(j['value'], j['ms_per_step'])"
done
