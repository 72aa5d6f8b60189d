#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03n; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_dedup_gpu.py -x -q -m gpu > $OUT/pytest_dedup.txt 2>&1
echo "pytest dedup rc $?"; tail -15 $OUT/pytest_dedup.txt
timeout -k 5 200 python bench.py --dedup --hot-permille 900 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench_dedup_hot.json 2> $OUT/bench_dedup_hot.err
echo "bench rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_dedup_hot.json')); print(j['value'], j['ms_per_step'])"
timeout -k 5 200 python bench.py --dedup --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench_dedup_zipf.json 2> $OUT/bench_dedup_zipf.err
echo "bench rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_dedup_zipf.json')); print(j['value'], j['ms_per_step'])"
tail -3 $OUT/bench_dedup_hot.err
