#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03r; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_dedup_gpu.py -x -q -m gpu > $OUT/pytest_dedup.txt 2>&1
echo "pytest dedup rc $?"; tail -4 $OUT/pytest_dedup.txt
for tag in hot zipf 10m; do
  extra=""; [ $tag = hot ] && extra="--hot-permille 900"; [ $tag = 10m ] && extra="--records 125000000 --flows 10000000 --max-entries 16777216"
  timeout -k 5 200 python bench.py --dedup $extra --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench_dedup_$tag.json 2> $OUT/bench_dedup_$tag.err
  echo "bench $tag rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_dedup_$tag.json')); print(j['value'], j['ms_per_step'], j['roofline']['lds_cache_hit_rate'])"
done
