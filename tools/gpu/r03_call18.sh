#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03q; mkdir -p $OUT
timeout -k 5 200 python bench.py --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench_10m_flows.json 2> $OUT/b1.err
echo "rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_10m_flows.json')); r=j['roofline']; print('acc 10M', j['value'], j['ms_per_step'], r['lds_cache_hit_rate'], r['launch_ms'])"
timeout -k 5 200 python bench.py --dedup --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench_dedup_10m_flows.json 2> $OUT/b2.err
echo "rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_dedup_10m_flows.json')); r=j['roofline']; print('dedup 10M', j['value'], j['ms_per_step'], r['lds_cache_hit_rate'], r['launch_ms'])"
tail -2 $OUT/b2.err
