#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03final; mkdir -p $OUT
timeout -k 5 400 python bench.py --steps 10 --warmup 2 > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "rc $? lines $(wc -l < $OUT/bench_n1.json)"
python -c "
import json; j=json.load(open('$OUT/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['frac_stream_floor'], j['roofline_evict']['frac']); print({k:(v.get('Mrecords_per_s') if isinstance(v,dict) else None) for k,v in j['extra'].items()})"
