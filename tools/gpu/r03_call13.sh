#!/bin/bash
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03ev; mkdir -p $O; cd $R
timeout -k 5 300 python bench.py --steps 10 --warmup 2 2>/dev/null | grep '^{' > $O/bench_n1.json
python -c "import json; j=json.load(open('$O/bench_n1.json')); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j['roofline']['frac_stream_floor'], j['roofline'].get('frac_traffic')); print(json.dumps(j.get('extra'), indent=0)[:3000])"
timeout -k 5 120 python tools/epoch_phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/epoch_timing.txt; cat $O/epoch_timing.txt
