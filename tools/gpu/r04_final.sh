#!/bin/bash
# round 4: the whole GPU suite, the smoke, the driver's default bench line at the final tree
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r04z/pytest_gpu.txt 2>&1
echo "pytest rc $?" >> gpurun_out/r04z/pytest_gpu.txt
tail -22 gpurun_out/r04z/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 500 python bench.py > gpurun_out/r04z/bench_n1.json 2> gpurun_out/r04z/bench_n1.err
echo "bench rc $?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04z/bench_n1.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j['roofline']['frac_traffic'], j['roofline'].get('traffic_source'))
print(json.dumps(j['cpu_baseline']['multicore']))
for k,v in j['extra'].items():
    print(k, json.dumps({a:b for a,b in v.items() if a!='what'})[:420])
bad=[(k,a,b) for k,v in j['extra'].items() if isinstance(v,dict) for a,b in v.items() if a.startswith('frac') and isinstance(b,(int,float)) and b>1]
print('frac>1:', bad)
PY
