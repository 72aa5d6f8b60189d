#!/bin/bash
# round 5: the whole GPU suite, the smoke, the driver's default bench line, two 150-s soaks of nfagg_account — at the final tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05z; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.txt
grep -v amdgpu $O/pytest_gpu.txt | tail -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc $?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r05z/bench_n1.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j['roofline']['frac_traffic'], j['roofline'].get('traffic_source'))
for k,v in j['extra'].items():
    print(k, json.dumps({a:b for a,b in v.items() if a!='what'})[:300])
bad=[(k,a,b) for k,v in j['extra'].items() if isinstance(v,dict) for a,b in v.items() if a.startswith('frac') and isinstance(b,(int,float)) and b>1]
print('frac>1:', bad)
PY
if [ "${SOAK:-1}" = "1" ]; then
timeout 200 python tests/tools/soak_account_par.py 150 20000 > $O/soak_account_a.txt 2>&1; grep -v amdgpu $O/soak_account_a.txt | tail -2
timeout 200 python tests/tools/soak_account_par.py 150 30000 > $O/soak_account_b.txt 2>&1; grep -v amdgpu $O/soak_account_b.txt | tail -2
fi
