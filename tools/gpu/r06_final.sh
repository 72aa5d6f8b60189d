#!/bin/bash
# round 6: PMC passes for the barrier-free pass 1 (variant 17, beside the product's: "by which counter"), then — at the final tree —
# the whole GPU suite, the smoke, the driver's default bench line, the general soak, the seeds soak and the account soaks.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06z; mkdir -p $O
export TMPDIR=/tmp
if [ "${PMC17:-1}" = "1" ]; then
for v in 0 17; do
  BENCH_ARGS="--variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="--variant $v --steps 1 --warmup 0 --cpu-sample 0 --no-extras" PMC_LIGHT=0 bash tools/profile_bench.sh > $O/prof_variant$v.log 2>&1
  rm -rf $O/prof_variant$v; cp -r gpurun_out/prof $O/prof_variant$v; find $O/prof_variant$v -name "*kernel_trace.csv" -size +30M -delete
  echo "prof variant $v: $(find $O/prof_variant$v -name '*.csv' | wc -l) csv"
done
fi
timeout 1800 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.txt
grep -v amdgpu $O/pytest_gpu.txt | tail -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc $?"; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r06z/bench_n1.json') if l.startswith('{')][0])
r=j['roofline']
print(j['value'], j['ms_per_step'], r['launch_ms'], r['frac'], r['frac_basis'], r['frac_traffic'], r.get('traffic_stale'), r.get('hbm_read_stream_measured_GBs'), r.get('traffic_over_measured_read_stream'))
print('cpu', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in j['cpu_baseline'].items() if k in ('value','multicore','multicore_local_fold','multicore_local_fold_all_cores','multicore_local_fold_pinned_best')}, j['cpu_baseline'].get('multicore_local_fold_pinned_best'))
for k,v in j['extra'].items():
    print(k, json.dumps({a:b for a,b in v.items() if a!='what'})[:400])
bad=[(k,a,b) for k,v in j['extra'].items() if isinstance(v,dict) for a,b in v.items() if a.startswith('frac') and isinstance(b,(int,float)) and b>1]
print('frac>1:', bad, 'roofline.frac', r['frac'])
PY
if [ "${SOAK:-1}" = "1" ]; then
timeout 220 python tests/tools/soak.py 150 > $O/soak_general.txt 2>&1; grep -v amdgpu $O/soak_general.txt | tail -2
timeout 220 python tests/tools/soak_seeds.py 120 > $O/soak_seeds.txt 2>&1; grep -v amdgpu $O/soak_seeds.txt | tail -2
timeout 220 python tests/tools/soak_account_par.py 150 20000 > $O/soak_account_a.txt 2>&1; grep -v amdgpu $O/soak_account_a.txt | tail -2
timeout 220 python tests/tools/soak_account_par.py 150 30000 --large > $O/soak_account_b.txt 2>&1; grep -v amdgpu $O/soak_account_b.txt | tail -2
fi
