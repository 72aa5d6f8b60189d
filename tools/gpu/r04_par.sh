#!/bin/bash
# the epoch-parallel account path (ingest_variant 31): its tests next to the default path's, a kernel trace, the bench line with its sub-leg
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r04par; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_account_par_gpu.py tests/test_account_gpu.py tests/test_config0_plumbing.py -q -m gpu 2>&1 | grep -v amdgpu | tail -4 > $O/tests.txt
cat $O/tests.txt
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 --variant 31 > $O/prof_run.json 2> $O/prof_err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" | cut -c1-200 > $O/kernel_stats_head.csv; cat $O/kernel_stats_head.csv
timeout 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04par/bench_n1.json') if l.startswith('{')][0])
print(j['value'], j['ms_per_step'])
c=j['extra']['cache_max_flows_5000']
for k in ('account_host_path','account_host_path_page_locked','account_device_resident'):
    print(k, c[k]['Mrecords_per_s'], c[k]['ms'], '| variant 31:', json.dumps(c['epochs_found_first_variant_31'].get(k, c['epochs_found_first_variant_31']))[:160])
PY
