#!/bin/bash
# ingest_variant 31 after a kernel change: its tests, then the 8 M-record call timed and traced
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r04par; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_account_par_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 > $O/tests2.txt
cat $O/tests2.txt
rm -rf $O/prof2
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 --variant 31 > $O/prof2_run.json 2> $O/prof2_err.txt)
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-150 | sed 's/(.*)"/"/' > $O/kernel_stats2_head.csv; cat $O/kernel_stats2_head.csv
for v in 31 0 31; do timeout 100 python tools/account_5000_prof.py --steps 3 --variant $v 2>&1 | grep -v amdgpu | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('variant', j['config']['ingest_variant'], 'ms_per_call', j['ms_per_call'], 'evictions', j['evictions_per_call'], 'flows', j['config']['evicted_flows_per_step'])"; done | tee $O/timing2.txt
