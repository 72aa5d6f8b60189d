#!/bin/bash
# round 5, first contact: the epochs-found-first path as the default of nfagg_account — its tests, timing, kernel trace
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c1; mkdir -p $O
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_account_par_gpu.py tests/test_account_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -25 > $O/tests.txt
cat $O/tests.txt
for v in 0 30; do timeout 200 python tools/account_paths_bench.py --variant $v --reps 3 2>&1 | grep -v amdgpu | tail -1; done | tee $O/paths.txt
timeout 100 python tools/account_paths_bench.py --variant 0 --reps 3 --sketches 2>&1 | grep -v amdgpu | tail -1 | tee $O/paths_sketches.txt
rm -rf $O/prof
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 > $O/prof_run.json 2> $O/prof_err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-160 | sed 's/(.*)"/"/' > $O/kernel_stats_head.csv; cat $O/kernel_stats_head.csv
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
