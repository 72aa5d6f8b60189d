#!/bin/bash
# round 5: the cut walk rebuilt (four rotating register sets, two instructions per record); the link in both directions at once
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c2; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_account_par_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -5 > $O/tests.txt
cat $O/tests.txt
timeout 200 python tools/account_paths_bench.py --variant 0 --reps 3 2>&1 | grep -v amdgpu | tail -1 | tee $O/paths.txt
timeout 100 python tools/gpu/pcie_duplex.py 2>&1 | grep -v amdgpu | tail -4 | tee $O/pcie_duplex.txt
rm -rf $O/prof
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 > $O/prof_run.json 2> $O/prof_err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-110 | sed 's/(.*)"/"/' > $O/kernel_stats_head.csv; cat $O/kernel_stats_head.csv
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
