#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/walker_ab; mkdir -p $O; rm -rf $O/prof
timeout 120 python -m pytest tests/test_account_par_gpu.py -x -q -m gpu -k "found_first" 2>&1 | grep -v amdgpu | tail -2
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 > $O/run.json 2> $O/err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); python3 -c "import csv,sys; [print(r[\"Name\"][:40], r[\"Calls\"], round(float(r[\"AverageNs\"])/1e3,1), \"us\") for r in csv.DictReader(open(sys.argv[1])) if \"k_par\" in r[\"Name\"]]" "$f"
python tools/account_paths_bench.py --reps 2 2>&1 | grep -v amdgpu | tail -1 | cut -c100-250
