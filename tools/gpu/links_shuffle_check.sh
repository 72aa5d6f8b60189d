#!/bin/bash
# round 5: k_par_links with one gather per record (the left neighbour's key by shuffle): parity on the account suites, the 8 M-record call, kernel stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/links_shuffle; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_account_par_gpu.py tests/test_account_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3 | tee $O/pytest.txt
timeout 120 python tools/account_paths_bench.py --reps 3 2>&1 | grep -v amdgpu | tail -1 | cut -c1-420 | tee $O/paths.txt
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 > $O/run.json 2> $O/err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 -c "import csv,sys; [print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us') for r in csv.DictReader(open(sys.argv[1])) if 'k_par' in r['Name']]" "$f" | tee $O/kernels.txt
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
timeout 100 python tests/tools/soak_account_par.py 40 41000 2>&1 | grep -v amdgpu | tail -1
