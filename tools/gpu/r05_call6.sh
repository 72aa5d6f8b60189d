#!/bin/bash
# round 5: the cut walk in parts (the segment folds of the epochs a part completed overlap the next part's walk) —
# parity on the account suites, then the 8 M-record call with 1 / 2 / 4 / 8 parts (libnfagg_diag.so, NFAGG_DIAG_WALK_PARTS), a short soak
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c6; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_account_par_gpu.py tests/test_account_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 | tee $O/pytest_account.txt
echo "--- product library (4 parts)" | tee $O/walk_parts.txt
timeout 120 python tools/account_paths_bench.py --reps 3 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/walk_parts.txt
for p in 1 2 4 8; do
  echo "--- diag library, NFAGG_DIAG_WALK_PARTS=$p" | tee -a $O/walk_parts.txt
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so NFAGG_DIAG_WALK_PARTS=$p timeout 120 python tools/account_paths_bench.py --reps 3 2>$O/diag_$p.err | tail -1 | tee -a $O/walk_parts.txt
  grep "account par" $O/diag_$p.err | tail -2 | tee -a $O/walk_parts.txt
done
timeout 100 python tests/tools/soak_account_par.py 60 9000 > $O/soak.txt 2>&1; grep -v amdgpu $O/soak.txt | tail -3
rm -rf $O/prof
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $GRAFT_REPO_ROOT/tools/account_5000_prof.py --steps 3 > $O/run.json 2> $O/err.txt)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_parts4.csv
python3 -c "import csv,sys; [print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us') for r in csv.DictReader(open(sys.argv[1])) if 'k_par' in r['Name']]" "$f" | tee $O/kernels.txt
find $O/prof -name "*kernel_trace.csv" -size +20M -delete
