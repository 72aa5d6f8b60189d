# one nfagg_account_device call (8 M records, CACHE_MAX_FLOWS $1, default 5000) as a kernel timeline: which launch waits for what
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; M=${1:-5000}
rm -rf $R/gpurun_out/prof_tl; mkdir -p $R/gpurun_out/prof_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -- python $R/tools/account_5000_prof.py --steps 3 --max-entries $M > $R/gpurun_out/r06_acc_tl_$M.json 2> $R/gpurun_out/r06_acc_tl_$M.err
f=$(find $R/gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python $R/tools/acc_timeline.py $f > $R/gpurun_out/r06_acc_timeline_$M.txt
cat $R/gpurun_out/r06_acc_tl_$M.json; tail -80 $R/gpurun_out/r06_acc_timeline_$M.txt
rm -rf $R/gpurun_out/prof_tl
