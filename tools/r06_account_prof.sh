cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "100000 0" "100000 1048576" "5000 1048576"; do set -- $cfg
  rm -rf $R/gpurun_out/prof_acc; mkdir -p $R/gpurun_out/prof_acc
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_acc -- python $R/tools/account_5000_prof.py --steps 3 --max-entries $1 --chunk $2 > $R/gpurun_out/r06_acc_$1_$2.json 2> $R/gpurun_out/r06_acc_$1_$2.err
  f=$(find $R/gpurun_out/prof_acc -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r06_acc_$1_$2_kernel_stats.csv
  cat $R/gpurun_out/r06_acc_$1_$2.json
done
rm -rf $R/gpurun_out/prof_acc
