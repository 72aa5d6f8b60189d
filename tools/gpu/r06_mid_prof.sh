# where a 1 Mi-record fold call's time goes, kernel by kernel (rocprofv3 --kernel-trace --stats of bench.py --chunk 1048576)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_mid; mkdir -p $R/gpurun_out/prof_mid
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mid -- python $R/bench.py --chunk ${1:-1048576} --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $R/gpurun_out/r06_mid.json 2> $R/gpurun_out/r06_mid.err
f=$(find $R/gpurun_out/prof_mid -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r06_mid_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/r06_mid_kernel_stats.csv")):
    print("%-70s calls %5s avg %8.1f us  %5s%%" % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
rm -rf $R/gpurun_out/prof_mid
