#!/bin/bash
# Round-6 evidence pass on the GPU box (run from the repo root via gpurun; summarised in the build container by
# tools/summarize_prof.py, which adds the git head — the box records the library's sha256):
#   the driver's bench line, then rocprofv3 kernel-trace + PMC passes for the headline, for every extra leg (the library changed under
#   all of them: the kernel-dedup fold feeds the sketches now) and for nfagg_account at CACHE_MAX_FLOWS 5000 / 10 000 / 100 000.
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06ev; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 500 python bench.py "$@" 2>$O/bench_$name.err | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms'), j['roofline'].get('frac'), j['roofline'].get('frac_basis'))"; }
if [ "${SKIP_BENCH:-0}" != "1" ]; then
b n1 --steps 10 --warmup 2
b chunk_1mi --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
fi
prof() { leg=$1; shift; BENCH_ARGS="$* --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="$* --steps 1 --warmup 0 --cpu-sample 0 --no-extras" PMC_LIGHT=${LIGHT:-1} bash tools/profile_bench.sh > $O/prof_$leg.log 2>&1; rm -rf $O/prof_$leg; cp -r $R/gpurun_out/prof $O/prof_$leg; echo "prof $leg: $(find $O/prof_$leg -name '*.csv' | wc -l) csv"; }
LIGHT=0 prof n1
prof configs2 --sketches
prof configs4_shape --dedup --hot-permille 900
prof dedup_zipf --dedup
prof flows_10m --flows 10000000 --max-entries 16777216
for M in 5000 10000 100000; do
  PROF_PROG="python $R/tools/account_5000_prof.py --max-entries $M" BENCH_ARGS="--steps 2" PMC_BENCH_ARGS="--steps 1" PMC_LIGHT=$([ $M = 5000 ] && echo 0 || echo 1) bash tools/profile_bench.sh > $O/prof_cache_max_flows_$M.log 2>&1; rm -rf $O/prof_cache_max_flows_$M; cp -r $R/gpurun_out/prof $O/prof_cache_max_flows_$M
  echo "prof cache_max_flows_$M: $(find $O/prof_cache_max_flows_$M -name '*.csv' | wc -l) csv"
done
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*.csv" | wc -l
du -sh $O
