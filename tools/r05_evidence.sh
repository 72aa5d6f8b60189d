#!/bin/bash
# Round-5 evidence pass on the GPU box (run from the repo root via gpurun; summarised in the build container by
# tools/summarize_prof.py, which adds the git head — the box records the library's sha256):
#   the driver's bench line, the world-size-1 RCCL preflight line, then rocprofv3 kernel-trace + PMC passes for the headline
#   and for the legs whose kernels changed this round (the default nfagg_account path at CACHE_MAX_FLOWS 5000).
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05ev; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 400 python bench.py "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms') if 'roofline' in j else '')"; }
if [ "${SKIP_BENCH:-0}" != "1" ]; then
b n1 --steps 10 --warmup 2
b n1_force_dist_nccl_world_1 --gpus 1 --force-dist --backend nccl --records 40000000 --flows 1250000 --steps 3 --warmup 1
b chunk_1mi --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
fi
prof() { leg=$1; shift; BENCH_ARGS="$* --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="$* --steps 1 --warmup 0 --cpu-sample 0 --no-extras" PMC_LIGHT=${LIGHT:-1} bash tools/profile_bench.sh > $O/prof_$leg.log 2>&1; rm -rf $O/prof_$leg; cp -r $R/gpurun_out/prof $O/prof_$leg; echo "prof $leg: $(find $O/prof_$leg -name '*.csv' | wc -l) csv"; }
LIGHT=0 prof n1
if [ "${ALL_LEGS:-0}" = "1" ]; then
prof configs2 --sketches
prof configs4_shape --dedup --hot-permille 900
prof dedup_zipf --dedup
prof flows_10m --flows 10000000 --max-entries 16777216
fi
PROF_PROG="python $R/tools/account_5000_prof.py" BENCH_ARGS="--steps 2" PMC_BENCH_ARGS="--steps 1" PMC_LIGHT=0 bash tools/profile_bench.sh > $O/prof_cache_max_flows_5000.log 2>&1; rm -rf $O/prof_cache_max_flows_5000; cp -r $R/gpurun_out/prof $O/prof_cache_max_flows_5000
find $O -name "*.csv" | wc -l
find $O -name "*kernel_trace.csv" -size +30M -delete
