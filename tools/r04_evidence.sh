#!/bin/bash
# Round-4 evidence pass on the GPU box: the driver's bench line, the N > 1 rehearsals (accounter and kernel-dedup), then the rocprofv3
# kernel-trace + PMC passes for the headline AND for every extra leg (tools/profile_bench.sh; PMC_LIGHT: the two traffic counters).
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04ev; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 400 python bench.py "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms') if 'roofline' in j else '')"; }
if [ "${SKIP_BENCH:-0}" != "1" ]; then
b n1 --steps 10 --warmup 2
b n2_rehearsal_gloo_same_device --gpus 2 --same-device --backend gloo --records 40000000 --flows 1250000 --steps 3 --warmup 1
b n2_dedup_rehearsal_gloo_same_device --gpus 2 --dedup --hot-permille 900 --no-sketches --same-device --backend gloo --records 40000000 --flows 500000 --steps 3 --warmup 1
b dedup_10m_flows --dedup --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b chunk_1mi --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b group_4_local_fold_on_one_gpu --group-devices 0,0,0,0 --group-local-fold --records 50000000 --steps 3 --warmup 1
b group_4_routed_on_one_gpu --group-devices 0,0,0,0 --records 50000000 --steps 3 --warmup 1
fi
prof() { leg=$1; shift; BENCH_ARGS="$* --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="$* --steps 1 --warmup 0 --cpu-sample 0 --no-extras" PMC_LIGHT=${LIGHT:-1} bash tools/profile_bench.sh > $O/prof_$leg.log 2>&1; rm -rf $O/prof_$leg; cp -r $R/gpurun_out/prof $O/prof_$leg; echo "prof $leg: $(find $O/prof_$leg -name '*.csv' | wc -l) csv"; }
LIGHT=0 prof n1
prof configs2 --sketches
prof configs4_shape --dedup --hot-permille 900
prof dedup_zipf --dedup
prof flows_10m --flows 10000000 --max-entries 16777216
PROF_PROG="python $R/tools/account_5000_prof.py" BENCH_ARGS="--steps 2" PMC_BENCH_ARGS="--steps 1" PMC_LIGHT=1 bash tools/profile_bench.sh > $O/prof_cache_max_flows_5000.log 2>&1; rm -rf $O/prof_cache_max_flows_5000; cp -r $R/gpurun_out/prof $O/prof_cache_max_flows_5000
find $O -name "*.csv" | wc -l
