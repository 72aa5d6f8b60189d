#!/bin/bash
# Round-2 evidence pass on the GPU box: bench lines for every configuration DESIGN.md quotes, the batch/table-size sweep, the
# host path, then the rocprofv3 kernel-trace + PMC passes for configs[1] and configs[2] (tools/profile_bench.sh).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02ev; mkdir -p $O; cd $R
b() { name=$1; shift; timeout 300 python bench.py "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms') if 'roofline' in j else '')"; }
b n1 --steps 10 --warmup 2
b sketches --sketches --steps 5 --warmup 1 --cpu-sample 0
b hot --hot-permille 900 --steps 5 --warmup 1 --cpu-sample 0
b dedup_hot --dedup --hot-permille 900 --steps 5 --warmup 1 --cpu-sample 0
b dedup_zipf --dedup --steps 5 --warmup 1 --cpu-sample 0
b configs3_share --sketches --records 125000000 --flows 1250000 --steps 5 --warmup 1 --cpu-sample 0
b 10m_flows --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0
b chunk_1mi --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0
b group_4_on_one_gpu --group-devices 0,0,0,0 --records 25000000 --flows 250000 --steps 3 --warmup 1 --sketches
b group_1 --group-devices 0 --steps 3 --warmup 1 --sketches
bash tools/sweep_batch_and_table_sizes.sh 2>&1 | grep -v amdgpu.ids > $O/sweep.txt
timeout 200 python tools/host_path_bench.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > $O/host_path.txt
bash tools/profile_bench.sh > $O/prof1.log 2>&1; rm -rf $O/prof_n1; cp -r $R/gpurun_out/prof $O/prof_n1
BENCH_ARGS="--sketches --steps 3 --warmup 1 --cpu-sample 0" PMC_BENCH_ARGS="--sketches --steps 1 --warmup 0 --cpu-sample 0" bash tools/profile_bench.sh > $O/prof2.log 2>&1; rm -rf $O/prof_sk; cp -r $R/gpurun_out/prof $O/prof_sk
find $O -name "*.csv" | wc -l
