#!/bin/bash
# Round-3 evidence pass on the GPU box: the driver's bench line, the N > 1 rehearsals, the group lines, the batch-size sweep,
# the epoch-kernel timing, then the rocprofv3 kernel-trace + PMC passes for configs[1] (tools/profile_bench.sh).
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03ev; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 300 python bench.py "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms') if 'roofline' in j else '')"; }
b n1 --steps 10 --warmup 2
b n2_rehearsal_gloo_same_device --gpus 2 --same-device --backend gloo --records 40000000 --flows 1250000 --steps 3 --warmup 1
b 10m_flows --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b configs3_share --sketches --records 125000000 --flows 1250000 --steps 5 --warmup 1 --cpu-sample 0 --no-extras
b hot --hot-permille 900 --steps 5 --warmup 1 --cpu-sample 0 --no-extras
b dedup_zipf --dedup --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b chunk_1mi --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b group_4_local_fold_on_one_gpu --group-devices 0,0,0,0 --group-local-fold --records 50000000 --steps 3 --warmup 1
b group_4_routed_on_one_gpu --group-devices 0,0,0,0 --records 50000000 --steps 3 --warmup 1
echo "chunk variant Mrec/s launch_ms(per call) value" > $O/batch_size_sweep.txt
for chunk in 65536 131072 262144 524288 1048576 2097152 4194304; do
 for v in 7 10; do
  timeout -k 5 120 python bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($chunk, $v, j['roofline']['kernel_Mrecords_per_s'], j['roofline']['launch_ms'], j['value'])" >> $O/batch_size_sweep.txt
 done
done
cat $O/batch_size_sweep.txt
timeout -k 5 120 python tools/epoch_phase_timing.py 2>&1 | grep -v amdgpu.ids > $O/epoch_timing.txt; cat $O/epoch_timing.txt
bash tools/profile_bench.sh > $O/prof1.log 2>&1; rm -rf $O/prof_n1; cp -r $R/gpurun_out/prof $O/prof_n1
find $O -name "*.csv" | wc -l
