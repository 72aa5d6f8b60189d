#!/bin/bash
# Round-3 evidence pass, second half (after the kernel-dedup rewrite, the page-locked buffers and the group threads): the driver's
# bench line again, the dedup lines with their rocprofv3 kernel stats and PMC traffic, the ablation of the partition pass,
# the 10 M-flow lines, the routed group fed by threads (with a kernel trace: do the sources' partitions overlap?).
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03ev2; mkdir -p $O; cd $R
b() { name=$1; shift; timeout -k 5 300 python bench.py "$@" 2>/dev/null | grep '^{' > $O/bench_$name.json; python -c "import json; j=json.load(open('$O/bench_$name.json')); print('$name', j['value'], j['ms_per_step'], j['roofline'].get('launch_ms') if 'roofline' in j else '')"; }
b n1 --steps 10 --warmup 2
b dedup_hot --dedup --hot-permille 900 --steps 5 --warmup 1 --cpu-sample 0 --no-extras
b dedup_zipf --dedup --steps 5 --warmup 1 --cpu-sample 0 --no-extras
b 10m_flows --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b dedup_10m_flows --dedup --records 125000000 --flows 10000000 --max-entries 16777216 --steps 3 --warmup 1 --cpu-sample 0 --no-extras
b group_4_routed_threads_on_one_gpu --group-devices 0,0,0,0 --group-threads --records 50000000 --steps 3 --warmup 1
timeout -k 5 200 python tools/dedup_ablation.py > $O/dedup_ablation_zipf.txt 2>&1; grep variant $O/dedup_ablation_zipf.txt
timeout -k 5 200 python tools/dedup_ablation.py 1000000 100000000 900 > $O/dedup_ablation_hot.txt 2>&1; grep variant $O/dedup_ablation_hot.txt
cd /tmp; export TMPDIR=/tmp
for tag in hot zipf; do
  extra=""; [ $tag = hot ] && extra="--hot-permille 900"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dedup_$tag -- python $R/bench.py --dedup $extra --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $O/prof_dedup_$tag.log 2>&1
  f=$(find $O/prof_dedup_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $O/dedup_${tag}_kernel_stats.csv; head -6 "$f" | cut -c1-160; fi
done
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_group_threads -- python $R/bench.py --group-devices 0,0,0,0 --group-threads --records 10000000 --flows 250000 --steps 2 --warmup 1 > $O/prof_group_threads.log 2>&1
f=$(find $O/prof_group_threads -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python - "$f" > $O/group_threads_partition_overlap.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_part_scatter" in r["Kernel_Name"] or "k_part_count" in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"].split("(")[0]) for r in rows)
over = 0
for a in range(len(iv)):
    for b in range(a + 1, len(iv)):
        if iv[b][0] >= iv[a][1]:
            break
        if iv[b][2] != iv[a][2]:
            over += min(iv[a][1], iv[b][1]) - iv[b][0]
tot = sum(e - s for s, e, _, _ in iv)
print("partition kernels: %d launches on %d streams/queues, %.3f ms in all, %.3f ms of it overlapped with a partition kernel of ANOTHER stream" % (len(iv), len(set(x[2] for x in iv)), tot / 1e6, over / 1e6))
for s, e, q, k in iv[:24]:
    print("  %s  stream/queue %s  start +%.3f ms  %.3f ms" % (k, q, (s - iv[0][0]) / 1e6, (e - s) / 1e6))
PY
cat $O/group_threads_partition_overlap.txt | head -12; fi
cd $R
BENCH_ARGS="--dedup --hot-permille 900 --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="--dedup --hot-permille 900 --steps 1 --warmup 0 --cpu-sample 0 --no-extras" bash tools/profile_bench.sh > $O/prof_dedup_pmc.log 2>&1
rm -rf $O/prof_dedup_pmc; cp -r $R/gpurun_out/prof $O/prof_dedup_pmc
find $O -name "*.csv" | wc -l
