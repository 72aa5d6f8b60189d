#!/bin/bash
# round 5 kept experiment (review item 2a, pass 1's side of it): queue writes as whole 64-byte sectors with a quarter of the partitions
# (libnfagg_diag.so ingest_variant 26; results WRONG — pass 2 is not adapted) against the shipping pass 1: k_pass1's time by rocprofv3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/queue_store; rm -rf $O; mkdir -p $O
export NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
for v in 10 26 10 26; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -- python $GRAFT_REPO_ROOT/bench.py --variant $v --no-extras --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> $O/t_$v.err)
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pass1' in r['Name'] or 'k_pass2' in r['Name']: print('variant $v', r['Name'][:44], r['Calls'], 'calls, avg', round(float(r['AverageNs'])/1e6,3), 'ms')
" "$f" | tee -a $O/times.txt
  rm -rf $O/t_$v
done
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc -- python $GRAFT_REPO_ROOT/bench.py --variant 26 --no-extras --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $O/pmc.err)
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_pass1' in r['Kernel_Name']: print('variant 26 k_pass1 WRITE_SIZE', float(r['Counter_Value']), 'KiB')
" "$f" | tee -a $O/times.txt
