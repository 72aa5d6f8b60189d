#!/bin/bash
# A/B of two builds of libnfagg on ONE box: lib/libnfagg_prev.so (the previous commit) against lib/libnfagg.so
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04ab; mkdir -p $O
export TMPDIR=/tmp
: > $O/ab.txt
timeout 600 python -m pytest tests/test_device_path_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "not dedup and not configs3 and not configs4" 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
line() { python -c "
import json,sys
j=json.loads([l for l in open('$1') if l.startswith('{')][0]); r=j['roofline']
print('$2', j['value'], j['ms_per_step'], r['launch_ms'], r.get('lds_cache_hit_rate'), j['config'].get('unique_flows_per_gpu'))"; }
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_prev.so; else unset NFAGG_LIB; fi
    timeout 200 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $O/b_${which}_$rep.json 2>/dev/null; line $O/b_${which}_$rep.json "$which 1M-flows" >> $O/ab.txt
    timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras --flows 10000000 --max-entries 16777216 > $O/b10_${which}_$rep.json 2>/dev/null; line $O/b10_${which}_$rep.json "$which 10M-flows" >> $O/ab.txt
  done
done
unset NFAGG_LIB
timeout 200 python bench.py --records 25165824 --flows 1000000 --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $O/chunk_new.json 2>/dev/null; line $O/chunk_new.json "new 1Mi-chunks" >> $O/ab.txt
NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_prev.so timeout 200 python bench.py --records 25165824 --flows 1000000 --chunk 1048576 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $O/chunk_prev.json 2>/dev/null; line $O/chunk_prev.json "prev 1Mi-chunks" >> $O/ab.txt
cat $O/ab.txt
