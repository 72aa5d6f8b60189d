#!/bin/bash
# round 5 experiment: pass 1 of the two-pass fold with its records requested TWO tiles ahead (libnfagg_diag.so ingest_variant 28) against the shipping kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
for rep in 1 2; do
for v in 0 28; do
  timeout 200 python bench.py --variant $v --no-extras --cpu-sample 0 --steps 5 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('variant', $v, 'value', j['value'], 'ms_per_step', j['ms_per_step'], 'launch_ms', r['launch_ms'], 'hit', r['lds_cache_hit_rate'], 'flows', j['config'].get('evicted_flows_per_step'))"
done; done
