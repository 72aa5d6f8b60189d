#!/bin/bash
# from a tree WITHOUT built artefacts: does the driver's command build what it needs and run?
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03x; mkdir -p $OUT
rm -f netobserv-ebpf-agent_amd/lib/*.so netobserv-ebpf-agent_amd/csrc/*.o netobserv-ebpf-agent_amd/csrc/diag/*.o oracle/*.so; rm -rf oracle/_ref
ls netobserv-ebpf-agent_amd/lib/ 2>/dev/null | head -3
( time timeout -k 5 900 python bench.py --steps 2 --warmup 1 > $OUT/bench_fresh.json 2> $OUT/bench_fresh.err ) 2>&1 | grep real
echo "bench rc $?"; python -c "
import json; j=json.load(open('$OUT/bench_fresh.json')); print(j['value'], j['ms_per_step'], sorted(j['extra'].keys()), j['cpu_baseline']['value'])"
tail -3 $OUT/bench_fresh.err
( time timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -4
