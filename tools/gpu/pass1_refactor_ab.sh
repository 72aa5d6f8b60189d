cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for rep in 1 2 3; do
for lib in libnfagg_head.so libnfagg.so; do
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 200 python bench.py --no-extras --cpu-sample 0 --steps 6 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib', 'value', j['value'], 'ms_per_step', j['ms_per_step'], 'launch_ms', r['launch_ms'])"
done; done
