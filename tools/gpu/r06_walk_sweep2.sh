cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06walk; mkdir -p $O; rm -f $O/sweep2.txt
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('best %.3f median %.3f ms' % (j['ms_best'], j['ms_median']), j['evictions_per_call'], j['config']['evicted_flows_per_step'])"; }
for M in 1000 2500 5000 10000 20000 50000 100000; do
for P in 1 2 3 4 6; do
  echo -n "diag M=$M parts=$P: " | tee -a $O/sweep2.txt
  NFAGG_DIAG_WALK_PARTS=$P NFAGG_DIAG_WALK_LAST=100 NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so timeout 300 python tools/account_5000_prof.py --steps 8 --max-entries $M 2>/dev/null | grep '^{' | one | tee -a $O/sweep2.txt
done; done
