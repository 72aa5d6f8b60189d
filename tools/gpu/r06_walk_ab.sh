# the cut walk un-gated from the host: parity of the account path first, then a same-box A/B of two builds (lib/libnfagg_prev.so = the
# tree before, lib/libnfagg.so) on nfagg_account_device, 8 M records, CACHE_MAX_FLOWS 5000 / 10000 / 100000; a sweep over the number
# of parts and the last part's share (diag build); then the new call's timeline
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06walk; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_account_par_gpu.py tests/test_isa_pins.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('best %.3f median %.3f ms' % (j['ms_best'], j['ms_median']), j['evictions_per_call'], j['config']['evicted_flows_per_step'])"; }
for rnd in 1 2; do
for lib in libnfagg_prev.so libnfagg.so; do
  for M in 5000 10000 100000; do
    echo -n "$lib M=$M: " | tee -a $O/ab.txt
    NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tools/account_5000_prof.py --steps 12 --max-entries $M 2>/dev/null | one | tee -a $O/ab.txt
  done
done
done
for M in 5000 100000; do
for P in 1 3 4 5 6 8; do for L in 100 50 25; do
  echo -n "diag M=$M parts=$P last=$L%: " | tee -a $O/sweep.txt
  NFAGG_DIAG_WALK_PARTS=$P NFAGG_DIAG_WALK_LAST=$L NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so timeout 300 python tools/account_5000_prof.py --steps 8 --max-entries $M 2>/dev/null | grep '^{' | one | tee -a $O/sweep.txt
done; done; done
bash tools/gpu/r06_acc_timeline.sh 5000 > /dev/null 2>&1; cp gpurun_out/r06_acc_timeline_5000.txt $O/timeline_5000.txt
