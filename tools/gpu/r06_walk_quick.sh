cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06walk4; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_account_par_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('best %.3f median %.3f ms' % (j['ms_best'], j['ms_median']), j['evictions_per_call'], j['config']['evicted_flows_per_step'])"; }
for rnd in 1 2; do
for lib in libnfagg_prev.so libnfagg.so; do
  for M in 5000 10000 20000 100000; do
    echo -n "$lib M=$M: " | tee -a $O/ab.txt
    NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tools/account_5000_prof.py --steps 12 --max-entries $M 2>/dev/null | one | tee -a $O/ab.txt
  done
done
done
bash tools/gpu/r06_acc_timeline.sh 5000 > /dev/null 2>&1; cp gpurun_out/r06_acc_timeline_5000.txt $O/timeline_5000.txt
bash tools/gpu/r06_acc_timeline.sh 100000 > /dev/null 2>&1; cp gpurun_out/r06_acc_timeline_100000.txt $O/timeline_100000.txt
