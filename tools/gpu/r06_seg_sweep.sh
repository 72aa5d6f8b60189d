# the segment folds' thresholds (kSegShort / kSegHuge / kHugeChunk of csrc/nfagg_epoch_par.hip, set on the compiler's command line:
# lib/libnfagg_seg_*.so differ from lib/libnfagg.so in nfagg_epoch_par.o only), same box, nfagg_account_device on 8 M records
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06seg; mkdir -p $O; rm -f $O/*.txt
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('best %.3f median %.3f ms' % (j['ms_best'], j['ms_median']), j['evictions_per_call'], j['config']['evicted_flows_per_step'])"; }
for rnd in 1 2; do
for lib in libnfagg.so libnfagg_seg_s4.so libnfagg_seg_s8.so libnfagg_seg_s32.so libnfagg_seg_h2048.so libnfagg_seg_h1024.so libnfagg_seg_h1024c256.so libnfagg_seg_h512c256.so; do
  for M in 5000 10000 100000; do
    echo -n "$lib M=$M: " | tee -a $O/sweep.txt
    NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tools/account_5000_prof.py --steps 10 --max-entries $M 2>/dev/null | one | tee -a $O/sweep.txt
  done
done
done
