# the two-pass fold's cache probe window (kProbe of csrc/nfagg_ingest_part.hip, set on the compiler's command line:
# lib/libnfagg_probe_*.so differ from lib/libnfagg.so in nfagg_ingest_part.o only), same box: the 100 M-record fold call at 1 M and 10 M flows
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06probe; mkdir -p $O; rm -f $O/*.txt
for rnd in 1 2; do
for lib in libnfagg.so libnfagg_probe_4.so libnfagg_probe_6.so libnfagg_probe_12.so libnfagg_probe_16.so; do
  echo "== $lib" | tee -a $O/sweep.txt
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tests/tools/pass1_free_ab.py --variants 0 --reps 4 2>/dev/null | tail -1 | tee -a $O/sweep.txt
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/$lib timeout 300 python tests/tools/pass1_free_ab.py --variants 0 --reps 3 --flows 10000000 2>/dev/null | tail -1 | tee -a $O/sweep.txt
done
done
