# mid-size fold calls: pass 1's cache entries / tiles per workgroup / partitions swept (libnfagg_diag.so), ms per call
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06mid; mkdir -p $O; rm -f $O/sweep.txt
L=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
one() { grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/call  %.0f M records/s' % (j['roofline']['launch_ms'], j['value']))"; }
for N in 1048576 4194304; do
  for E in 1024 256 64 16; do for T in 8 4; do for P in 0; do
    echo -n "chunk=$N entries=$E tiles/wg=$T: " | tee -a $O/sweep.txt
    NFAGG_DIAG_P1_ENTRIES=$E NFAGG_DIAG_P1_TILES=$T NFAGG_LIB=$L timeout 200 python bench.py --chunk $N --steps 2 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | one | tee -a $O/sweep.txt
  done; done; done
done
for E in 64 16; do for P in 256 1024; do
    echo -n "chunk=1048576 entries=$E tiles/wg=8 parts=$P: " | tee -a $O/sweep.txt
    NFAGG_DIAG_P1_ENTRIES=$E NFAGG_DIAG_PARTS=$P NFAGG_LIB=$L timeout 200 python bench.py --chunk 1048576 --steps 2 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | one | tee -a $O/sweep.txt
done; done
