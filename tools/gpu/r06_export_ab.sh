# pass 1's cache entries exported to pass 2 (ingest_variant 0) against merged into the table by pass 1 itself (18): parity subset, then A/B
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06exp; mkdir -p $O; rm -f $O/*.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_device_path_gpu.py tests/test_account_par_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest.txt
one() { grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('%.4f ms/call  %.0f M records/s' % (j['roofline']['launch_ms'], j['value']))"; }
for rnd in 1 2; do for N in 524288 1048576 4194304; do for V in 18 0; do
  echo -n "chunk=$N variant=$V: " | tee -a $O/ab.txt
  timeout 200 python bench.py --chunk $N --variant $V --steps 2 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | one | tee -a $O/ab.txt
done; done; done
timeout 400 python tests/tools/pass1_free_ab.py --variants 18,0 --reps 5 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 400 python tests/tools/pass1_free_ab.py --variants 18,0 --reps 3 --flows 10000000 2>/dev/null | tail -1 | tee -a $O/ab.txt
timeout 400 python tests/tools/pass1_free_ab.py --variants 18,0 --reps 3 --hot 900 2>/dev/null | tail -1 | tee -a $O/ab.txt
