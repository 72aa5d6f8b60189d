#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_device_path_gpu.py tests/test_optimistic_gpu.py tests/test_parity_gpu.py tests/test_account_gpu.py tests/test_partials_gpu.py -x -q -m gpu -k "not bench" > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $OUT/pytest.txt
echo "== pass 1, 1 Mi records per call (empty table)"; timeout -k 5 100 python tools/phase_timing.py 1000000 1048576 8 2>&1 | grep -E "kernel_ms|flush"
echo "chunk variant Mrec/s launch_ms(per call) value" > $OUT/sweep.txt
for chunk in 65536 262144 1048576 4194304; do
 for v in 7 10; do
  timeout 120 python bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($chunk, $v, j['roofline']['kernel_Mrecords_per_s'], j['roofline']['launch_ms'], j['value'])" >> $OUT/sweep.txt
 done
done
cat $OUT/sweep.txt
cd /tmp; export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -E "k_pass1|k_pass2|k_finalize|k_evict|k_merge" "$f" | cut -c1-60,150-230
cut -c1-200 $OUT/bench.json
