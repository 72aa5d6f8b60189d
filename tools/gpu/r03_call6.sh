#!/bin/bash
# round 3: spill partitions scaled to the batch, finalize with a ticket — parity, then the batch-size sweep
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03f; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_device_path_gpu.py tests/test_optimistic_gpu.py tests/test_parity_gpu.py tests/test_dedup_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $OUT/pytest.txt
echo "chunk variant Mrec/s launch_ms(per call) value" > $OUT/sweep.txt
for chunk in 65536 131072 262144 524288 1048576 2097152 4194304; do
 for v in 7 10; do
  timeout 120 python bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($chunk, $v, j['roofline']['kernel_Mrecords_per_s'], j['roofline']['launch_ms'], j['value'])" >> $OUT/sweep.txt
 done
done
cat $OUT/sweep.txt
timeout 120 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | cut -c1-200
