#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03s; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_device_path_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "12" > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $OUT/pytest.txt
echo "chunk variant Mrec/s launch_ms(per call) value" > $OUT/sweep.txt
for chunk in 65536 262144 1048576 2097152 4194304 8388608; do
for v in 10 12; do
  timeout -k 5 120 python bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant $v --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($chunk, $v, j['roofline']['kernel_Mrecords_per_s'], j['roofline']['launch_ms'], j['value'])" >> $OUT/sweep.txt
done
done
cat $OUT/sweep.txt
