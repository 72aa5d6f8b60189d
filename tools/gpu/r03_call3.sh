#!/bin/bash
# round 3: 16-byte-unit LDS cache + fill-driven drains — parity first, then the kernel times
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_device_path_gpu.py tests/test_parity_gpu.py tests/test_optimistic_gpu.py tests/test_full_size_gpu.py tests/test_sketch_rollup_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $OUT/pytest.txt
cd /tmp
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && grep -E "k_pass1|k_pass2|k_finalize|k_evict" "$f" | cut -c1-200
cut -c1-330 $OUT/bench.json
timeout -k 5 120 python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras --records 125000000 --flows 10000000 --max-entries 16777216 2>/dev/null | cut -c1-300
