#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03o; mkdir -p $OUT
timeout -k 5 500 python -m pytest tests/test_full_size_gpu.py tests/test_optimistic_gpu.py tests/test_account_gpu.py tests/test_sequence_window_gpu.py tests/test_sketch_rollup_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "dedup or DEDUP or configs4 or window" > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -5 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for tag in hot zipf; do
  extra=""; [ $tag = hot ] && extra="--hot-permille 900"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --dedup $extra --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/prof_$tag.log 2>&1
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $OUT/dedup_${tag}_kernel_stats.csv; head -8 "$f" | cut -c1-200; fi
done
