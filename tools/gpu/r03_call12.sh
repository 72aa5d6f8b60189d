#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_dedup_gpu.py tests/test_oracle_ref.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $OUT/pytest.txt
cd /tmp; export TMPDIR=/tmp
for cfg in "hot --dedup --hot-permille 900" "zipf --dedup"; do
set -- $cfg; name=$1; shift
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/b_$name.json 2> $OUT/b_$name.err
f=$(find $OUT/t_$name -name '*kernel_stats.csv' | head -1)
echo "== dedup $name"; [ -n "$f" ] && grep dedup "$f" | cut -c1-62,130-200
cut -c85-130 $OUT/b_$name.json
done
