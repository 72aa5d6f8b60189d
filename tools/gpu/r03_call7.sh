#!/bin/bash
exec < /dev/null
cd /tmp
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g; mkdir -p $OUT
for chunk in 1048576 262144; do
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$chunk -o t -- python $GRAFT_REPO_ROOT/bench.py --records 25165824 --flows 1000000 --chunk $chunk --variant 10 --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/b_$chunk.json 2> $OUT/b_$chunk.err
f=$(find $OUT/t_$chunk -name '*kernel_stats.csv' | head -1)
echo "== chunk $chunk"; [ -n "$f" ] && head -8 "$f" | cut -c1-60,150-260
done
