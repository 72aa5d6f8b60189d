#!/bin/bash
exec < /dev/null
cd /tmp
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j; mkdir -p $OUT
for cfg in "hot --dedup --hot-permille 900" "zipf --dedup"; do
set -- $cfg; name=$1; shift
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$name -o t -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $OUT/b_$name.json 2> $OUT/b_$name.err
f=$(find $OUT/t_$name -name '*kernel_stats.csv' | head -1)
echo "== dedup $name"; [ -n "$f" ] && head -9 "$f" | cut -c1-75,130-240
cut -c1-160 $OUT/b_$name.json
done
