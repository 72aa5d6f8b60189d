#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03z; mkdir -p $OUT
BENCH_ARGS="--dedup --steps 3 --warmup 1 --cpu-sample 0 --no-extras" PMC_BENCH_ARGS="--dedup --steps 1 --warmup 0 --cpu-sample 0 --no-extras" bash tools/profile_bench.sh > $OUT/prof.log 2>&1
rm -rf $OUT/prof_dedup_zipf_pmc; cp -r $GRAFT_REPO_ROOT/gpurun_out/prof $OUT/prof_dedup_zipf_pmc
find $OUT -name "*.csv" | wc -l
