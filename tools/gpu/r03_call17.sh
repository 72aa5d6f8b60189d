#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03p; mkdir -p $OUT
timeout -k 5 300 python tools/dedup_ablation.py > $OUT/dedup_ablation_zipf.txt 2>&1; cat $OUT/dedup_ablation_zipf.txt | tail -6
timeout -k 5 300 python tools/dedup_ablation.py 1000000 100000000 900 > $OUT/dedup_ablation_hot.txt 2>&1; cat $OUT/dedup_ablation_hot.txt | tail -6
