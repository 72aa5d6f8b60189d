#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03v; mkdir -p $OUT
timeout -k 5 300 python tools/dedup_ablation.py > $OUT/abl_zipf.txt 2>&1; grep "variant 10\|Error\|error" $OUT/abl_zipf.txt | head -5
timeout -k 5 300 python tools/dedup_ablation.py 1000000 100000000 900 > $OUT/abl_hot.txt 2>&1; grep "variant 10\|Error\|error" $OUT/abl_hot.txt | head -5
