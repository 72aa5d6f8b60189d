#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03u; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_group_gpu.py tests/test_dedup_gpu.py -x -q -m gpu -k "dedup_mode or sharded" > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $OUT/pytest.txt | tail -15
