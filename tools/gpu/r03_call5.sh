#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03e; mkdir -p $OUT
timeout -k 5 200 python -m pytest tests/test_account_gpu.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -12 $OUT/pytest.txt
timeout -k 5 120 python tools/epoch_phase_timing.py 2>&1 | grep -v amdgpu.ids
