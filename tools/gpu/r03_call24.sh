#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03w; mkdir -p $OUT
timeout -k 5 300 python -m pytest tests/test_c_driver.py -x -q -m gpu > $OUT/pytest.txt 2>&1
echo "pytest rc $?"; tail -12 $OUT/pytest.txt
