#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03suite; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc $?"; grep -n "passed\|failed\|error" $OUT/pytest_gpu.txt | tail -3
timeout -k 5 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
