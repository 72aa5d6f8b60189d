#!/bin/bash
exec < /dev/null
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_suite; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -15 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
