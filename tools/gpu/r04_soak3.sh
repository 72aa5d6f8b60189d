#!/bin/bash
# third soak pass (the round's last tree): another seed of each soak
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
F='^round\|local fold\|soak ok\|Error\|assert'
timeout 200 python -u tests/tools/soak.py 110 45 > gpurun_out/r04s/soak3_45.full 2>&1
grep -v amdgpu gpurun_out/r04s/soak3_45.full | grep "$F" | tail -2 > gpurun_out/r04s/soak3.txt
timeout 200 python -u tests/tools/soak_seeds.py 90 9000 > gpurun_out/r04s/soak3_seeds.full 2>&1
grep -v amdgpu gpurun_out/r04s/soak3_seeds.full | tail -2 >> gpurun_out/r04s/soak3.txt
cat gpurun_out/r04s/soak3.txt
