#!/bin/bash
# round 5, final library: the general randomised soak (tests/tools/soak.py) and the seed soak (tests/tools/soak_seeds.py), 150 s each
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s; mkdir -p $O
export TMPDIR=/tmp
F='^round\|local fold\|soak ok\|Error\|assert\|ok:'
timeout 400 python tests/tools/soak.py 150 5101 2>&1 | grep -v amdgpu | grep "$F" | tail -3 | tee $O/soak_general.txt
timeout 400 python tests/tools/soak_seeds.py 150 5201 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak_seeds.txt
