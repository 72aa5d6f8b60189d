#!/bin/bash
# the randomised soak (tests/tools/soak.py), two seeds; the failing round's configuration is kept
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
F='^round\|local fold\|soak ok\|Error\|assert'
timeout 300 python -m pytest tests/test_dedup_gpu.py -x -q -m gpu -k "one_launch or hot_key or many_flows" 2>&1 | tail -3 > gpurun_out/r04s/soak.txt
timeout 500 python tests/tools/mono_key_repro.py 2>&1 | grep -c " ok" >> gpurun_out/r04s/soak.txt
timeout 400 python tests/tools/soak.py ${SOAK_S:-150} 41 2>&1 | grep -v amdgpu | grep "$F" | tail -4 >> gpurun_out/r04s/soak.txt
timeout 400 python tests/tools/soak.py ${SOAK_S:-150} 42 2>&1 | grep -v amdgpu | grep "$F" | tail -4 >> gpurun_out/r04s/soak.txt
cat gpurun_out/r04s/soak.txt
