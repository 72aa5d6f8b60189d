#!/bin/bash
# second soak pass at the final tree: other seeds
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
F='^round\|local fold\|soak ok\|Error\|assert'
: > gpurun_out/r04s/soak2.txt
for seed in 43 44; do
  timeout 300 python -u tests/tools/soak.py 140 $seed > gpurun_out/r04s/soak2_$seed.full 2>&1
  grep -v amdgpu gpurun_out/r04s/soak2_$seed.full | grep "$F" | tail -3 >> gpurun_out/r04s/soak2.txt
done
timeout 300 python -u tests/tools/soak_seeds.py 140 5000 > gpurun_out/r04s/soak2_seeds.full 2>&1
grep -v amdgpu gpurun_out/r04s/soak2_seeds.full | tail -4 >> gpurun_out/r04s/soak2.txt
cat gpurun_out/r04s/soak2.txt
