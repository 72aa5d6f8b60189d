#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
export TMPDIR=/tmp
timeout ${SEEDS_T:-300} python -u tests/tools/soak_seeds.py ${SEEDS_S:-240} ${SEED0:-1000} > gpurun_out/r04s/soak_seeds_full.txt 2>&1
grep -v amdgpu gpurun_out/r04s/soak_seeds_full.txt | tail -25 > gpurun_out/r04s/soak_seeds.txt
cat gpurun_out/r04s/soak_seeds.txt
