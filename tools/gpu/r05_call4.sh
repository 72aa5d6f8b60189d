#!/bin/bash
# round 5: mid-size calls (where a 1 Mi-record call's time goes), soaks of the default nfagg_account path and of everything else
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c4; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_sketch_rollup_gpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3 | tee $O/tests.txt
{
for v in 8 9; do echo "== two-pass fold, 1 Mi records per call, variant $v"; timeout 100 python tools/phase_timing.py 1000000 1048576 $v 2>&1 | grep -v amdgpu | tail -9; done
echo "== single-pass cached kernel, 256 Ki records per call"; timeout 100 python tools/phase_timing.py 1000000 262144 6 2>&1 | grep -v amdgpu | tail -9
echo "== two-pass fold, 100 M records per call (for scale), variant 8"; timeout 100 python tools/phase_timing.py 1000000 100000000 8 2>&1 | grep -v amdgpu | tail -9
} > $O/midsize_phase_timing.txt 2>&1
cat $O/midsize_phase_timing.txt
timeout 200 python tests/tools/soak_account_par.py 150 1000 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak_account_a.txt
timeout 200 python tests/tools/soak_account_par.py 150 5000 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak_account_b.txt
timeout 160 python tests/tools/soak.py 120 2>&1 | grep -v amdgpu | tail -3 | tee $O/soak.txt
