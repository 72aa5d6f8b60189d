#!/bin/bash
# round 5: soaks of nfagg_account's default path after the capture fix (two seeds ranges, 150 s each)
cd "$GRAFT_REPO_ROOT"
O=$PWD/gpurun_out/r05c5; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python tests/tools/soak_account_par.py 150 1000 > $O/soak_account_a.txt 2>&1; grep -v amdgpu $O/soak_account_a.txt | tail -4
timeout 200 python tests/tools/soak_account_par.py 150 5000 > $O/soak_account_b.txt 2>&1; grep -v amdgpu $O/soak_account_b.txt | tail -4
