#!/bin/bash
# round 6, after the last kernel change (five-unit gather): longer soaks at the final tree, fresh seeds
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06soak; mkdir -p $O
timeout 420 python tests/tools/soak.py 360 11 > $O/soak_general.txt 2>&1; grep -v amdgpu $O/soak_general.txt | tail -1
timeout 300 python tests/tools/soak_seeds.py 240 7000 > $O/soak_seeds.txt 2>&1; grep -v amdgpu $O/soak_seeds.txt | tail -1
timeout 300 python tests/tools/soak_account_par.py 240 120000 > $O/soak_account_a.txt 2>&1; grep -v amdgpu $O/soak_account_a.txt | tail -1
timeout 300 python tests/tools/soak_account_par.py 240 130000 --large > $O/soak_account_b.txt 2>&1; grep -v amdgpu $O/soak_account_b.txt | tail -1
