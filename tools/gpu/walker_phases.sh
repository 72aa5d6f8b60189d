#!/bin/bash
# round 5: where the cycles of the cut walk go (libnfagg_diag.so: wave 0's cycle counter per phase, printed by nfagg_account_par.inc)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$PWD/gpurun_out/walker_phases; mkdir -p $O
for p in 1 4; do
  echo "--- NFAGG_DIAG_WALK_PARTS=$p"
  NFAGG_LIB=$PWD/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so NFAGG_DIAG_WALK_PARTS=$p timeout 120 python tools/account_paths_bench.py --reps 2 2>$O/diag_$p.err | tail -1 | cut -c1-200
  grep "account par" $O/diag_$p.err | grep -B2 "8000000 records" | tail -3
done
