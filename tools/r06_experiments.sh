#!/bin/bash
# Round 6 experiments on the GPU box (libnfagg_diag.so): where the epochs-found-first path should begin; k_finalize + k_evict on a dense identity layout.
exec < /dev/null
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06x; mkdir -p $O; cd $R
export NFAGG_LIB=$R/netobserv-ebpf-agent_amd/lib/libnfagg_diag.so
for bar in default 16384 32768 49152; do
  if [ $bar = default ]; then unset NFAGG_DIAG_PAR_MIN; else export NFAGG_DIAG_PAR_MIN=$bar; fi
  timeout 300 python tests/tools/par_entry_bar.py 2>/dev/null | tail -1 > $O/par_entry_bar_$bar.json; cat $O/par_entry_bar_$bar.json
done
unset NFAGG_DIAG_PAR_MIN
cd /tmp; export TMPDIR=/tmp
for on in 0 1; do
  rm -rf $O/prof_dense_$on
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dense_$on -- python $R/tests/tools/dense_identity_timing.py $on 2>/dev/null | tail -1 > $O/dense_identity_$on.json
  cat $O/dense_identity_$on.json
  f=$(find $O/prof_dense_$on -name "*kernel_stats.csv" | head -1); cp $f $O/dense_identity_${on}_kernel_stats.csv; rm -rf $O/prof_dense_$on
  grep -E "k_finalize|k_evict" $O/dense_identity_${on}_kernel_stats.csv | cut -c1-160
done
